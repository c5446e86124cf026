#!/bin/bash
# Round-4 GPU calls, one function per call (provenance of the gpurun tags the files under profiles/ cite: r04a ...).
#   usage on the GPU box (through gpurun):  bash tools/runs_r04.sh <letter>        e.g.  gpurun -- 'bash tools/runs_r04.sh a'
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R

# first pass: parity of the staged forward kernel (all GPU tests, then the scoring tests under the old kernel), A/B of its variants,
# the VALU counter calibration (tools/valu_rates under --pmc), the 8-rank rehearsal of the N>1 bench line on one device
call_a() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04a; mkdir -p $O
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( NP_HMM_KERNEL=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q ) > $O/pytest_k1.log 2>&1; echo "k1 pytest rc=$?" >> $O/pytest_k1.log
V=nanopolish_amd/variants
( timeout 900 python tools/hmm_ab.py --pool 4000 --tile 10 "@NP_HMM_KERNEL=1" "@NP_HMM_KERNEL=2" $V/libnp_hip_f2s.so $V/libnp_hip_f2w5.so $V/libnp_hip_f2w5s.so "@NP_HMM_KERNEL=2" "@NP_HMM_KERNEL=1" ) > $O/hmm_ab.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/valu_pmc -o v -- $R/tools/valu_rates > $R/$O/valu_rates_pmc.log 2>&1 )
$R/tools/valu_rates > $O/valu_rates.log 2>&1
find $O/valu_pmc -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/valu_counters.csv; find $O/valu_pmc -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/valu_kernel_trace.csv
( time NP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --pool 1000 --tile 5 --steps 2 --warmup 1 --legs 0 ) > $O/bench_8rank_gloo.json 2> $O/bench_8rank_gloo.err; echo "rc=$?" >> $O/bench_8rank_gloo.err
tail -4 $O/pytest.log; tail -3 $O/pytest_k1.log; cat $O/hmm_ab.log; head -c 600 $O/valu_counters.csv; tail -c 1800 $O/bench_8rank_gloo.json; tail -5 $O/bench_8rank_gloo.err
}

# kernel B, staged: where do the Gaussians come from (slab loads in two halves / all first / none = ablation), scheduling strategy
call_b() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04b; mkdir -p $O
V=nanopolish_amd/variants
( timeout 900 python tools/hmm_ab.py --pool 4000 --tile 10 "@NP_HMM_KERNEL=1" "@NP_HMM_KERNEL=2" $V/libnp_hip_g1.so $V/libnp_hip_g2.so $V/libnp_hip_g2s.so $V/libnp_hip_g1ilp.so "@NP_HMM_KERNEL=1" ) > $O/hmm_ab.log 2>&1
cat $O/hmm_ab.log
}

# kernel B, staged, emissions one step ahead (Gaussian requests behind the K chain): against kernel 1, with free emissions, 5 waves, stage barriers
call_c() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04c; mkdir -p $O
V=nanopolish_amd/variants
( timeout 900 python tools/hmm_ab.py --pool 4000 --tile 10 "@NP_HMM_KERNEL=1" "@NP_HMM_KERNEL=2" $V/libnp_hip_e1.so $V/libnp_hip_e1.so@NP_HMM_KERNEL=1 $V/libnp_hip_e_w5.so $V/libnp_hip_e_s.so "@NP_HMM_KERNEL=2" "@NP_HMM_KERNEL=1" ) > $O/hmm_ab.log 2>&1
cat $O/hmm_ab.log
( timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sites.py -m gpu -x -q ) 2>&1 | tail -3
}

# counters of the forward kernels, block-major (1) against stage-major (2), same workload (8192 reads, call-methylation step only)
call_d() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04d; mkdir -p $O
for k in 1 2; do
  W="python $GRAFT_REPO_ROOT/tools/pmc_workload.py --reads 8192 --ea-reads 0 --reps 2"
  ( cd /tmp && NP_HMM_KERNEL=$k timeout 120 $W > $O/units_k$k.json 2> $O/units_k$k.err ); echo "units k$k rc=$?"
  pass() { local name=$1; shift
    ( cd /tmp && NP_HMM_KERNEL=$k timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/k${k}_$name -o $name -- $W > $O/k${k}_$name.log 2>&1 ); echo "k$k $name rc=$?" | tee -a $O/passes.log; }
  pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
  pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  pass sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH GRBM_GUI_ACTIVE
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04d"
for k in (1, 2):
    tot = defaultdict(float); n = defaultdict(int)
    for f in glob.glob(O + "/k%d_*/**/*counter_collection.csv" % k, recursive=True):
        for r in csv.DictReader(open(f)):
            if "np_hmm_forward" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("kernel", k, {c: (round(v / 2), n[c] // 2) for c, v in sorted(tot.items())})
PY
cat $O/units_k1.json | tail -1 | cut -c1-400; cat $O/units_k2.json | tail -1 | cut -c1-400
}

# the threaded batch binding: parity (one context, two and three contexts on one device), throughput at 512 / 8192 records
call_e() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04e; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_batch_dropin.py tests/test_gpu_variants_dropin.py tests/test_gpu_eventalign_dropin.py tests/test_gpu_dropin.py tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 900 python tests/bench_batch_dropin.py --sizes ${SIZES:-512,8192} ) > $O/batch_dropin.json 2> $O/batch_dropin.err; echo "rc=$?" >> $O/batch_dropin.err
tail -6 $O/pytest.log; cat $O/batch_dropin.json | cut -c1-3000; tail -4 $O/batch_dropin.err
}

# the threaded batch binding: worker-pool size and batch size
call_f() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f; mkdir -p $O
for t in 12 14 16 20 24; do
  ( NP_HOST_THREADS=$t timeout 600 python tests/bench_batch_dropin.py --sizes 8192 --skip sync,pipelined_adc_ref_writer,pipelined_adc_4ctx,pipelined_adc_2ctx ) > $O/threads_$t.json 2> $O/threads_$t.err
  echo "threads $t: $(grep -o '"pipelined_adc": {"value": [0-9.]*\|"pipelined": {"value": [0-9.]*' $O/threads_$t.json | tr '\n' ' ')"
done
( timeout 600 python tests/bench_batch_dropin.py --sizes 2048,32768 --skip sync,pipelined_adc_ref_writer,pipelined_adc_4ctx ) > $O/sizes.json 2> $O/sizes.err
grep -o '"batch_size": [0-9]*\|"pipelined[a-z_0-9]*": {"value": [0-9.]*' $O/sizes.json | tr '\n' ' '
}

# per-call drop-in under OpenMP (flat combining) beside the reference's own function; its tests
call_g() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04g; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_variants_dropin.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( timeout 600 python tests/bench_percall_dropin.py ) > $O/percall.json 2> $O/percall.err; echo "rc=$?" >> $O/percall.err
tail -5 $O/pytest.log; cat $O/percall.json; tail -3 $O/percall.err
}

# chain kernel: round-4 sweep (pre-shifted codes, max3 + equality-chain arg-max, DPP-add neighbour exchange): parity + timing; per-call drop-in again
call_h() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04h; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 64 ) > $O/ea.json 2> $O/ea.err
( NP_HIP_LIB=$GRAFT_REPO_ROOT/nanopolish_amd/variants/libnp_hip_ea_r3.so timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 0 ) > $O/ea_r3.json 2> $O/ea_r3.err
for v in $EA_VARIANTS; do ( NP_HIP_LIB=$GRAFT_REPO_ROOT/nanopolish_amd/variants/libnp_hip_$v.so timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 64 ) > $O/$v.json 2> $O/$v.err; done
tail -3 $O/pytest.log
for f in ea ea_r3 $EA_VARIANTS; do echo "$f: $(grep -o '"value": [0-9.]*\|"eventalign_chain": [0-9.]*\|"rows_match": [a-z]*\|"copies_identical": [a-z]*' $O/$f.json | tr '\n' ' ')"; done
cat $O/percall.json
}

# direct-RNA reads on the device path; whole GPU suite
call_i() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04i; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_rna.py -m gpu -x -q ) > $O/pytest_rna.log 2>&1; echo "rna pytest rc=$?" >> $O/pytest_rna.log
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest_rna.log; tail -6 $O/pytest.log
}

# configs[2] literally: reads from a 5 Mb genome with indel / clip CIGARs, every row of 512 records against the reference
call_j() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04j; mkdir -p $O
( time timeout 900 python tests/bench_eventalign.py --steps 3 --warmup 1 ) > $O/ea.json 2> $O/ea.err; echo "rc=$?" >> $O/ea.err
cut -c1-2500 $O/ea.json; tail -5 $O/ea.err
}

# counter passes over the shipped kernels (A, B in the call-methylation mix and in the variants shape, the chain)
call_k() {
cd "$GRAFT_REPO_ROOT" || exit 1
PASS_TIMEOUT=240 bash profiles/collect_r04_pmc.sh r04pmc 8192 2>&1 | tail -14
}

# VALU calibration again: tools/valu_rates with 2 KB of LDS per wave (the first calibration's 16 KB capped a CU at 10 waves, not 32) at 8/4/2/1 waves per SIMD
call_l() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04l; mkdir -p $O
$R/tools/valu_rates > $O/valu_rates.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $R/$O/valu_pmc -o v -- $R/tools/valu_rates > $R/$O/valu_rates_pmc.log 2>&1 )
find $O/valu_pmc -name "*.db" | head -1 | xargs -I{} cp {} $O/valu.db
head -70 $O/valu_rates.log
}

# kernel B's log-sum with the constant out of the scalar register (NP_LSE_FORM 1: literal, 2: vector register): issue cost of the forms, A/B, parity
call_m() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04m; mkdir -p $O
$R/tools/valu_rates l > $O/valu_rates_forms.log 2>&1; cat $O/valu_rates_forms.log
V=nanopolish_amd/variants
( timeout 900 python tools/hmm_ab.py --pool 4000 --tile 10 "" $V/libnp_hip_lse1.so $V/libnp_hip_lse2.so "" $V/libnp_hip_lse1.so $V/libnp_hip_lse2.so ) > $O/hmm_ab.log 2>&1; cat $O/hmm_ab.log
for l in lse1 lse2; do ( NP_HIP_LIB=$R/$V/libnp_hip_$l.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sites.py -m gpu -x -q ) 2>&1 | tail -2; done
}

# kernel B with two blocks' log-sum chains in lock step (NP_HMM_PAIR, the tree's default from here on) against the block-major step, with and
# without SLP vectorisation; parity of the new default (scoring tests + whole GPU suite), variants leg
call_n() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04n; mkdir -p $O
V=nanopolish_amd/variants
( timeout 900 python tools/hmm_ab.py --pool 4000 --tile 10 "" $V/libnp_hip_old.so $V/libnp_hip_pairslp.so $V/libnp_hip_nopair.so $V/libnp_hip_oldnoslp.so "" $V/libnp_hip_old.so ) > $O/hmm_ab.log 2>&1; cat $O/hmm_ab.log
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
( timeout 600 python tests/bench_variants.py --steps 3 --warmup 1 ) > $O/variants.json 2> $O/variants.err; cut -c1-900 $O/variants.json; tail -3 $O/variants.err
}

# kernel B's new default (block-major, literal constant, no SLP) against the shipped build; the chain kernel without SLP vectorisation
call_o() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04o; mkdir -p $O
V=nanopolish_amd/variants
( timeout 900 python tools/hmm_ab.py --pool 4000 --tile 10 "" $V/libnp_hip_old.so "" $V/libnp_hip_old.so ) > $O/hmm_ab.log 2>&1; cat $O/hmm_ab.log
for l in "" $R/$V/libnp_hip_eanoslp.so "" $R/$V/libnp_hip_eanoslp.so; do
  ( NP_HIP_LIB=$l timeout 600 python tests/bench_eventalign.py --pool 1000 --tile 20 --steps 3 --cpu-sample 0 ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['kernel_ms_per_step'])"
done 2>&1 | tee $O/ea_ab.log
( NP_HIP_LIB=$R/$V/libnp_hip_eanoslp.so timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py tests/test_gpu_rna.py -m gpu -x -q ) 2>&1 | tail -2
( timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -2
}

# the drop-in tests under ASan + UBSan (shims) and UBSan-trap (library host code)
call_p() {
cd "$GRAFT_REPO_ROOT" || exit 1
( time timeout 1500 python -m pytest tests/test_gpu_sanitizers.py -m gpu -x -q ) 2>&1 | tail -40
}

# kernel B at five / six waves per SIMD (640- / 768-thread workgroups, 96 / 80 registers with a few spills); the split-phase guard test; shims after the scalings fix
call_q() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04q; mkdir -p $O
V=nanopolish_amd/variants
( timeout 900 python tools/hmm_ab.py --pool 4000 --tile 10 "" $V/libnp_hip_w5.so $V/libnp_hip_w6.so "" $V/libnp_hip_w5.so $V/libnp_hip_w6.so ) > $O/hmm_ab.log 2>&1; cat $O/hmm_ab.log
for l in w5 w6; do ( NP_HIP_LIB=$R/$V/libnp_hip_$l.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sites.py -m gpu -x -q ) 2>&1 | tail -2; done
( timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -3
}

# the one-wave-per-read glue kernels (event map, recalibration, work-item groups) at four waves per workgroup against one
call_r() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04r; mkdir -p $O
V=nanopolish_amd/variants
( timeout 900 python tools/hmm_ab.py --pool 4000 --tile 10 "" $V/libnp_hip_gluew1.so "" $V/libnp_hip_gluew1.so ) > $O/hmm_ab.log 2>&1; cat $O/hmm_ab.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/tr -o t -- python $R/tools/hmm_ab.py --pool 4000 --tile 10 "" > $R/$O/tr.log 2>&1 ); f=$(find $O/tr -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f | cut -c1-150 | head -24; rm -rf $O/tr
( timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -3
}

# glue of step i-1 beside the aligner of step i (two streams, the aligner's grid at 8 / 7 / 6 waves per SIMD)
call_s() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04s; mkdir -p $O
( timeout 600 python tools/overlap_probe.py ) > $O/overlap.log 2>&1; tail -8 $O/overlap.log
}

# np_hmm_score_host's small-batch path + the combiner's spin hand-over: parity, per-call calls/s
call_t() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04t; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -3
( timeout 600 python tests/bench_percall_dropin.py ) > $O/percall.json 2> $O/percall.err; cat $O/percall.json; tail -2 $O/percall.err
}

# soak against the reference itself with the round's final kernels: indel / clipped records through the whole chain (seeds 8-10)
call_u() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04u; mkdir -p $O; rm -f $O/soak.jsonl
for seed in 8 9 10; do timeout 900 python tests/gpu_soak.py --reads 1500 --seed $seed >> $O/soak.jsonl 2>> $O/soak.err; echo "rc=$?"; done
cat $O/soak.jsonl | cut -c1-400; tail -2 $O/soak.err
}

# the t-statistic's final quotient by a checked approximation (np_tstat_quotient): parity of the default build and of the two extremes
# (the reference's arithmetic everywhere / for every sample whose approximation is checked... never: margin 2^29), detection time from raw
call_v() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04v; mkdir -p $O
V=$R/nanopolish_amd/variants
for l in "" $V/libnp_hip_tsexact.so $V/libnp_hip_tsalways.so; do
  ( NP_HIP_LIB=$l timeout 600 python -m pytest tests/test_gpu_events.py tests/test_gpu_reflevel.py tests/test_gpu_rna.py -m gpu -x -q ) 2>&1 | tail -1
  ( NP_HIP_LIB=$l timeout 600 python bench.py --steps 3 --warmup 1 --from-raw 1 --cpu-sample 0 --legs 0 --streamed 0 --ragged 0 ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['roofline']['kernel_ms_per_step'])"
done 2>&1 | tee $O/ts_ab.log
for seed in 11; do NP_HIP_LIB= timeout 900 python tests/gpu_soak.py --reads 1500 --seed $seed | cut -c1-300; done
}

# v_readlane's lane select through M0 (kernel A's two band-end reads, the chain kernel's two event reads) against the scalar-register form
call_w() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04w; mkdir -p $O
V=$R/nanopolish_amd/variants
( timeout 900 python tools/hmm_ab.py --pool 10000 --tile 10 "" $V/libnp_hip_rlold.so "" $V/libnp_hip_rlold.so ) > $O/hmm_ab.log 2>&1; cut -c1-220 $O/hmm_ab.log
for l in "" $V/libnp_hip_rlold.so "" $V/libnp_hip_rlold.so; do
  ( NP_HIP_LIB=$l timeout 600 python tests/bench_eventalign.py --pool 1000 --tile 20 --steps 3 --cpu-sample 0 ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['kernel_ms_per_step'])"
done 2>&1 | tee $O/ea_ab.log
( timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -2
}

"call_$1"
