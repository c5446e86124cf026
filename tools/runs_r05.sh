#!/bin/bash
# Round-5 GPU calls, one function per call (provenance of the gpurun tags the files under profiles/ cite: r05a ...).
#   usage on the GPU box (through gpurun):  bash tools/runs_r05.sh <letter>        e.g.  gpurun -- 'bash tools/runs_r05.sh a'
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R

# first pass: the whole GPU suite after the pruning (with the new lse_oor / all-generic fallback tests), then configs[4]'s per-rank shape
# rehearsed on the one GPU of the lease (250 000 reads per step) against configs[1] (100 000) on the same box
call_a() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05a; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python bench.py --gpus 1 --pool 50000 --tile 5 --steps 5 --warmup 2 --legs 0 --streamed 0 ) > $O/bench_250k.json 2> $O/bench_250k.err; echo "rc=$?" >> $O/bench_250k.err
( time timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --legs 0 --streamed 0 --ragged 0 ) > $O/bench_100k.json 2> $O/bench_100k.err; echo "rc=$?" >> $O/bench_100k.err
tail -6 $O/pytest.log; tail -c 1500 $O/bench_250k.json; tail -4 $O/bench_250k.err; head -c 700 $O/bench_100k.json; tail -4 $O/bench_100k.err
}

# glue rewrite (multi-read recalibration, map without an init pass, bounds per pair of work items): parity suite, then the step's kernels by name
call_b() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-b}; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/bench.py --steps 3 --warmup 1 --legs 0 --streamed 0 --ragged 0 --cpu-sample 0 > $R/$O/bench.json 2> $R/$O/bench.err )
python profiles/summarize_rocpd.py $(find $O/prof -name "*_results.db" | head -1) > $O/kernels.txt 2>&1
tail -5 $O/pytest.log; head -40 $O/kernels.txt; head -c 400 $O/bench.json; python - <<EOF
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
EOF
}

# recalibration with the model in LDS: workgroup shapes A/B (16 waves x 2 reads, 8 x 4, 12 x 3), then the step's kernels by name
call_c() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-c}; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( timeout 600 python tools/hmm_ab.py --pool 4000 --tile 10 "@NP_RECAL_SHAPE=0" "@NP_RECAL_SHAPE=1" "@NP_RECAL_SHAPE=2" "@NP_RECAL_SHAPE=0" ) > $O/recal_ab.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/bench.py --steps 3 --warmup 1 --legs 0 --streamed 0 --ragged 0 --cpu-sample 0 > $R/$O/bench.json 2> $R/$O/bench.err )
python profiles/summarize_rocpd.py $(find $O/prof -name "*_results.db" | head -1) > $O/kernels.txt 2>&1
tail -5 $O/pytest.log; cat $O/recal_ab.log; grep -v "at::native\|rocclr\|probe\|np_align_\|hmm_forward" $O/kernels.txt | head -16; python - <<EOF
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
EOF
}

# coalesced device passes in the batched binding; the driver's bench command with the from-raw and binding legs folded in
call_e() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-e}; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 1200 python bench.py --steps ${STEPS:-10} --warmup 3 ) > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -5 $O/pytest.log; tail -5 $O/bench.err; python - <<EOF
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"])
for k in ("value_streamed","value_ragged","value_from_raw","value_eventalign","value_variants","value_binding_512","value_binding_8192"): print(k, d.get(k))
print(json.dumps(d.get("from_raw"))[:900]); print(json.dumps(d.get("binding"))[:2500])
EOF
}

# the binding's packer policy (gather waiting batches while the GPU is busy): binding tests, then reads/s at 512 / 2 048 / 8 192 records
# with the default policy and with NP_BATCH_COALESCE=1 (every batch a pass of its own: round 4's behaviour)
call_f() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-f}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_batch_dropin.py tests/test_gpu_sanitizers.py tests/test_gpu_dropin.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
SKIP=pipelined,pipelined_adc_ref_writer,pipelined_adc_4ctx,sync,pipelined_adc_2ctx
( timeout 600 python tests/bench_batch_dropin.py --sizes 512,2048,8192 --skip $SKIP ) > $O/binding.json 2> $O/binding.err
( NP_BATCH_COALESCE=1 timeout 600 python tests/bench_batch_dropin.py --sizes 512,2048 --skip $SKIP ) > $O/binding_nocoalesce.json 2>> $O/binding.err
( NP_BATCH_CONTEXTS=1 timeout 600 python tests/bench_batch_dropin.py --sizes 512,2048 --skip $SKIP ) > $O/binding_1ctx.json 2>> $O/binding.err
tail -5 $O/pytest.log; tail -3 $O/binding.err
python - <<EOF
import json
for f in ("binding","binding_nocoalesce","binding_1ctx"):
    for l in open("$O/%s.json" % f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["batch_size"], d["pipelined_adc"]["value"], d["pipelined_adc"]["ms_per_batch"], d["pipelined_adc"]["host_ms_per_batch"])
EOF
}

# host-constants mode (the libm repair) + the binding with all buffers sized at the first merged pass: the whole suite, then the binding's rates
call_h() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-h}; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
SKIP=pipelined,pipelined_adc_ref_writer,pipelined_adc_4ctx,sync,pipelined_adc_2ctx
( timeout 600 python tests/bench_batch_dropin.py --sizes 512,2048,8192 --skip $SKIP ) > $O/binding.json 2> $O/binding.err
( NP_BATCH_COALESCE=1 timeout 600 python tests/bench_batch_dropin.py --sizes 512,8192 --skip $SKIP ) > $O/binding_nocoalesce.json 2>> $O/binding.err
( timeout 600 python tests/bench_batch_dropin.py --sizes 512 --target-reads 524288 --skip $SKIP ) > $O/binding_long.json 2>> $O/binding.err
tail -7 $O/pytest.log; tail -3 $O/binding.err
python - <<EOF
import json
for f in ("binding","binding_nocoalesce","binding_long"):
    for l in open("$O/%s.json" % f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["batch_size"], d["batches"], d["pipelined_adc"]["value"], d["pipelined_adc"]["ms_per_batch"], d["pipelined_adc"]["host_ms_per_batch"])
EOF
}

# the libm tests again (stderr of the child visible), then the round's counter passes (kernels A, B, chain, glue, detector) at 8 192 reads
call_i() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-i}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_libm.py tests/test_gpu_batch_dropin.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
PASS_TIMEOUT=240 bash profiles/collect_r05_pmc.sh r05pmc 8192 2>&1 | tail -14
}

# detector arithmetic (three-sample sums, unwrapped sqrt / division): the suite, then the from-raw step's kernels
call_k() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-k}; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( timeout 600 python bench.py --from-raw 1 --pool 4000 --tile 25 --steps 4 --warmup 1 --legs 0 --streamed 0 --ragged 0 --cpu-sample 64 ) > $O/bench_raw.json 2> $O/bench_raw.err
( timeout 600 python tests/gpu_soak.py ) > $O/soak.log 2>&1
tail -5 $O/pytest.log; tail -4 $O/soak.log; python - <<EOF
import json
d=json.loads(open("$O/bench_raw.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["cpu_baseline"]["check"], d["cpu_baseline"].get("whole_function"))
EOF
}

# the round's counter passes on the final code: all families at 8 192 reads per launch; kernels A / B / glue at the benched 100 000
call_p() {
cd "$GRAFT_REPO_ROOT" || exit 1
PASS_TIMEOUT=240 bash profiles/collect_r05_pmc.sh r05pmc 8192 2>&1 | tail -12
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pmc100k; mkdir -p $O
W="python $R/tools/pmc_workload.py --reads 100000 --ea-reads 0 --distinct 2000 --reps 1"
( cd /tmp && timeout 300 $W --timing-reps 3 > $O/units.json 2> $O/units.err ); echo "units rc=$?"
for p in "sq1:SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  n=${p%%:*}; c=${p#*:}
  ( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$n -o $n -- $W > $O/$n.log 2>&1 ); echo "$n rc=$?" | tee -a $O/passes.log
done
tail -1 $O/units.json | cut -c1-400
}

# detector ablations (results wrong by construction): no sample loads / every lane the same cached segment / no event-list stores
call_q() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-q}; mkdir -p $O
for v in default edabl1 edabl2 edabl4 edabl7; do
  L=""; [ $v != default ] && L=$PWD/nanopolish_amd/variants/libnp_hip_$v.so
  ( NP_HIP_LIB=$L timeout 300 python bench.py --from-raw 1 --pool 2000 --tile 50 --steps 3 --warmup 1 --legs 0 --streamed 0 --ragged 0 --cpu-sample 0 ) > $O/$v.json 2> $O/$v.err
  python - <<EOF
import json
try:
    d=json.loads([l for l in open("$O/$v.json") if l.startswith("{")][-1]); print("$v", d["roofline"]["kernel_ms_per_step"])
except Exception as e: print("$v", "failed", e)
EOF
done
}

# lean check: the suite, the default step's kernels by name, the from-raw step's families
call_r() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-r}; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/bench.py --steps 3 --warmup 1 --legs 0 --streamed 0 --ragged 0 --cpu-sample 0 > $R/$O/bench.json 2> $R/$O/bench.err )
python profiles/summarize_rocpd.py $(find $O/prof -name "*_results.db" | head -1) > $O/kernels.txt 2>&1
( timeout 400 python bench.py --from-raw 1 --pool 2000 --tile 50 --steps 3 --warmup 1 --legs 0 --streamed 0 --ragged 0 --cpu-sample 0 ) > $O/bench_raw.json 2> $O/bench_raw.err
tail -4 $O/pytest.log; grep -v "at::native\|rocclr\|probe\|np_align_\|hmm_forward\|np_event_align" $O/kernels.txt | head -12 | cut -c1-160; python - <<EOF
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
d=json.loads(open("$O/bench_raw.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
EOF
}

# chain kernel check: the event-align suites, the eventalign leg's time with its kernels by name, the soak
call_s() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-s}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_eventalign_dropin.py tests/test_gpu_dropin.py tests/test_gpu_reflevel.py tests/test_gpu_rna.py tests/test_gpu_sites.py tests/test_gpu_fuzz.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/bench.py --workload eventalign --steps 3 --warmup 1 --cpu-sample 0 > $R/$O/bench_ea.json 2> $R/$O/bench_ea.err )
python profiles/summarize_rocpd.py $(find $O/prof -name "*_results.db" | head -1) > $O/kernels.txt 2>&1
( timeout 400 python tests/gpu_soak_eventalign.py ) > $O/soak.log 2>&1; echo "soak rc=$?" >> $O/soak.log
( NP_HIP_LIB=$PWD/nanopolish_amd/variants/libnp_hip_smalllist.so timeout 600 python -m pytest tests/test_gpu_eventalign_dropin.py tests/test_gpu_rna.py -m gpu -x -q ) > $O/pytest_smalllist.log 2>&1; echo "smalllist rc=$?" >> $O/pytest_smalllist.log; tail -2 $O/pytest_smalllist.log
tail -4 $O/pytest.log; grep -v "at::native\|rocclr\|probe" $O/kernels.txt | head -8 | cut -c1-160; tail -3 $O/soak.log; tail -1 $O/bench_ea.json | cut -c1-600
}

# chain kernel variants (NP_HIP_LIB): the eventalign leg's chain time for each
call_t() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-t}; mkdir -p $O
for v in default ${VARIANTS:-early3 early5 early7}; do
  L=""; [ $v != default ] && L=$PWD/nanopolish_amd/variants/libnp_hip_$v.so
  ( NP_HIP_LIB=$L timeout 300 python bench.py --workload eventalign --steps 3 --warmup 1 --cpu-sample 0 ) > $O/$v.json 2> $O/$v.err
  python -c "
import json
try:
    d=json.loads([l for l in open('$O/$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['kernel_ms_per_step']['eventalign_chain'], d['roofline'].get('wave_cycles_by_phase'))
except Exception as e: print('$v', 'failed', e)
"
done
}

# chain kernel: parity (event-align suites, the always-spilling variant), then the variants' A/B on this box
call_u() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-u}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_eventalign_dropin.py tests/test_gpu_chain_spill.py tests/test_gpu_rna.py tests/test_gpu_fuzz.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( timeout 400 python tests/gpu_soak_eventalign.py ) > $O/soak.log 2>&1; echo "soak rc=$?" >> $O/soak.log
tail -3 $O/pytest.log; tail -2 $O/soak.log
TAG=${TAG:-u} call_t
}

# work-item builder check: the suites that pin work items and sites, then the default step's kernels by name
call_v() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-v}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_jobs.py tests/test_gpu_sites.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_batch_dropin.py tests/test_gpu_long_reads.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/bench.py --steps 3 --warmup 1 --legs 0 --streamed 0 --ragged 0 --cpu-sample 0 > $R/$O/bench.json 2> $R/$O/bench.err )
python profiles/summarize_rocpd.py $(find $O/prof -name "*_results.db" | head -1) > $O/kernels.txt 2>&1
tail -4 $O/pytest.log; grep -v "at::native\|rocclr\|probe\|np_align_\|hmm_forward\|np_event_align" $O/kernels.txt | head -12 | cut -c1-160
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])
"
}

# soak on the round's final code: new seeds through the whole chain against the reference itself, then 512 full-size reads of eventalign rows
call_w() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-soak2}; mkdir -p $O
for seed in ${SEEDS:-1 31 32 33}; do ( timeout 300 python tests/gpu_soak.py --seed $seed ) >> $O/soak.log 2>&1; done
( timeout 400 python tests/gpu_soak_eventalign.py 512 ) > $O/soak_ea.log 2>&1; echo "soak_ea rc=$?" >> $O/soak_ea.log
( NP_EA_WALK_PRIO=1 timeout 300 python bench.py --workload eventalign --steps 3 --warmup 1 --cpu-sample 0 ) > $O/ea_prio.json 2> $O/ea_prio.err
( timeout 300 python bench.py --workload eventalign --steps 3 --warmup 1 --cpu-sample 0 ) > $O/ea_noprio.json 2> $O/ea_noprio.err
grep "^{" $O/soak.log | cut -c1-230; tail -2 $O/soak_ea.log
python -c "
import json
for n in ('ea_prio','ea_noprio'):
    d=json.loads([l for l in open('$O/%s.json' % n) if l.startswith('{')][-1]); print(n, d['value'], d['kernel_ms_per_step']['eventalign_chain'])
"
}

# detector check: the event suites and the binding (int16 in), then the from-raw step's kernels by name
call_x() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-x}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_events.py tests/test_gpu_batch_dropin.py tests/test_gpu_rna.py tests/test_gpu_eventalign_dropin.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/bench.py --from-raw 1 --pool 4000 --tile 25 --steps 3 --warmup 1 --legs 0 --streamed 0 --ragged 0 --cpu-sample 64 > $R/$O/bench_raw.json 2> $R/$O/bench_raw.err )
python profiles/summarize_rocpd.py $(find $O/prof -name "*_results.db" | head -1) > $O/kernels.txt 2>&1
tail -12 $O/pytest.log; grep "np_ed_\|adc_to_pa\|mom_fill" $O/kernels.txt | cut -c1-150
python -c "
import json
d=json.loads(open('$O/bench_raw.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['cpu_baseline'].get('whole_function') or d['cpu_baseline'].get('check'))
"
}

# from-raw step with library variants (NP_HIP_LIB): the detector family's time for each
call_y() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05${TAG:-y}; mkdir -p $O
for v in default ${VARIANTS:-evwin3072 evwin4096 evwin9216 evwin12288}; do
  L=""; [ $v != default ] && L=$PWD/nanopolish_amd/variants/libnp_hip_$v.so
  ( NP_HIP_LIB=$L timeout 300 python bench.py --from-raw 1 --pool 2000 --tile 50 --steps 3 --warmup 1 --legs 0 --streamed 0 --ragged 0 --cpu-sample 0 ) > $O/$v.json 2> $O/$v.err
  python -c "
import json
try:
    d=json.loads([l for l in open('$O/$v.json') if l.startswith('{')][-1]); print('$v', d['value'], d['roofline']['kernel_ms_per_step'])
except Exception as e: print('$v', 'failed', e)
"
done
}

"call_$1"
