#!/bin/bash
# Round-3 GPU calls, one function per call (provenance of the gpurun tags the files under profiles/ cite: r03a ... r03t).
#   usage on the GPU box (through gpurun):  bash tools/runs_r03.sh <letter>        e.g.  gpurun -- 'bash tools/runs_r03.sh p'
# The end-of-round measurement set is profiles/collect_r03_final.sh, the counter passes profiles/collect_r03_pmc.sh.
# (function bodies are not indented: several hold here-documents and multi-line python -c strings)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R

# round-3 first GPU pass: parity tests (incl. long reads, pipelined batch binding), the folded bench line, the batch-binding
# throughput, counter passes over the shipped kernels
call_a() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
NP_VERBOSE=1 python -c "
import torch
from nanopolish_amd.api import Context
c = Context(0); print(c.info()); c.close()" > $O/probe.log 2>&1
( time timeout 900 python bench.py --steps 3 --warmup 1 ) > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
( time timeout 600 python tests/bench_batch_dropin.py ) > $O/batch_dropin.json 2> $O/batch_dropin.err; echo "rc=$?" >> $O/batch_dropin.err
PASS_TIMEOUT=120 bash profiles/collect_r03_pmc.sh r03a_pmc 2048 > $O/pmc.log 2>&1
tail -4 $O/pytest.log; cat $O/probe.log | tail -2; tail -c 1500 $O/bench.json; tail -3 $O/bench.err; cat $O/batch_dropin.json | cut -c1-600; tail -3 $O/batch_dropin.err; tail -12 $O/pmc.log | cut -c1-400
}

# round-3 second GPU pass: counter passes at a launch that fills every wave slot (8192 reads), batch-binding throughput with host phase timers
call_b() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_batch_dropin.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python tests/bench_batch_dropin.py --sizes 512,8192 ) > $O/batch_dropin.json 2> $O/batch_dropin.err; echo "rc=$?" >> $O/batch_dropin.err
PASS_TIMEOUT=240 bash profiles/collect_r03_pmc.sh r03b_pmc 8192 > $O/pmc.log 2>&1
tail -4 $O/pytest.log; cat $O/batch_dropin.json | cut -c1-1200; tail -3 $O/batch_dropin.err; cat gpurun_out/r03b_pmc/passes.log
}

# round-3 third GPU pass: all GPU tests (new: variants / eventalign batched bindings), batch-binding throughput after the host fast paths
call_c() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python tests/bench_batch_dropin.py --sizes 512,8192,32768 ) > $O/batch_dropin.json 2> $O/batch_dropin.err; echo "rc=$?" >> $O/batch_dropin.err
tail -25 $O/pytest.log; cat $O/batch_dropin.json | cut -c1-1500; tail -3 $O/batch_dropin.err
}

# round-3 fourth GPU pass: batch binding after the one-fetch-per-batch reference and with int16 ADC input
call_d() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_batch_dropin.py -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python tests/bench_batch_dropin.py --sizes 512,2048,8192,32768 ) > $O/batch_dropin.json 2> $O/batch_dropin.err; echo "rc=$?" >> $O/batch_dropin.err
tail -6 $O/pytest.log; cat $O/batch_dropin.json | cut -c1-1800; tail -3 $O/batch_dropin.err
}

# round-3: the two-reads-per-wave eventalign chain kernel against the one-read kernel: parity tests under both, timing of both
call_e() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
for v in 2 1; do
  ( NP_EA_KERNEL=$v timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q -x ) > $O/pytest_k$v.log 2>&1; echo "k$v pytest rc=$?" >> $O/pytest_k$v.log
  ( NP_EA_KERNEL=$v timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 ) > $O/ea_k$v.json 2> $O/ea_k$v.err; echo "rc=$?" >> $O/ea_k$v.err
done
for v in 2 1; do tail -4 $O/pytest_k$v.log; cut -c1-900 $O/ea_k$v.json; tail -2 $O/ea_k$v.err; done
}

# round-3: phase breakdown of the two-read chain kernel
call_f() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
( NP_EA_KERNEL=2 timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 0 ) > $O/ea_k2.json 2> $O/ea_k2.err; echo "rc=$?" >> $O/ea_k2.err
cut -c1-1200 $O/ea_k2.json; tail -2 $O/ea_k2.err
}

# round-3: chain kernels after the branch-free walk step: kernel 1 parity, kernel 2 with / without walk priority, waves per CU
call_g() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
( NP_EA_KERNEL=1 timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_k1.log 2>&1; echo "k1 pytest rc=$?" >> $O/pytest_k1.log
( NP_EA_KERNEL=2 timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_k2.log 2>&1; echo "k2 pytest rc=$?" >> $O/pytest_k2.log
for cfg in "2 0 16" "2 3 16" "2 1 16" "2 0 12" "1 0 20"; do set -- $cfg
  ( NP_EA_KERNEL=$1 NP_EA_WALK_PRIO=$2 NP_EA_WAVES_PER_CU=$3 timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 0 ) > $O/ea_$1_$2_$3.json 2> $O/ea_$1_$2_$3.err
  echo "kernel $1 prio $2 waves $3: $(grep -o '"value": [0-9.]*\|"eventalign_chain": [0-9.]*\|"backtrack": [0-9]*\|"fill": [0-9]*' $O/ea_$1_$2_$3.json | tr '\n' ' ')"
done
tail -3 $O/pytest_k1.log; tail -3 $O/pytest_k2.log
}

# round-3: two-read chain kernel with the dual interleaved walk: parity, then timing at 16 / 20 waves per CU, with / without priority; kernel 1 as control
call_h() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
for w in 16 20; do ( NP_EA_KERNEL=2 NP_EA_WAVES_PER_CU=$w timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_k2_$w.log 2>&1; echo "k2 w$w pytest rc=$?" >> $O/pytest_k2_$w.log; done
( NP_EA_KERNEL=1 timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_k1.log 2>&1; echo "k1 pytest rc=$?" >> $O/pytest_k1.log
for cfg in "2 1 16" "2 0 16" "2 1 20" "2 0 20" "1 0 20"; do set -- $cfg
  ( NP_EA_KERNEL=$1 NP_EA_WALK_PRIO=$2 NP_EA_WAVES_PER_CU=$3 timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 64 ) > $O/ea_$1_$2_$3.json 2> $O/ea_$1_$2_$3.err
  echo "kernel $1 prio $2 waves $3: $(grep -o '"value": [0-9.]*\|"eventalign_chain": [0-9.]*\|"backtrack": [0-9]*\|"fill": [0-9]*\|"geometry": [0-9]*\|"rows_match": [a-z]*\|"copies_identical": [a-z]*' $O/ea_$1_$2_$3.json | tr '\n' ' ')"
done
tail -2 $O/pytest_k2_16.log; tail -2 $O/pytest_k2_20.log; tail -2 $O/pytest_k1.log
}

# round-3: two-read chain kernel, tournament arg-max: parity (all eventalign tests, both kernels), timing
call_i() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_default.log 2>&1; echo "default pytest rc=$?" >> $O/pytest_default.log
for cfg in "2 0 20" "2 0 16" "1 0 20"; do set -- $cfg
  ( NP_EA_KERNEL=$1 NP_EA_WALK_PRIO=$2 NP_EA_WAVES_PER_CU=$3 timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 256 ) > $O/ea_$1_$2_$3.json 2> $O/ea_$1_$2_$3.err
  echo "kernel $1 prio $2 waves $3: $(grep -o '"value": [0-9.]*\|"eventalign_chain": [0-9.]*\|"backtrack": [0-9]*\|"fill": [0-9]*\|"geometry": [0-9]*\|"rows_match": [a-z]*\|"copies_identical": [a-z]*' $O/ea_$1_$2_$3.json | tr '\n' ' ')"
done
tail -2 $O/pytest_default.log
}

call_j() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
}

# call 15: the detector's serial path (NP_ED_SERIAL) -- events tests verbose, then the whole GPU suite
call_k() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_events.py -m gpu -q -x -s 2>&1 | tail -15 > $O/events.log; cat $O/events.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
timeout 300 python bench.py --steps 2 --warmup 1 --from-raw 1 --cpu-sample 0 --legs 0 --streamed 0 --ragged 0 > $O/bench_from_raw.json 2> $O/bench_from_raw.err; tail -c 600 $O/bench_from_raw.json
}

# call 16: the aligner as two launches -- parity, then fused / split / pipelined timings at the bench's size
call_l() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_events.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
timeout 900 python tools/split_align_bench.py --steps 4 --bt-blocks 8,4 > $O/split.jsonl 2> $O/split.err; cat $O/split.jsonl; tail -5 $O/split.err
}

# call 17: kernel timeline of the pipelined pass (which kernels of the two streams run side by side?)
call_m() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O; cd /tmp
for bt in 4; do
  timeout 600 rocprofv3 --kernel-trace -d $O/tr$bt -o t -- python $R/tools/split_align_bench.py --steps 3 --modes pipelined --bt-blocks $bt > $O/run$bt.log 2>&1
  tail -2 $O/run$bt.log | cut -c1-400
  f=$(find $O/tr$bt -name "*results.db" | head -1); [ -n "$f" ] && python3 $R/profiles/timeline_rocpd.py $f --min-ms 2 --last 40 > $O/timeline$bt.md
  rm -rf $O/tr$bt
  cat $O/timeline$bt.md | cut -c1-260
done
}

# call 18: pipelined pass, wave priorities of the back-track launch and of the forward kernels
call_n() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
timeout 900 python tools/split_align_bench.py --steps 4 --modes pipelined --bt-blocks 4,8 --prios 0:0,0:2,1:2 > $O/split.jsonl 2> $O/split.err; cat $O/split.jsonl; tail -3 $O/split.err
}

# call 19: kernel B ablations -- emissions for free (upper bound of what a fused meth/unmeth pass could share), log-sums without the table
call_o() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
timeout 900 python tools/hmm_ab.py nanopolish_amd/variants/libnp_hip_hmm_base.so nanopolish_amd/variants/libnp_hip_hmm_noem.so nanopolish_amd/variants/libnp_hip_hmm_nolse.so > $O/hmm_ab.jsonl 2> $O/hmm_ab.err; cat $O/hmm_ab.jsonl; tail -3 $O/hmm_ab.err
}

# call 20: 14-instruction walk step -- parity suite, then fused / split timings
call_p() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
timeout 900 python tools/split_align_bench.py --steps 4 --modes fused,split > $O/split.jsonl 2> $O/split.err; cat $O/split.jsonl; tail -3 $O/split.err
}

# call 21: instruction counters of the shipped chain kernel (two reads per wave) and of the one-read kernel, same batch
call_q() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O; cd /tmp
W="python $R/tools/pmc_workload.py --reads 0 --ea-reads 8192 --reps 2"
timeout 120 $W > $O/units.json 2> $O/units.err; echo "units rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq1 -o sq1 -- $W > $O/sq1.log 2>&1; echo "sq1 rc=$?"
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("$O/sq1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    print(k, len(n[k]), {c: "%.4g" % (x / len(n[k])) for c, x in v.items()})
PY
cat $O/units.json
}

# call 22: work items on the side stream beside the aligner (cm_async)
call_r() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_jobs.py tests/test_gpu_parity.py tests/test_gpu_reflevel.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --steps 4 --warmup 1 --cpu-sample 64 --ragged 0 --legs 0 > $O/bench.json 2> $O/bench.err; python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d.get("value_streamed"), d["streamed"].get("results_equal_resident"), d.get("max_abs_dLLR_vs_cpu"))
PY
tail -3 $O/bench.err
}

# call 23: A/B on one box: work items in order vs beside the aligner
call_s() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; mkdir -p $O; cd $R
for a in 0 1 0 1; do
NP_CM_ASYNC=$a timeout 600 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0 > $O/bench$a.json 2> $O/bench$a.err; python3 - <<PY
import json
d=json.loads(open("$O/bench$a.json").read().strip().splitlines()[-1])
print($a, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
PY
done
NP_CM_ASYNC=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0 --from-raw 1 > $O/braw1.json 2> $O/braw1.err
NP_CM_ASYNC=0 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0 --from-raw 1 > $O/braw0.json 2> $O/braw0.err
for a in 0 1; do python3 - <<PY
import json
d=json.loads(open("$O/braw$a.json").read().strip().splitlines()[-1])
print("raw", $a, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
PY
done
}

# call 24: lane-per-read recalibration -- parity, then A/B on one box
call_t() {
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reflevel.py tests/test_gpu_batch_dropin.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
for a in 1000000000 8192 1000000000 8192; do
NP_RECAL_LANES_MIN=$a timeout 600 python bench.py --steps 5 --warmup 1 --cpu-sample 64 --streamed 0 --ragged 1 --legs 0 > $O/bench$a.json 2> $O/bench$a.err; python3 - <<PY
import json
d=json.loads(open("$O/bench$a.json").read().strip().splitlines()[-1])
print($a, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["value_ragged"], d["ragged"]["check"], d["max_abs_dLLR_vs_cpu"])
PY
done
}

# call 25: soak against the reference itself -- indel / clipped records through the whole chain, both chain kernels
call_u() {
O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
for cfg in "1 2" "2 2" "3 1"; do set -- $cfg
timeout 900 python tests/gpu_soak.py --reads 1500 --seed $1 --ea-kernel $2 >> $O/soak.jsonl 2>> $O/soak.err; echo "rc=$?"
done
cat $O/soak.jsonl; tail -3 $O/soak.err
}

# call 31: three-lane size class of the forward kernel -- all GPU tests, then the default line with its legs
call_v() {
O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
timeout 900 python bench.py --steps 4 --warmup 1 --cpu-sample 64 --streamed 0 --ragged 0 --legs 1 > $O/bench.json 2> $O/bench.err
python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["max_abs_dLLR_vs_cpu"])
print("ea", d["value_eventalign"], "var", d["value_variants"], d["variants"]["ms_per_step"], d["variants"]["cpu_baseline"])
PY
tail -2 $O/bench.err
}

"call_$1"
