#!/bin/bash
# Round-6 GPU calls, one function per call (provenance of the gpurun tags the files under profiles/ cite: r06a ...).
#   usage on the GPU box (through gpurun):  bash tools/runs_r06.sh <letter>        e.g.  gpurun -- 'bash tools/runs_r06.sh b'
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
V=nanopolish_amd/variants

# r06a was: ./tools/mfma_f64_probe (the matrix pipe as kernel A's fp64 adder: layout, exactness, rates)

# the GPU suite after the C-ABI changes (explicit ADC verdicts, host scoring vs a declared layout), the probe with the scalar-store trace
# mode, v_readlane's lane-select forms, and kernel A with one band-end lane select in M0 against the shipped build (A/B/A/B, same box)
call_b() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
./tools/mfma_f64_probe > $O/probe.log 2>&1
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/valu_rates.hip -o /tmp/valu_rates > /dev/null 2>&1 && /tmp/valu_rates l > $O/valu_l.log 2>&1
( timeout 900 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_base.so $V/libnp_hip_m0sel.so $V/libnp_hip_base.so $V/libnp_hip_m0sel.so ) > $O/align_ab.log 2>&1
( time timeout 600 python bench.py --steps 5 --warmup 2 --legs 0 --streamed 0 --ragged 0 --cpu-sample 0 ) > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.log; grep "waves/SIMD 8" $O/probe.log; cat $O/valu_l.log; cat $O/align_ab.log; head -c 600 $O/bench.json; tail -3 $O/bench.err
}

show_line() {      # rank 0's JSON line of a bench run: value, site table, shard check, per-rank rates
python - "$1" <<'PYEOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["n_gpus"], d["config"]["workload"][:100]); print(d.get("site_table")); print(d.get("shard_check"))
    print([(p["rank"], p["value"], p["table_and_allreduce_ms"]) for p in d["per_rank"]]); print(d["roofline"]["kernel_ms_per_step"])
except Exception as e:
    print("no line:", e)
PYEOF
}

# the genome-keyed N > 1 line: the new GPU tests, then bench.py in genome mode -- one rank small; 2 and 8 ranks over gloo on the
# one device of the lease (NP_BENCH_BACKEND=gloo: the N > 1 code path with host-side collectives); one rank at configs[4]'s 250 000 reads
call_d() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06d; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sites.py tests/test_gpu_events.py tests/test_gpu_jobs.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python bench.py --gpus 1 --genome 1 --pool 2000 --tile 5 --steps 3 --warmup 1 ) > $O/genome_n1_small.json 2> $O/genome_n1_small.err; echo "rc=$?" >> $O/genome_n1_small.err
( time NP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --pool 2000 --tile 5 --steps 3 --warmup 1 ) > $O/genome_gloo2.json 2> $O/genome_gloo2.err; echo "rc=$?" >> $O/genome_gloo2.err
( time NP_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --pool 1000 --tile 5 --steps 3 --warmup 1 --cpu-sample 0 ) > $O/genome_gloo8.json 2> $O/genome_gloo8.err; echo "rc=$?" >> $O/genome_gloo8.err
( time timeout 1500 python bench.py --gpus 1 --genome 1 --pool 50000 --tile 5 --steps 3 --warmup 1 ) > $O/genome_n1_250k.json 2> $O/genome_n1_250k.err; echo "rc=$?" >> $O/genome_n1_250k.err
tail -5 $O/pytest.log
for f in genome_n1_small genome_gloo2 genome_gloo8 genome_n1_250k; do echo "== $f"; tail -4 $O/$f.err; show_line $O/$f.json; done
}

# the half-wave recalibration kernel (recal_shape 3: 16 waves x 4 reads, four waves per SIMD): bit-equality across shapes, then the glue family A/B/A/B
call_e() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sites.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( timeout 900 python tools/hmm_ab.py --pool 4000 --tile 25 "@NP_RECAL_SHAPE=0" "@NP_RECAL_SHAPE=3" "@NP_RECAL_SHAPE=0" "@NP_RECAL_SHAPE=3" ) > $O/recal_ab.log 2>&1
tail -5 $O/pytest.log; cat $O/recal_ab.log
}

# the whole GPU suite on the round's code so far, then the reference-side batched binding after the packer's stretch cache (512 / 8 192 records)
call_f() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06${TAG:-f}; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 900 python tests/bench_batch_dropin.py --sizes 512,8192 --target-reads 262144 --skip pipelined,pipelined_adc_ref_writer,pipelined_adc_2ctx,sync ) > $O/binding.log 2>&1
tail -5 $O/pytest.log; grep "^{" $O/binding.log | cut -c1-1200
}

# the batched binding, same box: default (48 slots, two contexts) against 24 slots / one context / other pool sizes, at 512 and 8 192 records
call_g() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06g; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python tests/bench_batch_dropin.py --target-reads 262144 --skip pipelined,pipelined_adc_ref_writer,pipelined_adc_2ctx,sync"
for cfg in "default:" "slots24:NP_BATCH_SLOTS=24" "ctx1:NP_BATCH_CONTEXTS=1" "threads16:NP_HOST_THREADS=16" "threads32:NP_HOST_THREADS=32" "default2:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( env $envs timeout 600 $B --sizes 512,8192 ) > $O/binding_$name.log 2>&1
done
tail -5 $O/pytest.log
for f in $O/binding_*.log; do echo "== $f"; grep "^{" $f | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); p = d['pipelined_adc']
    print(d['batch_size'], p['value'], p['ms_per_batch'], {k: v for k, v in p['host_ms_per_batch'].items() if k in ('phase1a_fetch_sizes','phase1b_pack','finisher_wait_device','phase3_maps','collect_wait')})
"; done
}

# batches in pieces (NP_BATCH_PIECE): the binding's tests, then 512 / 2 048 / 8 192 / 32 768 records per batch on one box, pieces on and off
call_h() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06h; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_batch_dropin.py tests/test_gpu_sanitizers.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python tests/bench_batch_dropin.py --target-reads 262144 --skip pipelined,pipelined_adc_ref_writer,pipelined_adc_2ctx,sync"
for cfg in "pieces:" "whole:NP_BATCH_PIECE=1000000" "pieces2:" "pieces512:NP_BATCH_PIECE=512" "pieces2048:NP_BATCH_PIECE=2048"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( env $envs timeout 600 $B --sizes 512,2048,8192,32768 ) > $O/binding_$name.log 2>&1
done
tail -5 $O/pytest.log
for f in $O/binding_*.log; do echo "== $f"; grep "^{" $f | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); p = d['pipelined_adc']
    print(d['batch_size'], p['value'], p['ms_per_batch'], {k: v for k, v in p['host_ms_per_batch'].items() if k in ('phase1a_fetch_sizes','phase1b_pack','finisher_wait_device','phase3_maps','collect_wait')})
"; done
}

# pieces of 512 by default (at most a third of the slots per batch): the binding's tests, the four batch sizes twice; then the detector's
# two-block ring loop against the shipped one (from-raw step, A/B/A/B) and the whole suite on the variant
call_i() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06i; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_batch_dropin.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
B="python tests/bench_batch_dropin.py --target-reads 262144 --skip pipelined,pipelined_adc_ref_writer,pipelined_adc_2ctx,sync"
for name in a b; do ( timeout 600 $B --sizes 512,2048,8192,32768 ) > $O/binding_$name.log 2>&1; done
( timeout 900 python tools/hmm_ab.py --from-raw 1 --pool 2000 --tile 25 "" "$V/libnp_hip_edunroll.so" "" "$V/libnp_hip_edunroll.so" ) > $O/ed_ab.log 2>&1
( NP_HIP_LIB=$PWD/$V/libnp_hip_edunroll.so timeout 900 python -m pytest tests/test_gpu_events.py tests/test_gpu_reflevel.py -m gpu -x -q ) > $O/pytest_edunroll.log 2>&1; echo "pytest rc=$?" >> $O/pytest_edunroll.log
tail -4 $O/pytest.log; tail -3 $O/pytest_edunroll.log; cat $O/ed_ab.log
for f in $O/binding_*.log; do echo "== $f"; grep "^{" $f | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); p = d['pipelined_adc']
    print(d['batch_size'], p['value'], p['ms_per_batch'], {k: v for k, v in p['host_ms_per_batch'].items() if k in ('phase1a_fetch_sizes','phase1b_pack','finisher_wait_device','collect_wait')})
"; done
}

# soak against the reference itself on the round's kernels (the half-wave recalibration is on that chain), and the genome-placed batch's
# parity on 300 reads (the oracle's restatement of the per-read pass on the same BAM record)
call_m() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06m; mkdir -p $O
for seed in 31 32 33; do ( time timeout 900 python tests/gpu_soak.py --reads 1500 --seed $seed ) > $O/soak_$seed.log 2>&1; tail -4 $O/soak_$seed.log | head -1 | cut -c1-400; done
( time timeout 900 python tests/gpu_soak_eventalign.py 512 ) > $O/soak_ea.log 2>&1; grep "^{" $O/soak_ea.log | cut -c1-400
( time timeout 1200 python bench.py --gpus 1 --genome 1 --pool 4000 --tile 5 --steps 2 --warmup 1 --parity-reads 300 ) > $O/genome_parity300.json 2> $O/genome_parity300.err; echo "rc=$?"; show_line $O/genome_parity300.json
}

# more soak on the round's final binaries: six more seeds of 1 500 records through the whole chain against the reference, 2 048 full-size reads of
# eventalign rows, and the GPU suite once more on the rebuilt artefacts
call_q() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06q; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -h "passed\|failed" $O/pytest.log
for seed in 34 35 36 37 38 39; do ( timeout 900 python tests/gpu_soak.py --reads 1500 --seed $seed ) > $O/soak_$seed.log 2>&1; grep "^{" $O/soak_$seed.log | cut -c1-330; done
( timeout 1200 python tests/gpu_soak_eventalign.py 2048 ) > $O/soak_ea.log 2>&1; tail -2 $O/soak_ea.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
}

# the genome-keyed table with one row per motif site (np_genome_site_index_dev + np_site_table_genome_indexed_dev): the GPU suite, then the
# genome line at one rank with both row layouts (table_and_allreduce_ms: the zero fill + the table kernel), the gloo rehearsal at 2 ranks,
# and the 250 000-read line
call_s() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06s; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -h "passed\|failed" $O/pytest.log; grep -B30 "^FAILED\|Error" $O/pytest.log | tail -60
for rows in site base; do
( time timeout 600 python bench.py --gpus 1 --genome 1 --site-rows $rows --pool 2000 --tile 5 --steps 3 --warmup 1 --cpu-sample 0 ) > $O/genome_n1_$rows.json 2> $O/genome_n1_$rows.err; echo "rc=$?"; show_line $O/genome_n1_$rows.json
done
( time NP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --pool 2000 --tile 5 --steps 3 --warmup 1 --cpu-sample 0 ) > $O/genome_gloo2.json 2> $O/genome_gloo2.err; echo "rc=$?"; show_line $O/genome_gloo2.json
( time timeout 1500 python bench.py --gpus 1 --genome 1 --pool 50000 --tile 5 --steps 3 --warmup 1 --cpu-sample 0 ) > $O/genome_n1_250k.json 2> $O/genome_n1_250k.err; echo "rc=$?"; show_line $O/genome_n1_250k.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
}

# kernel A with the next band's emissions threaded through the band's tail (NP_A_PIPE) against the shipped build: A/B/A/B on one box (pairs crc),
# then the aligner's own GPU tests on the pipelined build and a short headline run
call_u() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06u; mkdir -p $O
( timeout 900 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_base.so $V/libnp_hip_pipe.so $V/libnp_hip_base.so $V/libnp_hip_pipe.so ) > $O/align_ab.log 2>&1; cat $O/align_ab.log | tail -8
( timeout 900 python tools/align_ab.py --pool 2048 --tile 16 --ragged 1 $V/libnp_hip_base.so $V/libnp_hip_pipe.so ) > $O/align_ab_ragged.log 2>&1; cat $O/align_ab_ragged.log | tail -4
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_long_reads.py tests/test_gpu_fallbacks.py tests/test_gpu_edges.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
( time timeout 600 python bench.py --steps 5 --warmup 2 --legs 0 --streamed 0 --ragged 0 ) > $O/bench.json 2> $O/bench.err; head -c 700 $O/bench.json; tail -3 $O/bench.err
}

# soak on the tree as it stands after the genome site index (kernels of the headline path unchanged; the library rebuilt): six new seeds of 1 500
# records through the whole chain against the reference compiled in place, 1 024 full-size eventalign reads
call_y() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06y; mkdir -p $O
for seed in 40 41 42 43 44 45; do ( timeout 900 python tests/gpu_soak.py --reads 1500 --seed $seed ) > $O/soak_$seed.log 2>&1; grep "^{" $O/soak_$seed.log | cut -c1-330; done
( timeout 1200 python tests/gpu_soak_eventalign.py 1024 ) > $O/soak_ea.log 2>&1; tail -2 $O/soak_ea.log | cut -c1-300
}

# the detector's divisions by the window length in two operations (div_small_*): the exhaustive self-test, the detector's and the from-raw
# chain's tests, then the from-raw step A/B against the previous build is the bench line itself (event_detect in kernel_ms_per_step)
call_z() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06z; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
( timeout 900 python tests/gpu_soak.py --reads 1500 --seed 46 ) > $O/soak_46.log 2>&1; grep "^{" $O/soak_46.log | cut -c1-330
( time timeout 900 python bench.py --steps 5 --warmup 2 --streamed 0 --ragged 0 ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PYEOF2'
import json
d = json.loads(open("gpurun_out/r06z/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["kernel_ms_per_step"]); fr = d["from_raw"]; print(fr["value"], fr["kernel_ms_per_step"], fr.get("check")); print(d["value_eventalign"], d["eventalign"].get("kernel_ms_per_step"))
PYEOF2
}

# counts in, events out (np_detect_events_adc_dev) against the two calls, same box, alternating; then the GPU suite
call_aa() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06aa; mkdir -p $O
( timeout 900 python tools/detect_ab.py ) > $O/detect_ab.log 2>&1; grep "^rep" $O/detect_ab.log
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -h "passed\|failed" $O/pytest.log; grep "^FAILED" $O/pytest.log
}

# the emission's division in four operations (np_div_exact2) in kernel A and the chain: A/B of kernel A, the GPU suite (the division self-test with
# its enumeration of the edge divisors), a soak seed, the bench line
call_ad() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06ad; mkdir -p $O
( timeout 900 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_base.so $V/libnp_hip_div2.so $V/libnp_hip_base.so $V/libnp_hip_div2.so ) > $O/align_ab.log 2>&1; cat $O/align_ab.log | tail -4
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -h "passed\|failed" $O/pytest.log; grep "^FAILED" $O/pytest.log
( timeout 900 python tests/gpu_soak.py --reads 1500 --seed 47 ) > $O/soak_47.log 2>&1; grep "^{" $O/soak_47.log | cut -c1-330
( timeout 900 python tests/gpu_soak_eventalign.py 512 ) > $O/soak_ea.log 2>&1; tail -1 $O/soak_ea.log | cut -c1-200
( time timeout 900 python bench.py --steps 10 --warmup 3 --streamed 0 --ragged 0 ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PYEOF2'
import json
d = json.loads(open("gpurun_out/r06ad/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["kernel_ms_per_step"], d["max_abs_dLLR_vs_cpu"], d["cpu_baseline"]["check"]); fr = d["from_raw"]; print(fr["value"], fr["kernel_ms_per_step"], fr.get("check")); print(d["value_eventalign"], d["eventalign"].get("kernel_ms_per_step"))
PYEOF2
}

# the filtered t-statistic ratio: the GPU suite (its self-test among them), the randomised detection comparison, two soak seeds, the bench line
call_am() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06am; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -h "passed\|failed" $O/pytest.log; grep "^FAILED\|^E  " $O/pytest.log | head
python - <<'PYEOF2'
import ctypes as C
from nanopolish_amd.api import Context
c = Context(0); bad, near, far = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
for seed in (1, 2, 3):
    c.L.np_selftest_tstat_ratio(c.h, 1 << 32, seed, C.byref(bad), C.byref(near), C.byref(far)); print("ratio selftest seed", seed, "trusted mismatches", bad.value, "sent to exact", near.value, "farthest unfiltered disagreement", far.value)
PYEOF2
( timeout 900 python tools/fuzz_detect_adc.py --batches 60 --seed 9 ) > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
for seed in 48 49; do ( timeout 900 python tests/gpu_soak.py --reads 1500 --seed $seed ) > $O/soak_$seed.log 2>&1; grep "^{" $O/soak_$seed.log | cut -c1-330; done
( time timeout 900 python bench.py --steps 10 --warmup 3 --streamed 0 --ragged 0 ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PYEOF2'
import json
d = json.loads(open("gpurun_out/r06am/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["kernel_ms_per_step"], d["max_abs_dLLR_vs_cpu"]); fr = d["from_raw"]; print(fr["value"], fr["kernel_ms_per_step"], fr.get("check")); print(d["value_eventalign"], d["eventalign"].get("kernel_ms_per_step"))
PYEOF2
}

# soak on the round's final binaries: eight seeds of 1 500 records from int16 counts (the counts-in detector), four from float samples, and
# 1 024 full-size eventalign reads, all against the reference compiled in place
call_ao() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06ao; mkdir -p $O
for seed in 50 51 52 53 54 55 56 57; do ( timeout 900 python tests/gpu_soak.py --reads 1500 --seed $seed --adc 1 ) > $O/soak_adc_$seed.log 2>&1; grep "^{" $O/soak_adc_$seed.log | cut -c1-330; done
for seed in 58 59 60 61; do ( timeout 900 python tests/gpu_soak.py --reads 1500 --seed $seed ) > $O/soak_$seed.log 2>&1; grep "^{" $O/soak_$seed.log | cut -c1-330; done
( timeout 1200 python tests/gpu_soak_eventalign.py 1024 ) > $O/soak_ea.log 2>&1; tail -1 $O/soak_ea.log | cut -c1-200
}

"call_$1"
