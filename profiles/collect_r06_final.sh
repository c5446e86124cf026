#!/bin/bash
# Round-6 end-of-round measurement set (run on the GPU box through gpurun):  bash profiles/collect_r06_final.sh [tag]
# GPU tests (incl. the sanitizer builds and the skewed-libm child processes), the DRIVER's bench command (configs[1] + the from-raw,
# eventalign, variants and binding legs in one line), configs[4] (genome-placed reads, genome-keyed site table) rehearsed on the one GPU
# (250 000 reads per step) and through the N > 1 launcher over gloo (2 and 8 ranks), the configs[0] plumbing line, the binding at 512 ... 32 768 records per batch with batches in pieces and whole, the per-call shim, kernel traces of the default step, the from-raw step and the two legs.
# Counter passes: profiles/collect_r06_pmc.sh.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06fin}; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
NP_VERBOSE=1 python -c "
import torch
from nanopolish_amd.api import Context
c = Context(0); print(c.info()); c.close()" > $O/probe.log 2>&1
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
# configs[4]: reads placed on the 5 Mb genome, the per-site table keyed by genome position -- one rank at 250 000 reads per step, and the N > 1 launcher
# (2 and 8 ranks over gloo on the one device, per-rank parity on)
timeout 1200 python bench.py --gpus 1 --genome 1 --pool 50000 --tile 5 --steps 5 --warmup 2 > $O/bench_250k.json 2> $O/bench_250k.err
NP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --pool 2000 --tile 5 --steps 2 --warmup 1 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
NP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --pool 1000 --tile 5 --steps 2 --warmup 1 > $O/bench_8rank_gloo.json 2> $O/bench_8rank_gloo.err
timeout 300 python bench.py --workload cpu-t1 --cpu-sample 200 > $O/bench_cpu_t1.json 2> $O/bench_cpu_t1.err
SKIP=pipelined,pipelined_adc_ref_writer,pipelined_adc_4ctx,pipelined_adc_2ctx
timeout 900 python tests/bench_batch_dropin.py --sizes 512,2048,8192,32768 --target-reads 262144 --skip $SKIP > $O/batch_dropin.json 2> $O/batch_dropin.err
NP_BATCH_PIECE=1000000 timeout 900 python tests/bench_batch_dropin.py --sizes 2048,8192,32768 --target-reads 262144 --skip $SKIP,sync > $O/batch_dropin_whole.json 2>> $O/batch_dropin.err
timeout 600 python tests/bench_percall_dropin.py > $O/percall.json 2> $O/percall.err
for w in "default:--steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0" "from_raw:--from-raw 1 --pool 4000 --tile 25 --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0" "eventalign:--workload eventalign --steps 3 --warmup 1 --cpu-sample 0" "variants:--workload variants --steps 3 --warmup 1 --cpu-sample 0"; do
  n=${w%%:*}; a=${w#*:}
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace_$n -o t -- python $R/bench.py $a > $O/trace_$n.log 2>&1 )
  f=$(find $O/trace_$n -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace_$n.md
  rm -rf $O/trace_$n
done
tail -3 $O/pytest.log; for f in default 250k 8rank_gloo cpu_t1; do tail -c 300 $O/bench_$f.json; echo; tail -2 $O/bench_$f.err; done; head -8 $O/trace_default.md | cut -c1-170
