#!/bin/bash
# Round-4 end-of-round measurement set (run on the GPU box through gpurun):  bash profiles/collect_r04_final.sh [tag]
# GPU tests (incl. the sanitizer builds), the folded default bench line (configs[1] + eventalign + variants legs), the from-raw line, the
# configs[0] plumbing line, the 8-rank gloo rehearsal of the N > 1 line on one device, the reference-side bindings (batched: one and two
# contexts; per call under OpenMP), kernel traces of the default command and of the two legs.  Counter passes: profiles/collect_r04_pmc.sh.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04fin}; mkdir -p $O; cd $R
( time timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
NP_VERBOSE=1 python -c "
import torch
from nanopolish_amd.api import Context
c = Context(0); print(c.info()); c.close()" > $O/probe.log 2>&1
timeout 1200 python bench.py --steps 5 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 3 --warmup 1 --from-raw 1 --cpu-sample 256 --legs 0 > $O/bench_from_raw.json 2> $O/bench_from_raw.err
timeout 300 python bench.py --workload cpu-t1 --cpu-sample 200 > $O/bench_cpu_t1.json 2> $O/bench_cpu_t1.err
NP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --pool 1000 --tile 5 --steps 2 --warmup 1 --legs 0 > $O/bench_8rank_gloo.json 2> $O/bench_8rank_gloo.err
timeout 900 python tests/bench_batch_dropin.py --sizes 512,8192 --skip pipelined_adc_ref_writer,pipelined_adc_4ctx > $O/batch_dropin.json 2> $O/batch_dropin.err
timeout 600 python tests/bench_percall_dropin.py > $O/percall.json 2> $O/percall.err
# kernel traces: (a) the default step alone (kernel A's average launch time must agree with roofline.avg_launch_ms of the line above),
# (b) the eventalign leg, (c) the variants leg
for w in "default:--steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0" "eventalign:--workload eventalign --steps 3 --warmup 1 --cpu-sample 0" "variants:--workload variants --steps 3 --warmup 1 --cpu-sample 0"; do
  n=${w%%:*}; a=${w#*:}
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace_$n -o t -- python $R/bench.py $a > $O/trace_$n.log 2>&1 )
  f=$(find $O/trace_$n -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace_$n.md
  rm -rf $O/trace_$n
done
tail -3 $O/pytest.log; for f in default from_raw cpu_t1 8rank_gloo; do tail -c 300 $O/bench_$f.json; echo; tail -2 $O/bench_$f.err; done; head -8 $O/trace_default.md | cut -c1-170
