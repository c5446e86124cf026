#!/bin/bash
# Round-3 end-of-round measurement set (run on the GPU box through gpurun):  bash profiles/collect_r03_final.sh [tag]
# Tests, the folded default bench line (configs[1] + eventalign + variants legs), the from-raw line, the 2-rank rehearsal, the batch
# binding, a kernel trace of the default command.  The counter passes are profiles/collect_r03_pmc.sh.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03fin}; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
NP_VERBOSE=1 python -c "
import torch
from nanopolish_amd.api import Context
c = Context(0); print(c.info()); c.close()" > $O/probe.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 3 --warmup 1 --from-raw 1 --cpu-sample 256 --legs 0 > $O/bench_from_raw.json 2> $O/bench_from_raw.err
timeout 300 python bench.py --workload cpu-t1 --cpu-sample 200 > $O/bench_cpu_t1.json 2> $O/bench_cpu_t1.err
NP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --pool 4000 --tile 5 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
timeout 600 python tests/bench_batch_dropin.py --sizes 512,2048,8192,32768 > $O/batch_dropin.json 2> $O/batch_dropin.err
# kernel traces: (a) the default step alone (kernel A's average launch time must agree with roofline.avg_launch_ms of the line above),
# (b) the two folded legs
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0 > $O/trace.log 2>&1 )
f=$(find $O/trace -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace.md
rm -rf $O/trace
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_ea -o t -- python $R/bench.py --workload eventalign --steps 3 --warmup 1 --cpu-sample 0 > $O/trace_ea.log 2>&1 )
f=$(find $O/trace_ea -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace_eventalign.md
rm -rf $O/trace_ea
tail -3 $O/pytest.log; for f in default from_raw cpu_t1 2rank_gloo; do tail -c 400 $O/bench_$f.json; echo; tail -2 $O/bench_$f.err; done; head -12 $O/trace.md | cut -c1-170
