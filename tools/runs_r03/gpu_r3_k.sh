#!/bin/bash
# call 15: the detector's serial path (NP_ED_SERIAL) -- events tests verbose, then the whole GPU suite
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_events.py -m gpu -q -x -s 2>&1 | tail -15 > $O/events.log; cat $O/events.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
timeout 300 python bench.py --steps 2 --warmup 1 --from-raw 1 --cpu-sample 0 --legs 0 --streamed 0 --ragged 0 > $O/bench_from_raw.json 2> $O/bench_from_raw.err; tail -c 600 $O/bench_from_raw.json
