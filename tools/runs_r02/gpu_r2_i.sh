#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02i; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_jobs.py -m gpu -q 2>&1 | tail -3 ) > $O/pytest.log 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/$O/prof_streamed -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --pool 8000 --tile 5 --cpu-sample 0 --ragged 0 > $GRAFT_REPO_ROOT/$O/prof_streamed.log 2>&1
cd $GRAFT_REPO_ROOT; cat $O/pytest.log; tail -c 1500 $O/prof_streamed.log
