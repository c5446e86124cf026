#!/bin/bash
# call 16: the aligner as two launches -- parity, then fused / split / pipelined timings at the bench's size
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_events.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
timeout 900 python tools/split_align_bench.py --steps 4 --bt-blocks 8,4 > $O/split.jsonl 2> $O/split.err; cat $O/split.jsonl; tail -5 $O/split.err
