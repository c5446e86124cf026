#!/bin/bash
# call 20: 14-instruction walk step -- parity suite, then fused / split timings
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
timeout 900 python tools/split_align_bench.py --steps 4 --modes fused,split > $O/split.jsonl 2> $O/split.err; cat $O/split.jsonl; tail -3 $O/split.err
