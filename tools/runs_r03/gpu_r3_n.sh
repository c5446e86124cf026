#!/bin/bash
# call 18: pipelined pass, wave priorities of the back-track launch and of the forward kernels
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
timeout 900 python tools/split_align_bench.py --steps 4 --modes pipelined --bt-blocks 4,8 --prios 0:0,0:2,1:2 > $O/split.jsonl 2> $O/split.err; cat $O/split.jsonl; tail -3 $O/split.err
