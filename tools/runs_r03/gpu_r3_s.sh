#!/bin/bash
# call 23: A/B on one box: work items in order vs beside the aligner
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
