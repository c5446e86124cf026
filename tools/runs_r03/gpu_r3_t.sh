#!/bin/bash
# call 24: lane-per-read recalibration -- parity, then A/B on one box
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reflevel.py tests/test_gpu_batch_dropin.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
for a in 1000000000 8192 1000000000 8192; do
NP_RECAL_LANES_MIN=$a timeout 600 python bench.py --steps 5 --warmup 1 --cpu-sample 64 --streamed 0 --ragged 1 --legs 0 > $O/bench$a.json 2> $O/bench$a.err; python3 - <<PY
import json
d=json.loads(open("$O/bench$a.json").read().strip().splitlines()[-1])
print($a, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["value_ragged"], d["ragged"]["check"], d["max_abs_dLLR_vs_cpu"])
PY
done
