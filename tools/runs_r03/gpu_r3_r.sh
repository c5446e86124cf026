#!/bin/bash
# call 22: work items on the side stream beside the aligner (cm_async)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_jobs.py tests/test_gpu_parity.py tests/test_gpu_reflevel.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --steps 4 --warmup 1 --cpu-sample 64 --ragged 0 --legs 0 > $O/bench.json 2> $O/bench.err; python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d.get("value_streamed"), d["streamed"].get("results_equal_resident"), d.get("max_abs_dLLR_vs_cpu"))
PY
tail -3 $O/bench.err
