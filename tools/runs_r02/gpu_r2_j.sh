#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02j; mkdir -p $O
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 4 --warmup 1 --pool 8000 --tile 5 --cpu-sample 0 --ragged 0 > $O/bench_q$q.log 2> $O/bench_q$q.err
done
python - <<'PY'
import json
for q in (4,8,16):
    for l in open("gpurun_out/r02j/bench_q%d.log"%q):
        if l.startswith("{"):
            d=json.loads(l); print(q, d["value"], d["ms_per_step"], d["value_streamed"], d["streamed"]["ms_per_step"])
PY
