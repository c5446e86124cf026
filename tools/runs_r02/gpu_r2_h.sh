#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02h; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/pytest.log 2>&1
( time timeout 500 python bench.py --steps 3 --warmup 1 --cpu-sample 512 ) > $O/bench.log 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
( time timeout 500 python bench.py --workload eventalign --steps 2 --warmup 1 ) > $O/bench_eventalign.log 2> $O/bench_eventalign.err; echo "rc=$?" >> $O/bench_eventalign.err
( time timeout 500 python bench.py --workload variants --steps 2 --warmup 1 ) > $O/bench_variants.log 2> $O/bench_variants.err; echo "rc=$?" >> $O/bench_variants.err
cat $O/pytest.log
python - <<'PY'
import json
for f in ("bench","bench_eventalign","bench_variants"):
    for l in open("gpurun_out/r02h/%s.log"%f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["value"], d.get("ms_per_step"), d.get("value_streamed"), d.get("value_ragged"), (d.get("roofline") or {}).get("kernel_ms_per_step"), d.get("cpu_baseline",{}).get("value") if d.get("cpu_baseline") else None)
PY
tail -3 $O/bench_eventalign.err $O/bench_variants.err
