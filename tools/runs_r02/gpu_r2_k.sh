#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02k; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
( time timeout 400 python bench.py --steps 5 --warmup 1 --cpu-sample 512 ) > $O/bench.log 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
bash profiles/collect_pmc_r02.sh r02k > $O/pmc_collect.log 2>&1
cat $O/pytest.log; tail -4 $O/pmc_collect.log
python - <<'PY'
import json
for l in open("gpurun_out/r02k/bench.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["value_streamed"], d["streamed"]["steady_state_ms_per_step"], d["value_ragged"], d["roofline"]["kernel_ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["check"])
PY
