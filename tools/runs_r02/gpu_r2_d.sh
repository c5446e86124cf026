#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
V=nanopolish_amd/variants
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -5 ) > $O/pytest.log 2>&1
timeout 900 python tools/align_ab.py $V/libnp_hip_all.so $V/libnp_hip_noearly.so $V/libnp_hip_abl4.so $V/libnp_hip_abl132.so $V/libnp_hip_all.so > $O/align_ab.txt 2>&1
timeout 300 python tools/align_ab.py --ragged 1 $V/libnp_hip_all.so >> $O/align_ab.txt 2>&1
( time timeout 900 python bench.py --steps 3 --warmup 1 --cpu-sample 1024 ) > $O/bench.log 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
bash profiles/collect_pmc_r02.sh r02d > $O/pmc_collect.log 2>&1
cat $O/pytest.log $O/align_ab.txt; tail -3 $O/pmc_collect.log
python - <<'PY'
import json
for l in open("gpurun_out/r02d/bench.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["value_streamed"], d["streamed"]["ms_per_step"], d["value_ragged"], d["roofline"]["kernel_ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["check"])
PY
