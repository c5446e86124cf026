#!/bin/bash
# round-2 third GPU pass: kernel A ablation matrix, occupancy, counters available, streamed-feed host timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
V=nanopolish_amd/variants
timeout 1500 python tools/align_ab.py $V/libnp_hip_all.so $V/libnp_hip_nointc.so $V/libnp_hip_w5.so $V/libnp_hip_w6.so $V/libnp_hip_w8.so $V/libnp_hip_all.so \
    $V/libnp_hip_abl132.so $V/libnp_hip_abl133.so $V/libnp_hip_abl134.so $V/libnp_hip_abl140.so $V/libnp_hip_abl148.so $V/libnp_hip_abl164.so \
    $V/libnp_hip_abl196.so $V/libnp_hip_abl204.so $V/libnp_hip_abl220.so $V/libnp_hip_abl255.so > $O/align_ab.txt 2>&1
( cd /tmp && timeout 120 rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$O/counters_avail.txt 2>&1 )
( time timeout 600 python bench.py --steps 4 --warmup 1 --pool 8000 --tile 5 --cpu-sample 0 --ragged 0 ) > $O/bench_streamed.log 2> $O/bench_streamed.err
cat $O/align_ab.txt; python - <<'PY'
import json
for l in open("gpurun_out/r02c/bench_streamed.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["streamed"])
PY
grep -c . $O/counters_avail.txt
