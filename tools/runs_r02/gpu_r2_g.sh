#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02g; mkdir -p $O
export TMPDIR=/tmp
V=nanopolish_amd/variants
timeout 500 python tools/align_ab.py $V/libnp_hip_all.so $V/libnp_hip_u8off.so $V/libnp_hip_cmpxoff.so $V/libnp_hip_traceoff.so $V/libnp_hip_nobt.so > $O/align_ab.txt 2>&1
timeout 200 python tools/align_ab.py --ragged 1 $V/libnp_hip_all.so >> $O/align_ab.txt 2>&1
cat $O/align_ab.txt
( timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_reflevel.py tests/test_gpu_dropin.py -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
cat $O/pytest.log
