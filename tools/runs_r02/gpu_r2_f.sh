#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
V=nanopolish_amd/variants
timeout 500 python tools/align_ab.py $V/libnp_hip_all.so $V/libnp_hip_s8.so $V/libnp_hip_s16.so $V/libnp_hip_v8.so $V/libnp_hip_v16.so $V/libnp_hip_nobt.so > $O/align_ab.txt 2>&1
cat $O/align_ab.txt
