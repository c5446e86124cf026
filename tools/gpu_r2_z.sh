export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02z4}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
timeout 300 python tools/align_ab.py --pool 2048 --tile 16 --reps 4 $V/libnp_hip_fprio1.so $V/libnp_hip_walk0.so $V/libnp_hip_cur.so $V/libnp_hip_walk0.so $V/libnp_hip_fprio1.so $V/libnp_hip_cur.so >> $O/ab.jsonl 2>&1
cat $O/ab.jsonl
