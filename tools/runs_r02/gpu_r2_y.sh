export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02y}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
for i in 1 2; do
timeout 300 python tools/align_ab.py --pool 2048 --tile 16 --reps 4 $V/libnp_hip_cur.so $V/libnp_hip_prio1.so $V/libnp_hip_prio3.so >> $O/ab.jsonl 2>&1
done
timeout 300 python tools/align_ab.py --pool 2048 --tile 16 --reps 4 --ragged 1 $V/libnp_hip_cur.so $V/libnp_hip_prio3.so 2>&1 | sed "s/^{/{\"ragged\": 1, /" >> $O/ab.jsonl
cat $O/ab.jsonl
