export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02z8}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
timeout 300 python tools/align_ab.py --pool 2048 --tile 16 --reps 4 $V/libnp_hip_strace_nobt.so $V/libnp_hip_nobt.so $V/libnp_hip_strace_nobt.so $V/libnp_hip_nobt.so >> $O/ab.jsonl 2>&1
cat $O/ab.jsonl
