export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02q}; mkdir -p $O; cd /tmp
NP_HIP_LIB=$R/nanopolish_amd/variants/libnp_hip_defer.so timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/tools/align_ab.py --child --pool 2048 --tile 16 --reps 3 > $O/trace.log 2>&1
cd $R; f=$(find $O/trace -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace.md; head -12 $O/trace.md | cut -c1-200; tail -3 $O/trace.log
