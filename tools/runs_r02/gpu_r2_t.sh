export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02t}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
( timeout 900 python -m pytest tests -m gpu -q -k "fuzz" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
for d in 0 1; do
  NP_ALIGN_DEFER=$d timeout 300 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_trim.so 2>&1 | sed "s/^{/{\"defer\": $d, /" >> $O/ab.jsonl
  NP_ALIGN_DEFER=$d timeout 300 python tools/align_ab.py --pool 2048 --tile 16 --ragged 1 $V/libnp_hip_trim.so 2>&1 | sed "s/^{/{\"defer\": $d, \"ragged\": 1, /" >> $O/ab.jsonl
done
cd /tmp; NP_HIP_LIB=$R/$V/libnp_hip_trim.so timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/tools/align_ab.py --child --pool 2048 --tile 16 --reps 3 > $O/trace.log 2>&1
cd $R; f=$(find $O/trace -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace.md
tail -3 $O/pytest.log; cat $O/ab.jsonl; head -6 $O/trace.md | cut -c1-170
