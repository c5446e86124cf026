export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02zk}; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests -m gpu -q -k "events or reflevel or batch" 2>&1 | tail -3 ) > $O/pytest.log 2>&1
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --pool 8000 --tile 5 --from-raw 1 --cpu-sample 0 --streamed 0 --ragged 0 > $O/trace.log 2>&1
cd $R; f=$(find $O/trace -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace.md
tail -2 $O/pytest.log; grep "np_ed_\|np_mom\|np_adc" $O/trace.md | cut -c1-150
