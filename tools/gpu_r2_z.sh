export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02z5}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
( timeout 900 python -m pytest tests -m gpu -q -k "align or parity or fuzz or edges or dropin or reflevel" 2>&1 | tail -3 ) > $O/pytest.log 2>&1
timeout 300 python tools/align_ab.py --pool 2048 --tile 16 --reps 4 $V/libnp_hip_pipe.so $V/libnp_hip_cur.so $V/libnp_hip_pipe.so $V/libnp_hip_cur.so >> $O/ab.jsonl 2>&1
tail -2 $O/pytest.log; cat $O/ab.jsonl
