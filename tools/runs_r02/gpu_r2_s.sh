export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02s}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
( timeout 900 python -m pytest tests -m gpu -q -k "align or parity or fuzz or edges or dropin or reflevel" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
for d in 0 1; do
  NP_ALIGN_DEFER=$d timeout 300 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_ilv.so $V/libnp_hip_trim.so 2>&1 | sed "s/^{/{\"defer\": $d, /" >> $O/ab.jsonl
done
NP_ALIGN_DEFER=0 timeout 300 python tools/align_ab.py --pool 2048 --tile 16 --ragged 1 $V/libnp_hip_ilv.so $V/libnp_hip_trim.so 2>&1 | sed "s/^{/{\"ragged\": 1, /" >> $O/ab.jsonl
tail -5 $O/pytest.log; cat $O/ab.jsonl
