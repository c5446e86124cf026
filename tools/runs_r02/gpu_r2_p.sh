# deferred lane-per-read back-track: tests, then timing (defer on / off) of kernel A's family
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02p}; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
V=nanopolish_amd/variants
for d in 0 1; do
  NP_ALIGN_DEFER=$d timeout 300 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_defer.so 2>&1 | sed "s/^{/{\"defer\": $d, /" >> $O/ab.jsonl
  NP_ALIGN_DEFER=$d timeout 300 python tools/align_ab.py --pool 2048 --tile 16 --ragged 1 $V/libnp_hip_defer.so 2>&1 | sed "s/^{/{\"defer\": $d, \"ragged\": 1, /" >> $O/ab.jsonl
done
tail -15 $O/pytest.log; cat $O/ab.jsonl
