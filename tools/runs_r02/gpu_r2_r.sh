export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02r}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
for d in 0 2 1; do
  NP_ALIGN_DEFER=$d timeout 300 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_defer.so $V/libnp_hip_nobt.so 2>&1 | sed "s/^{/{\"defer\": $d, /" >> $O/ab.jsonl
done
cat $O/ab.jsonl
