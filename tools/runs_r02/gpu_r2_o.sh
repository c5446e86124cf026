# kernel A: where the time goes (ablation builds, all WITHOUT the back-track unless named) and how it scales with resident waves
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02o}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
timeout 400 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_base.so $V/libnp_hip_nobt.so $V/libnp_hip_notrace.so $V/libnp_hip_fixmove.so $V/libnp_hip_fp32.so $V/libnp_hip_noem.so > $O/abl.jsonl 2>&1
for w in 8 6 5 4 3 2; do
  NP_ALIGN_BLOCKS_PER_CU=$w timeout 200 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_base.so $V/libnp_hip_nobt.so 2>&1 | sed "s/^{/{\"waves\": $w, /" >> $O/waves.jsonl
done
cat $O/abl.jsonl $O/waves.jsonl
