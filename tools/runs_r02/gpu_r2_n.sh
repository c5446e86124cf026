# kernel A rewrite (interleaved ring): parity tests, then A/B timing against the previous kernel:  bash tools/gpu_r2_n.sh [tag]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02n}; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests -m gpu -q -k "align or parity or reflevel or dropin or jobs or sites or events" 2>&1 | tail -15 ) > $O/pytest.log 2>&1
timeout 300 python tools/align_ab.py --pool 2048 --tile 16 nanopolish_amd/variants/libnp_hip_old.so nanopolish_amd/variants/libnp_hip_ilv.so > $O/ab.jsonl 2>&1
timeout 300 python tools/align_ab.py --pool 2048 --tile 16 --ragged 1 nanopolish_amd/variants/libnp_hip_old.so nanopolish_amd/variants/libnp_hip_ilv.so > $O/ab_ragged.jsonl 2>&1
tail -15 $O/pytest.log; cat $O/ab.jsonl $O/ab_ragged.jsonl
