export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02zp}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
( timeout 300 python -m pytest tests -m gpu -q -k "eventalign or reflevel" 2>&1 | tail -2 ) > $O/pytest.log 2>&1
for l in eaprio0 cur eaprio0 cur; do
  NP_HIP_LIB=$R/$V/libnp_hip_$l.so timeout 200 python bench.py --workload eventalign --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['ms_per_step'], d['kernel_ms_per_step']['eventalign_chain'])"
done
tail -1 $O/pytest.log
