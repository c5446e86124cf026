#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02m; mkdir -p $O
V=nanopolish_amd/variants
for t in hmm_nospec hmm_spec hmm_nospec hmm_spec; do
  NP_HIP_LIB=$PWD/$V/libnp_hip_$t.so timeout 200 python bench.py --steps 3 --warmup 1 --pool 4000 --tile 5 --cpu-sample 0 --streamed 0 --ragged 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$t', d['value'], d['roofline']['kernel_ms_per_step'])" >> $O/hmm_ab.txt
done
cat $O/hmm_ab.txt
( timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_dropin.py tests/test_gpu_sites.py -m gpu -q 2>&1 | tail -3 )
