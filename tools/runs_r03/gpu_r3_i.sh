#!/bin/bash
# round-3: two-read chain kernel, tournament arg-max: parity (all eventalign tests, both kernels), timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_default.log 2>&1; echo "default pytest rc=$?" >> $O/pytest_default.log
for cfg in "2 0 20" "2 0 16" "1 0 20"; do set -- $cfg
  ( NP_EA_KERNEL=$1 NP_EA_WALK_PRIO=$2 NP_EA_WAVES_PER_CU=$3 timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 256 ) > $O/ea_$1_$2_$3.json 2> $O/ea_$1_$2_$3.err
  echo "kernel $1 prio $2 waves $3: $(grep -o '"value": [0-9.]*\|"eventalign_chain": [0-9.]*\|"backtrack": [0-9]*\|"fill": [0-9]*\|"geometry": [0-9]*\|"rows_match": [a-z]*\|"copies_identical": [a-z]*' $O/ea_$1_$2_$3.json | tr '\n' ' ')"
done
tail -2 $O/pytest_default.log
