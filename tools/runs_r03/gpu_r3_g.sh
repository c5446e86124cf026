#!/bin/bash
# round-3: chain kernels after the branch-free walk step: kernel 1 parity, kernel 2 with / without walk priority, waves per CU
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
( NP_EA_KERNEL=1 timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_k1.log 2>&1; echo "k1 pytest rc=$?" >> $O/pytest_k1.log
( NP_EA_KERNEL=2 timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_k2.log 2>&1; echo "k2 pytest rc=$?" >> $O/pytest_k2.log
for cfg in "2 0 16" "2 3 16" "2 1 16" "2 0 12" "1 0 20"; do set -- $cfg
  ( NP_EA_KERNEL=$1 NP_EA_WALK_PRIO=$2 NP_EA_WAVES_PER_CU=$3 timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 0 ) > $O/ea_$1_$2_$3.json 2> $O/ea_$1_$2_$3.err
  echo "kernel $1 prio $2 waves $3: $(grep -o '"value": [0-9.]*\|"eventalign_chain": [0-9.]*\|"backtrack": [0-9]*\|"fill": [0-9]*' $O/ea_$1_$2_$3.json | tr '\n' ' ')"
done
tail -3 $O/pytest_k1.log; tail -3 $O/pytest_k2.log
