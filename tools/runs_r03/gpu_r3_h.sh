#!/bin/bash
# round-3: two-read chain kernel with the dual interleaved walk: parity, then timing at 16 / 20 waves per CU, with / without priority; kernel 1 as control
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
for w in 16 20; do ( NP_EA_KERNEL=2 NP_EA_WAVES_PER_CU=$w timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_k2_$w.log 2>&1; echo "k2 w$w pytest rc=$?" >> $O/pytest_k2_$w.log; done
( NP_EA_KERNEL=1 timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q ) > $O/pytest_k1.log 2>&1; echo "k1 pytest rc=$?" >> $O/pytest_k1.log
for cfg in "2 1 16" "2 0 16" "2 1 20" "2 0 20" "1 0 20"; do set -- $cfg
  ( NP_EA_KERNEL=$1 NP_EA_WALK_PRIO=$2 NP_EA_WAVES_PER_CU=$3 timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 64 ) > $O/ea_$1_$2_$3.json 2> $O/ea_$1_$2_$3.err
  echo "kernel $1 prio $2 waves $3: $(grep -o '"value": [0-9.]*\|"eventalign_chain": [0-9.]*\|"backtrack": [0-9]*\|"fill": [0-9]*\|"geometry": [0-9]*\|"rows_match": [a-z]*\|"copies_identical": [a-z]*' $O/ea_$1_$2_$3.json | tr '\n' ' ')"
done
tail -2 $O/pytest_k2_16.log; tail -2 $O/pytest_k2_20.log; tail -2 $O/pytest_k1.log
