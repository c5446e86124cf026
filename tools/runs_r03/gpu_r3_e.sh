#!/bin/bash
# round-3: the two-reads-per-wave eventalign chain kernel against the one-read kernel: parity tests under both, timing of both
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
for v in 2 1; do
  ( NP_EA_KERNEL=$v timeout 600 python -m pytest tests/test_gpu_reflevel.py tests/test_gpu_eventalign_dropin.py -m gpu -q -x ) > $O/pytest_k$v.log 2>&1; echo "k$v pytest rc=$?" >> $O/pytest_k$v.log
  ( NP_EA_KERNEL=$v timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 ) > $O/ea_k$v.json 2> $O/ea_k$v.err; echo "rc=$?" >> $O/ea_k$v.err
done
for v in 2 1; do tail -4 $O/pytest_k$v.log; cut -c1-900 $O/ea_k$v.json; tail -2 $O/ea_k$v.err; done
