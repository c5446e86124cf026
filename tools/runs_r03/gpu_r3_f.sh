#!/bin/bash
# round-3: phase breakdown of the two-read chain kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
( NP_EA_KERNEL=2 timeout 600 python tests/bench_eventalign.py --steps 3 --warmup 1 --cpu-sample 0 ) > $O/ea_k2.json 2> $O/ea_k2.err; echo "rc=$?" >> $O/ea_k2.err
cut -c1-1200 $O/ea_k2.json; tail -2 $O/ea_k2.err
