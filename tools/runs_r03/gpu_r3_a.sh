#!/bin/bash
# round-3 first GPU pass: parity tests (incl. long reads, pipelined batch binding), the folded bench line, the batch-binding
# throughput, counter passes over the shipped kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
NP_VERBOSE=1 python -c "
import torch
from nanopolish_amd.api import Context
c = Context(0); print(c.info()); c.close()" > $O/probe.log 2>&1
( time timeout 900 python bench.py --steps 3 --warmup 1 ) > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
( time timeout 600 python tests/bench_batch_dropin.py ) > $O/batch_dropin.json 2> $O/batch_dropin.err; echo "rc=$?" >> $O/batch_dropin.err
PASS_TIMEOUT=120 bash profiles/collect_r03_pmc.sh r03a_pmc 2048 > $O/pmc.log 2>&1
tail -4 $O/pytest.log; cat $O/probe.log | tail -2; tail -c 1500 $O/bench.json; tail -3 $O/bench.err; cat $O/batch_dropin.json | cut -c1-600; tail -3 $O/batch_dropin.err; tail -12 $O/pmc.log | cut -c1-400
