#!/bin/bash
# round-3 third GPU pass: all GPU tests (new: variants / eventalign batched bindings), batch-binding throughput after the host fast paths
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python tests/bench_batch_dropin.py --sizes 512,8192,32768 ) > $O/batch_dropin.json 2> $O/batch_dropin.err; echo "rc=$?" >> $O/batch_dropin.err
tail -25 $O/pytest.log; cat $O/batch_dropin.json | cut -c1-1500; tail -3 $O/batch_dropin.err
