#!/bin/bash
# round-3 second GPU pass: counter passes at a launch that fills every wave slot (8192 reads), batch-binding throughput with host phase timers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_batch_dropin.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python tests/bench_batch_dropin.py --sizes 512,8192 ) > $O/batch_dropin.json 2> $O/batch_dropin.err; echo "rc=$?" >> $O/batch_dropin.err
PASS_TIMEOUT=240 bash profiles/collect_r03_pmc.sh r03b_pmc 8192 > $O/pmc.log 2>&1
tail -4 $O/pytest.log; cat $O/batch_dropin.json | cut -c1-1200; tail -3 $O/batch_dropin.err; cat gpurun_out/r03b_pmc/passes.log
