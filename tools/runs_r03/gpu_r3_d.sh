#!/bin/bash
# round-3 fourth GPU pass: batch binding after the one-fetch-per-batch reference and with int16 ADC input
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_batch_dropin.py -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python tests/bench_batch_dropin.py --sizes 512,2048,8192,32768 ) > $O/batch_dropin.json 2> $O/batch_dropin.err; echo "rc=$?" >> $O/batch_dropin.err
tail -6 $O/pytest.log; cat $O/batch_dropin.json | cut -c1-1800; tail -3 $O/batch_dropin.err
