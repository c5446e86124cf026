export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02v2}; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > $O/pytest.log 2>&1
timeout 500 python bench.py --steps 5 --warmup 1 --cpu-sample 64 > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest.log; tail -c 1500 $O/bench_default.json; tail -2 $O/bench_default.err
