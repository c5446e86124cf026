export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02fin}; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
timeout 500 python bench.py --steps 5 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
timeout 500 python bench.py --steps 3 --warmup 1 --from-raw 1 --cpu-sample 256 > $O/bench_from_raw.json 2> $O/bench_from_raw.err
timeout 400 python bench.py --workload eventalign --steps 3 --warmup 1 > $O/bench_eventalign.json 2> $O/bench_eventalign.err
tail -4 $O/pytest.log; for f in default from_raw eventalign; do tail -c 200 $O/bench_$f.json; echo; done
