export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02zn}; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > $O/pytest.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 200 python bench.py --steps 2 --warmup 1 --pool 4000 --tile 5 --cpu-sample 32 --ragged 0 > $O/bench.json 2> $O/bench.err
tail -2 $O/pytest.log; tail -1 $O/smoke.log; tail -c 400 $O/bench.json
