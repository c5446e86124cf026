export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02ze}; mkdir -p $O; cd $R
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
NP_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --pool 4000 --tile 2 --cpu-sample 0 --streamed 0 --ragged 0 > $O/bench_2rank.json 2> $O/bench_2rank.err
timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --pool 4000 --tile 2 --cpu-sample 0 --streamed 0 --ragged 0 > $O/bench_1rank.json 2> $O/bench_1rank.err
python - <<PY
import json
for f in ("bench_2rank","bench_1rank"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["n_gpus"], d["value"], d["ms_per_step"], d.get("site_table"))
    except Exception as e:
        print(f, "FAILED", e, open("$O/%s.err"%f).read()[-600:])
PY
