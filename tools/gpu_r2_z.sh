export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02zg}; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests -m gpu -q -k "events or reflevel or batch or dropin" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
timeout 400 python bench.py --steps 2 --warmup 1 --pool 4000 --tile 5 --from-raw 1 --cpu-sample 32 --streamed 0 --ragged 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["cpu_baseline"].get("check"))
except Exception as e:
    print("bench failed", e, open("$O/bench.err").read()[-800:])
PY
tail -8 $O/pytest.log
