export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02x}; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
timeout 300 python bench.py --steps 4 --warmup 1 --pool 8000 --tile 5 --cpu-sample 64 --streamed 0 --ragged 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["cpu_baseline"].get("check"))
PY
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 3 --warmup 1 --pool 8000 --tile 5 --cpu-sample 0 --streamed 0 --ragged 0 > $O/trace.log 2>&1
cd $R; f=$(find $O/trace -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace.md
tail -4 $O/pytest.log; sed -n 3,20p $O/trace.md | cut -c1-130
