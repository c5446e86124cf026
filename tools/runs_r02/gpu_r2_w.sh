export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02w}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > $O/pytest.log 2>&1
for l in cur bglob; do
  NP_HIP_LIB=$R/$V/libnp_hip_$l.so timeout 300 python bench.py --steps 4 --warmup 1 --pool 8000 --tile 5 --cpu-sample 0 --streamed 0 --ragged 0 > $O/bench_$l.json 2> $O/bench_$l.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$l.json").read().strip().splitlines()[-1])
print("$l", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d.get("check"))
PY
done
cd /tmp; NP_HIP_LIB=$R/$V/libnp_hip_cur.so timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 3 --warmup 1 --pool 8000 --tile 5 --cpu-sample 0 --streamed 0 --ragged 0 > $O/trace.log 2>&1
cd $R; f=$(find $O/trace -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace.md
tail -2 $O/pytest.log; head -22 $O/trace.md | cut -c1-150
