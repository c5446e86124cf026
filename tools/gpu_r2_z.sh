export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02zj}; mkdir -p $O; cd $R
for w in 256 128 64 32; do
NP_ED_WARMUP=$w timeout 400 python bench.py --steps 2 --warmup 1 --pool 4000 --tile 5 --from-raw 1 --cpu-sample 32 --streamed 0 --ragged 0 > $O/bench_$w.json 2> $O/bench_$w.err
python - <<PY
import json
d=json.loads(open("$O/bench_$w.json").read().strip().splitlines()[-1])
print("warmup $w", d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]["event_detect"], d["cpu_baseline"]["check"]["pairs_bit_exact"], d["cpu_baseline"]["check"]["max_abs_dLLR"])
PY
done
