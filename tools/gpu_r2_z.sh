export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02zf}; mkdir -p $O; cd $R
V=nanopolish_amd/variants
for l in cur b640 b768; do
  NP_HIP_LIB=$R/$V/libnp_hip_$l.so timeout 300 python bench.py --steps 3 --warmup 1 --pool 8000 --tile 5 --cpu-sample 64 --streamed 0 --ragged 0 > $O/bench_$l.json 2> $O/bench_$l.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$l.json").read().strip().splitlines()[-1])
print("$l", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]["hmm_score"], d["cpu_baseline"]["check"]["max_abs_dLLR"], d["cpu_baseline"]["check"]["groups_missing_on_gpu"])
PY
done
