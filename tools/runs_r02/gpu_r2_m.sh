# from-raw bench line + configs[2]/[3] with the quota-sized CPU thread count:  bash tools/gpu_r2_m.sh [tag]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02z2}; mkdir -p $O; cd $R
timeout 500 python bench.py --steps 3 --warmup 1 --from-raw 1 --cpu-sample 256 > $O/bench_from_raw.json 2> $O/bench_from_raw.err
timeout 400 python bench.py --workload eventalign --steps 3 --warmup 1 > $O/bench_eventalign.json 2> $O/bench_eventalign.err
timeout 400 python bench.py --workload variants --steps 3 --warmup 1 > $O/bench_variants.json 2> $O/bench_variants.err
for f in from_raw eventalign variants; do tail -c 600 $O/bench_$f.json; echo; tail -3 $O/bench_$f.err; done
