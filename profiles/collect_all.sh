# Round-1 measurement set (run on the GPU box through gpurun):  bash profiles/collect_all.sh [tag]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r01e}; mkdir -p $O; cd /tmp
# 1. the bench line (default workload: 32768 reads/step from events, calibration on the device) + CPU baseline
timeout 500 python $R/bench.py --steps 5 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
# 2. the same from raw signal (event detection + MoM on the device)
timeout 500 python $R/bench.py --steps 3 --warmup 1 --from-raw 1 > $O/bench_from_raw.json 2> $O/bench_from_raw.err
# 3. kernel trace + stats of the default command
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 > $O/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_raw -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --from-raw 1 > $O/trace_raw.log 2>&1
# 4. PMC passes (separate runs, kernel-trace only), smaller batch
bash $R/profiles/collect_pmc.sh > $O/pmc.log 2>&1
bash $R/profiles/collect_pmc_lds.sh >> $O/pmc.log 2>&1
# 5. BASELINE configs 3 and 4: eventalign (from raw signal) and variants screening, with their reference-backed CPU legs
timeout 600 python $R/tests/bench_eventalign.py --pool 256 --tile 80 --cpu-sample 256 > $O/bench_eventalign.json 2> $O/bench_eventalign.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_ea -o t -- python $R/tests/bench_eventalign.py --pool 256 --tile 80 --cpu-sample 0 > $O/trace_ea.log 2>&1
timeout 400 python $R/tests/bench_variants.py > $O/bench_variants.json 2> $O/bench_variants.err
bash $R/profiles/collect_pmc_eventalign.sh > $O/pmc_ea.txt 2>&1
for d in trace trace_raw trace_ea; do f=$(find $O/$d -name "*results.db" | head -1); [ -n "$f" ] && python3 $R/profiles/summarize_rocpd.py $f > $O/$d.md; done
tail -c 300 $O/bench_default.json; echo; tail -c 200 $O/bench_from_raw.json; echo; tail -c 400 $O/bench_eventalign.json; echo; tail -c 300 $O/bench_variants.json
