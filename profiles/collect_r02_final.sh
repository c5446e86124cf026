# Round-2 end-of-round measurement set (run on the GPU box through gpurun):  bash profiles/collect_r02_final.sh [tag]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02f}; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
# 1. the bench line (default: configs[1] literal, resident + streamed + ragged, CPU baseline on the box's cgroup CPUs)
timeout 500 python bench.py --steps 5 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
# 2. the same from raw signal (int16 ADC counts up, event detection + MoM on the device)
timeout 500 python bench.py --steps 3 --warmup 1 --from-raw 1 --cpu-sample 256 > $O/bench_from_raw.json 2> $O/bench_from_raw.err
# 3. configs[2], configs[3], configs[0]
timeout 400 python bench.py --workload eventalign --steps 3 --warmup 1 > $O/bench_eventalign.json 2> $O/bench_eventalign.err
timeout 400 python bench.py --workload variants --steps 3 --warmup 1 > $O/bench_variants.json 2> $O/bench_variants.err
timeout 200 python bench.py --workload cpu-t1 --cpu-sample 200 > $O/bench_cpu_t1.json 2> $O/bench_cpu_t1.err
# 4. kernel trace + stats of the default command (resident variant only)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 > $O/trace.log 2>&1 )
f=$(find $O/trace -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/trace.md
# 5. HBM counters of kernel A alone (a process that launches little else; 8192 reads per launch; one counter per pass, --pmc only with
#    --kernel-trace).  The instruction-counter passes (SQ_INSTS_*, eight counters per pass) and FETCH_SIZE at 16384 reads per launch did
#    not finish inside 150 s each when tried over the final kernel (gpurun r02f): not repeated here.
A="python $R/tools/align_ab.py --child --pool 1024 --tile 8 --reps 2"
( cd /tmp
timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -o p3 -- $A > $O/pmc3.log 2>&1; echo "rc=$?" >> $O/pmc3.log
timeout 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc4 -o p4 -- $A > $O/pmc4.log 2>&1; echo "rc=$?" >> $O/pmc4.log )
python3 profiles/pmc_summary.py $O 8192 13463.2 > $O/pmc.json 2> $O/pmc.err
tail -3 $O/pytest.log; for f in default from_raw eventalign variants cpu_t1; do tail -c 300 $O/bench_$f.json; echo; done; head -8 $O/trace.md | cut -c1-160; tail -n 1 $O/pmc3.log $O/pmc4.log; head -c 600 $O/pmc.json
