#!/usr/bin/env python3
"""Assembles profiles/r02_round_end.md from the outputs of `bash profiles/collect_r02_final.sh r02z` and
`bash tools/gpu_r2_m.sh r02z2` (gpurun_out/r02z, gpurun_out/r02z2).   usage: make_r02_summary.py > profiles/r02_round_end.md"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = os.path.join(ROOT, "gpurun_out", "r02z"); B = os.path.join(ROOT, "gpurun_out", "r02z2")
line = lambda d, f: open(os.path.join(d, f)).read().strip().splitlines()[-1]
o = []
o.append("# Round 2 -- end-of-round measurement set (MI355X, 1 GPU)\n")
o.append("Collected through gpurun with `bash profiles/collect_r02_final.sh r02z` (tests, default bench, configs[0], kernel trace, HBM counters of "
         "kernel A) and `bash tools/gpu_r2_m.sh r02z2` (from-raw bench, configs[2], configs[3]); raw outputs under `gpurun_out/` (scratch), "
         "summarised here by `profiles/make_r02_summary.py`.\n")
o.append("GPU tests on the same box: `" + [l for l in open(os.path.join(A, "pytest.log")).read().splitlines() if " passed" in l][-1].strip() + "`\n")
o.append("## Bench lines\n")
o.append("Default (`python bench.py --steps 5 --warmup 1`): BASELINE.json configs[1] -- 100 000 reads per step (20 000 distinct x 5), ~8k events each, "
         "work items generated, reads recalibrated and scored on the device; `value` resident, `value_streamed` host-fed, `value_ragged` log-normal lengths:\n")
o.append("```json\n" + line(A, "bench_default.json") + "\n```\n")
o.append("From raw signal (`--from-raw 1 --steps 3 --cpu-sample 256`: int16 ADC counts -> pA -> scrappie event detection -> MoM scalings on the device in "
         "front of the same pass; the streamed variant uploads the ADC counts):\n")
o.append("```json\n" + line(B, "bench_from_raw.json") + "\n```\n")
o.append("configs[2], eventalign from raw signal (`python bench.py --workload eventalign --steps 3 --warmup 1`; CPU leg = the reference's own `SquiggleRead` + "
         "`align_read_to_ref`, OpenMP over reads on the box's cgroup CPUs):\n")
o.append("```json\n" + line(B, "bench_eventalign.json") + "\n```\n")
o.append("configs[3], variants screening shape (`python bench.py --workload variants --steps 3 --warmup 1`; CPU leg = the reference's own `profile_hmm_score`):\n")
o.append("```json\n" + line(B, "bench_variants.json") + "\n```\n")
o.append("configs[0] plumbing line (`python bench.py --workload cpu-t1 --cpu-sample 200`: the reference's code, one host thread, no GPU):\n")
o.append("```json\n" + line(A, "bench_cpu_t1.json") + "\n```\n")
o.append("## rocprofv3 --kernel-trace --stats\n")
o.append("`rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0` (4 launches incl. warm-up), "
         "summarised with `profiles/summarize_rocpd.py`; the average duration of `np_event_align_kernel` agrees with `roofline.avg_launch_ms` of the "
         "bench line (HIP events inside bench.py):\n")
o.append(open(os.path.join(A, "trace.md")).read())
o.append("\n## HBM traffic of kernel A (PMC)\n")
o.append("`rocprofv3 --kernel-trace --pmc <counter> -- python tools/align_ab.py --child --pool 1024 --tile 8 --reps 2` -- one counter per pass, a process "
         "that launches kernel A alone, 8192 reads per launch (counter collection on the full bench command does not finish inside its time limit at "
         "16 384 or 100 000 reads per launch: only the instruction/wait pass of `profiles/r02_pmc.json` comes from the bench command).  Sums over "
         "all instances of the counter, per launch of `np_event_align_kernel`:\n")
o.append("| counter | launches | raw per launch (KB) | correction | bytes per read |\n|---|---|---|---|---|")
for c, corr in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    per = {}
    for f in glob.glob(os.path.join(A, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "np_event_align_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    if per:
        v = sum(per.values()) / len(per)
        o.append("| %s | %d | %.1f | x%g | %.0f |" % (c, len(per), v, corr, v * 1024 * corr / 8192))
o.append("\n(corrections calibrated in round 1 with `tools/hbm_counter_calib.hip`; the algorithmic figure of SURVEY 8d is 1.45 MB per read -- the kernel moves "
         "less than that because its trace is 32 B per band instead of the reference's 100 B.)\n")
print("\n".join(o))
