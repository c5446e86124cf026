#!/usr/bin/env python3
"""Assembles profiles/r02_round_end.md from the outputs of `bash profiles/collect_r02_final.sh <tag>` (gpurun_out/<tag>).
usage: make_r02_summary.py [tag] > profiles/r02_round_end.md"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r02f"); B = A
line = lambda d, f: open(os.path.join(d, f)).read().strip().splitlines()[-1]
o = []
o.append("# Round 2 -- end-of-round measurement set (MI355X, 1 GPU)\n")
o.append("Collected through gpurun with `bash profiles/collect_r02_final.sh` (tests, the bench lines of all four configs, kernel trace, counter passes "
         "over kernel A alone); raw outputs under `gpurun_out/` (scratch), summarised here by `profiles/make_r02_summary.py`.  The tests and the default, "
         "from-raw and eventalign lines are those of the round's last GPU call (`tools/runs_r02/gpu_r2_z.sh r02fin`, after the event detector's fusion); the variants and "
         "cpu-t1 lines, the kernel trace and the counters are from the collection just before it (r02end: the kernels of the default step did not change in between).\n")
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
PMC_READS = 16384 if os.path.exists(os.path.join(A, "pmc1.log")) else 8192
o.append("\n## Counters of kernel A (PMC)\n")
o.append("`rocprofv3 --kernel-trace --pmc <counters> -- python tools/align_ab.py --child --pool 2048 --tile 8 --reps 2` -- one counter set per pass, a "
         "process that launches kernel A alone, 8192 reads per launch.  Passes that did not finish inside their 150 s limit are listed as such "
         "(counter collection over these kernels is unpredictable on this pool; the mid-round passes are in `r02_pmc.json`).\n")
o.append("| pass | counters | outcome |\n|---|---|---|")
names = {1: "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY",
         2: "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH", 3: "FETCH_SIZE", 4: "WRITE_SIZE"}
PMC_DONE = [k for k in (1, 2, 3, 4) if os.path.exists(os.path.join(A, 'pmc%d.log' % k))]
for k in PMC_DONE:
    log = os.path.join(A, "pmc%d.log" % k)
    rc = open(log).read().strip().splitlines()[-1] if os.path.exists(log) else "not run"
    per = {}
    for f in glob.glob(os.path.join(A, "pmc%d" % k, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "np_event_align_kernel" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    res = "; ".join("%s = %.1f per launch (%d launches)" % (c, sum(v.values()) / len(v), len(v)) for c, v in sorted(per.items()))
    o.append("| %d | %s | %s%s |" % (k, names[k], rc, (": " + res) if res else (" (timed out)" if rc == "rc=124" else "")))
o.append("\nFETCH_SIZE / WRITE_SIZE are in KB: x1024 / reads per launch = bytes per read (corrections x2 / x1, calibrated in round 1 with `tools/hbm_counter_calib.hip`); the "
         "algorithmic figure of SURVEY 8d is 1.45 MB per read moved in total -- the kernel moves less because its trace is 32 B per band instead of the "
         "reference's 100 B.\n")
print("\n".join(o))
