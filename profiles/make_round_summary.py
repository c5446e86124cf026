#!/usr/bin/env python3
"""Assembles profiles/r01_e_round_end.md from the outputs of `bash profiles/collect_all.sh r01e` (gpurun_out/r01e/) and of
`bash profiles/collect_counter_calib.sh`, `bash profiles/collect_pmc_from_raw.sh` (gpurun_out/).   usage: make_round_summary.py > out.md"""
import csv, glob, os, re, collections, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
R = os.path.join(G, "r01e")
line = lambda f: open(os.path.join(R, f)).read().strip().splitlines()[-1]
out = []
out.append("# Round 1 (e) -- end-of-round measurement set (MI355X, 1 GPU)\n")
out.append("Collected with `bash profiles/collect_all.sh r01e` through gpurun (one box, one call); raw outputs under `gpurun_out/r01e/` (scratch), summaries here (`profiles/make_round_summary.py`).\n")
out.append("## Bench lines\n")
out.append("Default (`python bench.py --steps 5 --warmup 1`): 32768 reads per step, ~8k events / 5445 k-mers each, events resident in HBM, recalibration on the device.\n")
out.append("```json\n" + line("bench_default.json") + "\n```\n")
out.append("From raw signal (`--from-raw 1`: scrappie event detection + MoM scalings on the device in front of the same pass; `cpu_baseline.whole_function` = the reference's own `SquiggleRead` from raw + `calculate_methylation_for_read`):\n")
out.append("```json\n" + line("bench_from_raw.json") + "\n```\n")
out.append("BASELINE configs[2], eventalign from raw signal (`python tests/bench_eventalign.py --pool 256 --tile 80 --cpu-sample 256`; 20480 reads per step; CPU leg = the reference's own `SquiggleRead` + `align_read_to_ref`, OpenMP over reads):\n")
out.append("```json\n" + line("bench_eventalign.json") + "\n```\n")
out.append("BASELINE configs[3], variants screening shape (`python tests/bench_variants.py`; CPU leg = the reference's own `profile_hmm_score`):\n")
out.append("```json\n" + line("bench_variants.json") + "\n```\n")
out.append("## rocprofv3 --kernel-trace --stats\n")
out.append("`rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --cpu-sample 0` (6 launches incl. warm-up), summarised with `profiles/summarize_rocpd.py`; the average duration of `np_event_align_kernel` agrees with `roofline.avg_launch_ms` of the bench line (HIP events inside bench.py):\n")
out.append(open(os.path.join(R, "trace.md")).read())
out.append("\nSame with `--from-raw 1 --steps 3`:\n")
out.append(open(os.path.join(R, "trace_raw.md")).read())
out.append("\n`rocprofv3 --kernel-trace --stats -- python tests/bench_eventalign.py --pool 256 --tile 80 --cpu-sample 0` (20480 reads per launch):\n")
out.append(open(os.path.join(R, "trace_ea.md")).read())
out.append("\n## PMC\n")
out.append("Counter corrections are measured, not assumed: `tools/hbm_counter_calib.hip` streams 1 GiB once per kernel under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (`profiles/collect_counter_calib.sh`): FETCH_SIZE reports 0.5000x for coalesced 4 B/lane and 16 B/lane reads alike (-> x2), WRITE_SIZE 1.0000x for coalesced 4 B/lane and 8 B/lane stores (-> x1).\n")
out.append("`profiles/r01_pmc.json` (kernel A / B / glue, `profiles/collect_pmc.sh` + `collect_pmc_lds.sh` -> `pmc_summary.py`, 8192 reads per launch): kernel A 86.0 VALU + 44.7 SALU wave-instructions per band, HBM traffic 0.754 MB fetched + 0.594 MB written per read (algorithmic figure of SURVEY 8d: 1.45 MB/read).\n")
ea = [l for l in open(os.path.join(R, "pmc_ea.txt")).read().splitlines() if l and l[0].isupper()]
out.append("Eventalign chain kernel (`profiles/collect_pmc_eventalign.sh`, 8192 reads = 1.497 M segments per launch), sums over all SEs per launch:\n")
out.append("```\n" + "\n".join(ea) + "\n```\n")
out.append("i.e. ~20.8 k VALU + ~12.6 k SALU wave-instructions per segment (fill: ~212 sweep steps x ~95 VALU for two k-mer blocks; the back-track is scalar), ~200 LDS instructions per segment (staging + walk), ~68 VMEM reads (event blocks, staging, path tail) and ~228 VMEM writes (one 128-byte back-pointer line per sweep step, path flushes, output rows) per segment.\n")
# from-raw traffic table
tot = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(G + "/rawpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(np_\w+)", r["Kernel_Name"])
        if m:
            tot[(m.group(1), r["Counter_Name"])] += float(r["Counter_Value"]); n[(m.group(1), r["Counter_Name"])].add(r["Dispatch_Id"])
if tot:
    out.append("\n## HBM traffic per kernel, from raw signal\n")
    out.append("`bash profiles/collect_pmc_from_raw.sh` (`bench.py --from-raw 1 --pool 512 --tile 16`: 8192 reads = 398 M raw samples per launch; FETCH_SIZE x2, WRITE_SIZE x1 as calibrated above; the HMM kernel is summed over its size classes):\n")
    out.append("| kernel | fetched MB | written MB |\n|---|---|---|")
    for k in sorted({k[0] for k in tot}):
        f = tot.get((k, "FETCH_SIZE"), 0) / max(1, len(n.get((k, "FETCH_SIZE"), [1]))) * 1024 * 2 / 1e6
        w = tot.get((k, "WRITE_SIZE"), 0) / max(1, len(n.get((k, "WRITE_SIZE"), [1]))) * 1024 / 1e6
        out.append("| `%s` | %.0f | %.0f |" % (k, f, w))
    out.append("\nThe raw samples are 1.59 GB and the two t-statistics 3.18 GB per launch: event detection moves ~22 GB per 8192 reads in ~8 ms, i.e. it runs at ~2.7 TB/s and is the one HBM-bound stage of the chain; the parallel peak walk re-reads the t-statistics 2.9 times (64 lanes of a wave stream 64 distant segments in 64-byte blocks, each with a warm-up overlap) and writes its peak positions 4 bytes at a time -- the first candidate for fusion with the t-statistic kernel.\n")
print("\n".join(out))
