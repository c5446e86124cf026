#!/usr/bin/env python3
"""Per-kernel PMC summary from the rocprofv3 --pmc passes of profiles/collect_pmc.sh.

    pmc_summary.py gpurun_out reads_per_launch bands_per_read > profiles/rNN_pmc.json

Sums every counter per kernel over the launches of the run and reports per-launch / per-read figures for the three
hot kernels.  FETCH_SIZE / WRITE_SIZE: KB -> x1024 bytes, then the gfx950 corrections measured on known byte counts in this
kernel's own access widths (tools/hbm_counter_calib.hip, profiles/collect_counter_calib.sh: 1 GiB streamed once per kernel):
FETCH_SIZE reports exactly 1/2 of the bytes of coalesced 4 B/lane AND 16 B/lane reads -> x2 (the guide's figure for wide reads
holds for narrow ones too); WRITE_SIZE is exact for coalesced 4 B/lane and 8 B/lane stores -> x1.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

root, reads, bands = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
FETCH_CORR, WRITE_CORR = 2.0, 1.0      # measured, see the docstring
tot = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
dur = defaultdict(float)
for f in sorted(glob.glob(root + "/pmc*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("event_align", "hmm_forward", "recalibrate", "build_map"):
            if key in k:
                tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[(key, f)].add(r["Dispatch_Id"])
out = {}
for key, c in tot.items():
    n = max(len(v) for (k, f), v in launches.items() if k == key)
    d = {"launches_per_pass": n}
    for name, v in sorted(c.items()):
        d[name + "_per_launch"] = v / n
    if key == "event_align":
        per_read = lambda x: c.get(x, 0.0) / n / reads
        d.update(reads_per_launch=reads,
                 fetch_bytes_per_read=per_read("FETCH_SIZE") * 1024 * FETCH_CORR, write_bytes_per_read=per_read("WRITE_SIZE") * 1024 * WRITE_CORR,
                 fetch_size_correction=FETCH_CORR, write_size_correction=WRITE_CORR,
                 valu_insts_per_read=per_read("SQ_INSTS_VALU"), salu_insts_per_read=per_read("SQ_INSTS_SALU"),
                 vmem_rd_per_read=per_read("SQ_INSTS_VMEM_RD"), vmem_wr_per_read=per_read("SQ_INSTS_VMEM_WR"),
                 valu_insts_per_band=per_read("SQ_INSTS_VALU") / bands, salu_insts_per_band=per_read("SQ_INSTS_SALU") / bands,
                 valu_per_band=per_read("SQ_INSTS_VALU") / bands, salu_per_band=per_read("SQ_INSTS_SALU") / bands,
                 branch_per_band=per_read("SQ_INSTS_BRANCH") / bands if "SQ_INSTS_BRANCH" in c else None)
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            d.update(wait_inst_any_frac=c.get("SQ_WAIT_INST_ANY", 0.0) / wc, wait_any_frac=c.get("SQ_WAIT_ANY", 0.0) / wc,
                     active_inst_any_frac=c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc)
        if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
            # rocprof's VALUBusy: 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / GRBM_GUI_ACTIVE with the per-XCD maximum of GUI_ACTIVE;
            # the csv sums GUI_ACTIVE over the 8 XCDs, hence / 8
            d["valu_busy_pct"] = 100.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_BUSY_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["valu_active_over_busy"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"]
    if "SQ_WAVE_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        d["mean_waves_in_flight_per_busy_cycle"] = c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"]
    out[key] = d
out["note"] = ("FETCH_SIZE/WRITE_SIZE per launch: raw rocprofv3 KB; *_bytes_per_read: x1024 and corrected (FETCH x2, WRITE x1, "
               "calibrated with tools/hbm_counter_calib.hip); counters are summed over all XCDs/SEs as rocprofv3 reports them")
json.dump(out, sys.stdout, indent=1)
print()
