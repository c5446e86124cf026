#!/usr/bin/env python3
"""Summary of the round-3 counter passes (profiles/collect_r03_pmc.sh) -> profiles/r03_pmc.json.

    pmc_summary_r03.py gpurun_out/<tag>

Per kernel family (event_align = np_event_align_kernel, hmm_forward = every size class of np_hmm_forward_kernel, chain =
np_eventalign_chain2_kernel, or np_eventalign_chain_kernel with ea_kernel = 1): counter totals per launch, instructions per unit of work (band / HMM call and cell-state / segment and
lattice cell; the units come from tools/pmc_workload.py's own JSON line), VALUBusy, and HBM bytes per unit.
Units of the counters (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles);
SQ_BUSY_CYCLES is summed over the shader engines; GRBM_GUI_ACTIVE is summed over the 8 XCDs in the csv.
    VALUBusy = 100 * SQ_ACTIVE_INST_VALU * 4 / N_SIMD / GRBM_GUI_ACTIVE(per XCD)            (rocprof's gfx94x formula)
FETCH_SIZE / WRITE_SIZE: KB -> x1024 bytes, then the gfx950 corrections calibrated for these kernels' access widths in round 1
(tools/hbm_counter_calib.hip): FETCH x2 (the counter tallies 128-byte requests at 64 bytes), WRITE x1.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
FETCH_CORR, WRITE_CORR, N_SIMD = 2.0, 1.0, 1024
FAM = (("event_align", "np_event_align_kernel"), ("hmm_forward", "np_hmm_forward_kernel"), ("chain", "np_eventalign_chain"),
       ("recalibrate", "np_recalibrate_kernel"), ("build_map", "np_build_map_kernel"), ("cm_items", "np_cm_items_kernel"))
units = {}
try:
    for line in open(root + "/units.json"):
        if line.startswith("{"):
            units = json.loads(line)
except OSError:
    pass
reps = units.get("reps", 2)
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
rows = defaultdict(list)
for f in sorted(glob.glob(root + "/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for key, pat in FAM:
            if pat in r["Kernel_Name"]:
                rows[(key, f)].append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
for (key, f), lst in rows.items():
    ids = sorted({d for d, _, _ in lst})
    # the workload runs the call-methylation step first, then the eventalign step, which launches kernel A again on ITS batch (more
    # events per read): only the call-methylation launches (the first `reps` dispatches) are normalised by that step's bands
    keep = set(ids[:reps]) if key == "event_align" else set(ids)
    for d, name, v in lst:
        if d in keep:
            tot[key][name] += v
            disp[key][name].add((f, d))
out = {"source": "rocprofv3 --kernel-trace --pmc over tools/pmc_workload.py (profiles/collect_r03_pmc.sh), gpurun tag %s" % root.rstrip("/").split("/")[-1],
       "units": units, "fetch_size_correction": FETCH_CORR, "write_size_correction": WRITE_CORR}
cm, ea = units.get("call_methylation", {}), units.get("eventalign", {})
for key, c in tot.items():
    # counters are totals over `reps` steps; a "launch" here = one step's launches of the family (all size classes for kernel B)
    d = {"dispatches_per_step": {k: len(v) / reps for k, v in disp[key].items()}}
    per = {k: v / reps for k, v in c.items()}
    for name, v in sorted(per.items()):
        d[name + "_per_step"] = v
    if "SQ_ACTIVE_INST_VALU" in per and per.get("GRBM_GUI_ACTIVE"):
        d["valu_busy_pct"] = round(100.0 * per["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMD / (per["GRBM_GUI_ACTIVE"] / 8.0), 2)
    if "SQ_ACTIVE_INST_SCA" in per and per.get("GRBM_GUI_ACTIVE"):
        d["salu_busy_pct"] = round(100.0 * per["SQ_ACTIVE_INST_SCA"] * 4.0 / N_SIMD / (per["GRBM_GUI_ACTIVE"] / 8.0), 2)
    if "SQ_ACTIVE_INST_LDS" in per and per.get("GRBM_GUI_ACTIVE"):
        d["lds_inst_busy_pct"] = round(100.0 * per["SQ_ACTIVE_INST_LDS"] * 4.0 / N_SIMD / (per["GRBM_GUI_ACTIVE"] / 8.0), 2)
    if per.get("SQ_WAVE_CYCLES"):
        wc = per["SQ_WAVE_CYCLES"]
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
            if nm in per:
                d[nm.lower() + "_over_wave_cycles"] = round(per[nm] / wc, 4)
    if per.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in per:
        d["lds_bank_conflict_frac"] = round(per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"], 4)
        if per.get("SQ_INSTS_LDS"):
            d["lds_cycles_per_lds_inst"] = round(per["SQ_LDS_IDX_ACTIVE"] / per["SQ_INSTS_LDS"], 3)
    fb = per.get("FETCH_SIZE", 0.0) * 1024 * FETCH_CORR if "FETCH_SIZE" in per else None
    wb = per.get("WRITE_SIZE", 0.0) * 1024 * WRITE_CORR if "WRITE_SIZE" in per else None
    if fb is not None:
        d["fetch_bytes_per_step"] = fb
    if wb is not None:
        d["write_bytes_per_step"] = wb
    unit = None
    if key == "event_align" and cm.get("bands"):
        unit = ("band", cm["bands"]); d["reads_per_launch"] = cm["reads"]; d["bands_per_read"] = cm["bands"] / cm["reads"]
        d["algo_bytes_per_band"] = cm["algo_bytes_align"] / cm["bands"]
    elif key == "hmm_forward" and cm.get("hmm_calls"):
        unit = ("call", cm["hmm_calls"]); d["cell_states_per_call"] = cm["hmm_cell_states"] / cm["hmm_calls"]
        d["algo_bytes_per_call"] = cm["hmm_algo_bytes"] / cm["hmm_calls"]
    elif key == "chain" and ea.get("segments"):
        unit = ("segment", ea["segments"]); d["lattice_cells_per_segment"] = ea["lattice_cells"] / ea["segments"]
        d["lattice_rows_per_segment"] = ea["lattice_rows"] / ea["segments"]; d["reads_per_launch"] = ea["reads"]
    if unit:
        nm, n = unit
        for cname, label in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_LDS", "lds"), ("SQ_INSTS_VMEM_RD", "vmem_rd"),
                             ("SQ_INSTS_VMEM_WR", "vmem_wr"), ("SQ_INSTS_SMEM", "smem")):
            if cname in per:
                d["%s_per_%s" % (label, nm)] = round(per[cname] / n, 3)
        if fb is not None:
            d["fetch_bytes_per_%s" % nm] = round(fb / n, 3)
        if wb is not None:
            d["write_bytes_per_%s" % nm] = round(wb / n, 3)
        if per.get("GRBM_GUI_ACTIVE"):
            d["simd_cycles_per_%s" % nm] = round(per["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMD / n, 2)
    out[key] = d
json.dump(out, sys.stdout, indent=1)
print()
