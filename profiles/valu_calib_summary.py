#!/usr/bin/env python3
"""VALU counter calibration: tools/valu_rates under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` -> per instruction class: wave-instructions counted, SQ_ACTIVE_INST_VALU per instruction, the
kernel's duration per instruction and SIMD (ns and cycles at the clock the trace implies), and what rocprof's VALUBusy formula
(100 * SQ_ACTIVE_INST_VALU * 4 / SIMDs / GRBM_GUI_ACTIVE per XCD) would print for a stream that does nothing but issue that class.

    valu_calib_summary.py <results.db> tools/valu_rates.hip > profiles/r04_valu_calibration.json
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict

db, src = sys.argv[1], sys.argv[2]
names = {}
for m in re.finditer(r'run<(\d+)>\("([^"]+)"', open(src).read()):
    names.setdefault(int(m.group(1)), m.group(2))
c = sqlite3.connect(db)
rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size, workgroup_size from counters_collection").fetchall()
per = defaultdict(dict)
for did, kn, cn, v, dur, grid, wg in rows:
    d = per[did]
    d["kernel"] = kn; d[cn] = d.get(cn, 0.0) + float(v); d["duration_ns"] = dur; d["waves"] = grid // 64
N_SIMD, CLK = 1024, 2.4e9
out = {"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- tools/valu_rates",
       "note": "each kernel: 16000 iterations x 16 instructions of the class per wave (+ loop overhead), 2 KB of LDS per one-wave workgroup so that 32 waves "
               "co-reside on a CU (the first calibration of this round, gpurun r04a, declared 16 KB: ten waves per CU in four unequal rounds, which "
               "inflated every figure ~1.5 x -- 3.7 / 6.3 cycles; superseded).  `classes` = 8 waves per SIMD (the issue limit); `by_occupancy` = the "
               "same streams at 4, 2 and 1 waves per SIMD (what one wave alone can issue).  The short warm-up launch of every class (10 iterations) is skipped.  active_per_inst = SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU: what the counter adds per wave-instruction "
               "(quad-cycles).  cycles_per_inst_simd = kernel duration x 2.4 GHz x SIMDs / SQ_INSTS_VALU: the issue time the class really takes.  "
               "valubusy_formula_pct: rocprof's gfx94x VALUBusy for this saturated stream -- 100 % would be a utilisation metric, anything else is not.",
       "classes": [], "by_occupancy": []}
for did in sorted(per):
    d = per[did]
    m = re.search(r"k<(\d+)>", d["kernel"])
    if not m or d.get("SQ_INSTS_VALU", 0) < 1e8:
        continue
    op = int(m.group(1))
    insts, act = d["SQ_INSTS_VALU"], d.get("SQ_ACTIVE_INST_VALU", 0.0)
    gui = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wps = d["waves"] // N_SIMD
    out["classes" if wps == 8 else "by_occupancy"].append(dict(op=op, waves_per_simd=wps, name=names.get(op, "?"), insts_valu=insts, active_inst_valu=act, active_per_inst=round(act / insts, 3),
                               duration_ms=round(d["duration_ns"] / 1e6, 3), ns_per_inst_simd=round(d["duration_ns"] * N_SIMD / insts, 3),
                               cycles_per_inst_simd=round(d["duration_ns"] * 1e-9 * CLK * N_SIMD / insts, 2),
                               gui_cycles_per_inst_simd=round(gui * N_SIMD / insts, 2) if gui else None,
                               valubusy_formula_pct=round(100.0 * act * 4.0 / N_SIMD / gui, 1) if gui else None,
                               sq_busy_cycles=d.get("SQ_BUSY_CYCLES")))
json.dump(out, sys.stdout, indent=1)
print()
