"""Per-unit counter figures of the shipped kernels for the bench lines' `roofline.traffic` / `issue` fields, read from the committed
summary of this round's counter passes (profiles/r06_pmc.json <- profiles/collect_r06_pmc.sh + profiles/pmc_summary_r06.py).  The passes
run at a launch size they finish at (8 192 reads); a bench line scales the PER-UNIT figures (bytes per band / call / segment) by the
units its own launch processed, and says so in `source`."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILE = os.path.join("profiles", "r06_pmc.json")


def family(name):
    """dict of the family's per-unit figures (without the raw per-step totals), or {} when the summary is missing."""
    try:
        d = json.load(open(os.path.join(ROOT, FILE)))[name]
    except Exception:  # noqa: BLE001
        return {}
    out = {k: v for k, v in d.items() if not k.endswith("_per_step") and k != "loop_class_mix"}
    out["source"] = FILE + " (rocprofv3 --pmc over the shipped kernel at 8 192 reads per launch, profiles/collect_r06_pmc.sh; per-unit figures x this launch's units)"
    return out


def traffic(name, unit, n_units):
    """HBM bytes of a launch of `n_units` units (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 corrections of tools/hbm_counter_calib.hip), or None."""
    d = family(name)
    f, w = d.get("fetch_bytes_per_" + unit), d.get("write_bytes_per_" + unit)
    return None if f is None or w is None else int((f + w) * n_units)


def issue(name, unit, simd_cycles_per_unit):
    """Vector-issue figures of a launch that took `simd_cycles_per_unit` (launch time x 2.4 GHz x 1024 SIMDs / units): instructions per unit from
    the counters, priced at the calibrated issue cycles (profiles/r04_valu_calibration.json).  floor: every instruction at the fastest
    class's cost (a lower bound of the vector port's busy fraction); priced: by the class mix of the kernel's loops (an estimate)."""
    d = family(name)
    v = d.get("valu_per_" + unit)
    if v is None or simd_cycles_per_unit <= 0:
        return None
    fast = json.load(open(os.path.join(ROOT, FILE))).get("fast_class_cycles_per_instruction", 2.49)
    out = dict(kind="counters x calibrated issue cycles", valu_per_unit=v, salu_per_unit=d.get("salu_per_" + unit), lds_per_unit=d.get("lds_per_" + unit),
               unit=unit, simd_cycles_per_unit=round(simd_cycles_per_unit, 1), valu_issue_floor=round(v * fast / simd_cycles_per_unit, 3), source=d["source"])
    if d.get("valu_issue_priced") and d.get("valu_issue_floor"):
        out["valu_issue_priced"] = round(out["valu_issue_floor"] * d["valu_issue_priced"] / d["valu_issue_floor"], 3)
        out["note"] = ("the vector port's busy fraction lies between valu_issue_floor and 1; valu_issue_priced charges every instruction its class's cost in an "
                       "ISOLATED stream (profiles/r04_valu_calibration.json) -- mixed streams overlap by ~10 % (cmp + cndmask alternating: 3.8 cycles against "
                       "4.4), so a value at or above 1 reads: this launch runs at the issue bound of its instruction mix")
    return out


def roofline_issue(name, unit, simd_cycles_per_unit, kernel=None):
    """The roofline these kernels actually sit under: vector-instruction ISSUE.  peak = one wave64 VALU instruction
    per 2 cycles and SIMD (MI355X_MICROARCH.md; the calibration of profiles/r04_valu_calibration.json measures 2.49 for the fastest
    class and 4.4 for fp64 / compares / selects / conversions, so a kernel of mixed classes saturates the port well below frac 1);
    achieved = the counters' vector instructions per unit x 2 cycles / the SIMD-cycles this launch spent per unit."""
    d = family(name)
    v = d.get("valu_per_" + unit)
    if v is None or simd_cycles_per_unit <= 0:
        return None
    frac = v * 2.0 / simd_cycles_per_unit
    out = dict(bound="valu_issue", achieved=round(frac, 4), peak=1.0, unit="fraction of the vector port's issue slots (2 cycles per wave64 VALU instruction)",
               frac=round(frac, 4), peak_cycles_per_inst=2, valu_per_unit=v, per=unit, simd_cycles_per_unit=round(simd_cycles_per_unit, 1),
               source=d["source"])
    if kernel:
        out["kernel"] = kernel
    return out
