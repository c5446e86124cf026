#!/usr/bin/env python3
"""Issue-cycle estimate of a stretch of gfx950 assembly from the calibrated per-class costs (profiles/r04_valu_calibration.json:
tools/valu_rates under rocprofv3 --pmc): every VALU instruction is priced at the measured cycles per wave-instruction and SIMD of
its class.  Usage:
    issue_cost.py file.s <kernel-name-substring> [first_line last_line]     (line numbers relative to the kernel's label)
Without a range: the whole kernel, plus a per-basic-block table (blocks >= 40 instructions)."""
import json
import os
import re
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAL = {c["name"]: c["cycles_per_inst_simd"] for c in json.load(open(os.path.join(ROOT, "profiles", "r04_valu_calibration.json")))["classes"]}
FAST = (CAL["v_add_f32"] + CAL["v_sub_f32"] + CAL["v_mul_f32"] + CAL["v_and_b32"] + CAL["v_or_b32"] + CAL["v_mov_b32"] + CAL["v_add_u32"] + CAL["v_sub_u32"]) / 8
FMA = (CAL["v_fma_f32"] + CAL["v_fma_f32 3src"]) / 2
SLOW = (CAL["v_cvt_f64_f32"] + CAL["v_add_f64"] + CAL["v_max3_f32"] + CAL["v_cmp_eq_f32"] + CAL["v_lshl_or_b32"] + CAL["cndmask e64 s"] + CAL["v_max_f32"] + CAL["v_cvt_u32_f32"]
        + CAL["v_mov_dpp ror"] + CAL["v_addc vcc"] + CAL["v_lshlrev_b32"]) / 11
PK = (CAL["v_pk_fma_f32"] + CAL["v_pk_mul_f32"] + CAL["v_pk_add_f32"]) / 3
LANE = CAL["v_readlane"]
FAST_OPS = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_add_u32", "v_sub_u32",
            "v_subrev_u32", "v_not_b32", "v_mac_f32", "v_fmac_f32", "v_add_co_u32", "v_accvgpr")


def klass(op, line):
    if not op.startswith("v_"):
        return None
    if "dpp" in line or "_dpp" in op or "sdwa" in op:
        return "slow"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_pk_"):
        return "pk"
    base = re.sub(r"_e(32|64)$", "", op)
    if base.startswith(("v_fma_f32", "v_fmac_f32", "v_mad_f32")):
        return "fma"
    if base in FAST_OPS or base.startswith("v_accvgpr"):
        # a fast opcode with a scalar-register source was measured in the slow class (v_add_f32 s,v / v_mov_b32 v,s: 4.3 cycles against 2.5)
        srcs = line.split(None, 1)[1].split(",")[1:] if len(line.split(None, 1)) > 1 else []
        if any(re.match(r"\s*(s\d+|s\[|vcc|exec|ttmp)", x) for x in srcs):
            return "slow"
        return "fast"
    return "slow"


COST = dict(fast=FAST, fma=FMA, slow=SLOW, pk=PK, lane=LANE)


def price(lines):
    n = Counter(); ops = Counter()
    other = Counter()
    for l in lines:
        t = l.strip()
        if not t or t[0] in ".;" or t.endswith(":"):
            continue
        op = t.split()[0]
        k = klass(op, t)
        if k:
            n[k] += 1; ops[re.sub(r"_e(32|64)$", "", op)] += 1
        elif op.startswith("ds_"):
            other["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            other["vmem"] += 1
        elif op.startswith("s_"):
            other["salu"] += 1
    cyc = sum(n[k] * COST[k] for k in n)
    return n, other, cyc, ops


def loop_mix(sfile, pat):
    """Class mix of the VALU instructions inside the loops of the kernel / function whose mangled name contains `pat` (blocks annotated
    "in Loop" / "Loop Header" by the assembler listing), unweighted: {"n": {class: count}, "mean_cycles": cycles per wave-instruction}."""
    L = open(sfile).read().splitlines()
    start = next(i for i, l in enumerate(L) if l.startswith("_Z") and pat in l.split(":")[0])
    end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i] or "s_setpc_b64" in L[i] or L[i].startswith(".Lfunc_end"))
    K = L[start:end]
    labs = [i for i, l in enumerate(K) if l.startswith(".LBB") or l.startswith("; %bb.")] + [len(K)]
    tot = Counter()
    for a, b in zip(labs, labs[1:]):
        head = " ".join(K[a:a + 3])
        if "Loop" not in head:
            continue
        n, o, cyc, ops = price(K[a:b])
        tot.update(n)
    nv = sum(tot.values())
    return dict(n=dict(tot), mean_cycles=round(sum(tot[k] * COST[k] for k in tot) / max(nv, 1), 3), class_cycles={k: round(v, 2) for k, v in COST.items()})


def main():
    L = open(sys.argv[1]).read().splitlines()
    pat = sys.argv[2]
    start = next(i for i, l in enumerate(L) if l.startswith("_Z") and pat in l.split(":")[0])
    end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i] or "s_setpc_b64" in L[i] or L[i].startswith(".Lfunc_end"))
    K = L[start:end]
    print("class costs (cycles per wave-instruction and SIMD): fast %.2f  fma %.2f  slow %.2f  packed %.2f  lane %.2f" % (FAST, FMA, SLOW, PK, LANE))
    if len(sys.argv) >= 5:
        a, b = int(sys.argv[3]), int(sys.argv[4])
        n, o, cyc, ops = price(K[a:b])
        print("lines %d..%d: valu %d %s  other %s  issue cycles %.0f" % (a, b, sum(n.values()), dict(n), dict(o), cyc))
        print("top opcodes:", ops.most_common(14))
        return
    labs = [i for i, l in enumerate(K) if l.startswith(".LBB") or l.startswith("; %bb.")] + [len(K)]
    for a, b in zip(labs, labs[1:]):
        n, o, cyc, ops = price(K[a:b])
        if sum(n.values()) >= 40:
            print("%6d %6d  valu %4d %-58s other %-36s cycles %6.0f" % (a, b, sum(n.values()), dict(n), dict(o), cyc))
            if os.environ.get("NP_ISSUE_OPS"):
                print("        ", ops.most_common(18))


if __name__ == "__main__":
    main()
