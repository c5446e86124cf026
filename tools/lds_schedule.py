#!/usr/bin/env python3
"""Order of LDS look-ups (R), global loads (G), scratch accesses (S!) and waits in the large basic blocks of one kernel of a
hipcc -S listing -- how many table look-ups the forward kernel keeps in flight.  Usage: lds_schedule.py file.s kernel-name-substring [min_lines]"""
import sys
L = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
minl = int(sys.argv[3]) if len(sys.argv) > 3 else 600
start = next(i for i, l in enumerate(L) if l.startswith("_Z") and pat in l.split(":")[0])
end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
L = L[start:end]
labs = [i for i, l in enumerate(L) if l.startswith(".LBB") or l.startswith("; %bb.")]
for a, b in zip(labs, labs[1:] + [len(L)]):
    if b - a > minl:
        seq, nv = [], 0
        for l in L[a:b]:
            t = l.strip().split()
            if not t:
                continue
            if t[0].startswith("v_"):
                nv += 1
            if t[0].startswith("ds_read"):
                seq.append("R")
            elif t[0].startswith("global_load"):
                seq.append("G")
            elif t[0].startswith("scratch"):
                seq.append("S!")
            elif t[0] == "s_waitcnt":
                seq.append("W(" + "".join(t[1:]) + ")")
        print(a, b, "valu", nv, " ".join(seq))
