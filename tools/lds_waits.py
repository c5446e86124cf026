#!/usr/bin/env python3
"""Per basic block of a kernel in a gfx950 assembly listing: vector instructions, LDS reads, `s_waitcnt lgkmcnt` count and scratch
(spill) accesses -- how many table look-ups a step of kernel B waits for one at a time.   lds_waits.py file.s <kernel-name-substring>"""
import sys
L = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
start = next(i for i, l in enumerate(L) if l.startswith("_Z") and pat in l.split(":")[0])
end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
K = L[start:end]
labs = [i for i, l in enumerate(K) if l.startswith(".LBB")] + [len(K)]
tot = [0, 0, 0, 0]
for a, b in zip(labs, labs[1:]):
    blk = [l.strip() for l in K[a:b]]
    nv = sum(1 for l in blk if l.startswith("v_")); nds = sum(1 for l in blk if l.startswith("ds_read"))
    nw = sum(1 for l in blk if l.startswith("s_waitcnt") and "lgkmcnt" in l); nsc = sum(1 for l in blk if l.startswith("scratch_"))
    inloop = "Loop" in " ".join(K[a:a + 3])
    if nv > 20 or nsc:
        print("%-10s %s valu %4d ds_read %3d lgkm-waits %3d scratch %2d" % (K[a].split(":")[0], "loop" if inloop else "    ", nv, nds, nw, nsc))
