#!/usr/bin/env python3
"""Soak check (manual, through gpurun): eventalign rows of N full-size synthetic reads (5450 bases, both strands) from the device
chain against the reference's own align_read_to_ref (oracle/_ref/libnp_ref_full.so), every row.   python tests/gpu_soak_eventalign.py [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import load_models
from oracle.ref_full import FullRef
from nanopolish_amd import api
from nanopolish_amd.api import Context
from nanopolish_amd.pipeline import build_host_batch_records, CallMethylationBatch
from nanopolish_amd.synth import synth_raw

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
models = load_models()
ctx = Context(0); ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
recs = []
for rid in range(1000, 1000 + N):
    rd = synth_raw(rid, models["nucleotide"], L=5450)
    ref = api.reverse_complement("nucleotide", rd["seq"]) if rd["rc"] else rd["seq"]
    recs.append(dict(seq=rd["seq"], raw=rd["raw"], rc=rd["rc"], pos=0, cigar=api.cigar_words([("M", len(rd["seq"]))]), contig=ref))
hb = build_host_batch_records(models, recs, "")
b = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True, workload="eventalign")
b.step()
res = b.eventalign_results()
F = FullRef()
bad = rows = 0
for i, r in enumerate(recs):
    fr = F.read("r%d" % i, r["seq"], r["raw"])
    ea = fr.eventalign(r["rc"], 0, r["cigar"], r["contig"], r["contig"]) if fr.n_events else None
    g = res[i]
    ok = (ea is None and len(g["event_idx"]) == 0) or (ea is not None and np.array_equal(ea["ref_position"], g["ref_position"]) and
                                                       np.array_equal(ea["event_idx"], g["event_idx"]) and np.array_equal(ea["hmm_state"], g["hmm_state"]))
    bad += not ok; rows += len(g["event_idx"])
    fr.close()
print("reads %d rows %d mismatching reads %d statuses %s" % (N, rows, bad, sorted(set(x["status"] for x in res))))
sys.exit(1 if bad else 0)
