#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE ITSELF (oracle/_ref/libnp_ref.so, built in place from
/root/reference by `make -C oracle ref`).  Run in the build container only:

    python tests/gen_golden.py

Outputs (committed, small):
  models_r9.4_450bps.npz   pore-model tables (nucleotide + cpg 6-mer template) as the reference loads them
  golden_tables.npz        flogsum_lookup, alphabet/string KATs, emission KATs, transitions, MoM scalings
  golden_reads.npz         seeded synthetic reads -> adaptive_banded_simple_event_align pairs,
                           call-methylation work items -> profile_hmm_score (unmeth/meth), eventalign
                           segments -> profile_hmm_align states, profile_hmm_score_set values
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import Oracle, RefOracle  # noqa: E402
from cases import methylation_jobs, eventalign_segments, synth_read, K, HAF_PRE, HAF_POST  # noqa: E402

GOLD = os.path.join(HERE, "golden")


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = RefOracle()
    orc = Oracle()   # only used for the read-level glue that has no compiled-reference counterpart
    nuc = ref.model("nucleotide")
    cpg = ref.model("cpg")
    np.savez_compressed(os.path.join(GOLD, "models_r9.4_450bps.npz"),
                        **{a + "_" + f: m[f] for a, m in (("nucleotide", nuc), ("cpg", cpg))
                           for f in ("level_mean", "level_stdv", "level_log_stdv")})

    # ---------------- tables & KATs ----------------
    rng = np.random.default_rng(12345)
    out = {"flogsum_table": ref.flogsum_table()}
    a = rng.uniform(-300, 0, 4000).astype(np.float32)
    b = (a + rng.uniform(-20, 20, 4000)).astype(np.float32)
    a[::97] = -np.inf; b[::89] = -np.inf
    out["addlogs_a"] = a; out["addlogs_b"] = b
    out["addlogs_out"] = np.array([ref.add_logs(float(x), float(y)) for x, y in zip(a, b)], np.float32)

    alph = {"nucleotide": "ACGT", "cpg": "ACGMT", "gpc": "ACGMT", "dam": "ACGMT", "dcm": "ACGMT"}
    kat = []
    for name, sym in alph.items():
        for t in range(60):
            n = int(rng.integers(1, 40))
            s = "".join(rng.choice(list("ACGT"), n))
            if name != "nucleotide":
                s = ref.methylate(name, s) if t % 2 else s
            kat.append((name, s, ref.reverse_complement(name, s), ref.methylate(name, s.replace("M", "C") if name != "dam" else s.replace("M", "A")),
                        ref.unmethylate(name, s), "".join("1" if ref.is_motif_match(name, s, i) else "0" for i in range(max(len(s) - 1, 0))),
                        ref.kmer_rank(name, s[:6]) if len(s) >= 6 else -1))
    # the reference's own unit-test vectors (src/test/nanopolish_test.cpp:27-265)
    out["kat_alphabet"] = np.array([k[0] for k in kat]); out["kat_in"] = np.array([k[1] for k in kat])
    out["kat_rc"] = np.array([k[2] for k in kat]); out["kat_meth"] = np.array([k[3] for k in kat])
    out["kat_unmeth"] = np.array([k[4] for k in kat]); out["kat_motif"] = np.array([k[5] for k in kat])
    out["kat_rank6"] = np.array([k[6] for k in kat], np.int64)

    # emission: the reference's "scalings" test pattern (src/test/nanopolish_test.cpp:277-325), drift included
    ranks = rng.integers(0, 4096, 200)
    levels = rng.uniform(60, 130, 200).astype(np.float32)
    out["emis_rank"] = ranks; out["emis_level"] = levels
    out["emis_scal"] = np.array([10.0, 1.2, 0.0, 1.3])
    out["emis_lp"] = np.array([ref.log_probability_match_r9("nucleotide", int(r), float(x), 10.0, 1.2, 0.0, 1.3)
                               for r, x in zip(ranks, levels)], np.float32)
    cr = rng.integers(0, 15625, 200)
    out["emis_cpg_rank"] = cr
    out["emis_cpg_lp"] = np.array([ref.log_probability_match_r9("cpg", int(r), float(x), -3.5, 0.97, 0.0, 1.11)
                                   for r, x in zip(cr, levels)], np.float32)
    np.savez_compressed(os.path.join(GOLD, "golden_tables.npz"), **out)

    # ---------------- reads ----------------
    g = {}
    cases = [(0, 400), (1, 400), (2, 900), (3, 900), (4, 1500), (5, 1500), (6, 3000), (7, 150)]
    g["read_ids"] = np.array([c[0] for c in cases]); g["read_L"] = np.array([c[1] for c in cases])
    for rid, L in cases:
        rd = synth_read(rid, nuc, L=L)
        p = "r%d_" % rid
        sh, sc = ref.estimate_scalings_mom(rd["seq"], rd["events"])
        pairs = ref.event_align(rd["events"], rd["seq"], sh, sc)
        g[p + "mom"] = np.array([sh, sc])
        g[p + "pairs"] = pairs
        if len(pairs) == 0:
            continue
        epb, jobs = methylation_jobs(orc, rd, pairs)
        g[p + "epb"] = np.array(epb)
        su, sm = [], []
        for j in jobs:
            su.append(ref.hmm_score("cpg", j["subseq"], j["rc_subseq"], rd["events"], j["e1"], j["e2"], j["stride"], j["rc"],
                                    rd["shift"], rd["scale"], rd["var"], epb, 1.0, HAF_PRE | HAF_POST))
            sm.append(ref.hmm_score("cpg", j["m_subseq"], j["rc_m_subseq"], rd["events"], j["e1"], j["e2"], j["stride"], j["rc"],
                                    rd["shift"], rd["scale"], rd["var"], epb, 1.0, HAF_PRE | HAF_POST))
        g[p + "job_first"] = np.array([j["first"] for j in jobs]); g[p + "job_e1"] = np.array([j["e1"] for j in jobs])
        g[p + "job_e2"] = np.array([j["e2"] for j in jobs])
        g[p + "score_unmeth"] = np.array(su, np.float32); g[p + "score_meth"] = np.array(sm, np.float32)
        # flag variants + indel bias on the first few jobs (variants uses flags 0 and bias .8/.9)
        fv = []
        for j in jobs[:6]:
            for flags in (0, HAF_PRE, HAF_POST):
                for bias in (1.0, 0.9):
                    fv.append(ref.hmm_score("cpg", j["subseq"], j["rc_subseq"], rd["events"], j["e1"], j["e2"], j["stride"], j["rc"],
                                            rd["shift"], rd["scale"], rd["var"], epb, bias, flags))
        g[p + "score_flagvar"] = np.array(fv, np.float32)
        if not rd["rc"]:
            epb2, segs = eventalign_segments(orc, rd, pairs)
            for si, sg in enumerate(segs[:6]):
                ev, km, lf, st = ref.hmm_align("nucleotide", sg["seq"], None, rd["events"], sg["e1"], sg["e2"], 1, 0,
                                               rd["shift"], rd["scale"], rd["var"], epb2)
                q = p + "seg%d_" % si
                g[q + "e"] = np.array([sg["e1"], sg["e2"]]); g[q + "start"] = np.array(rd["seq"].index(sg["seq"]))
                g[q + "event_idx"] = ev; g[q + "kmer_idx"] = km; g[q + "l_fm"] = lf; g[q + "state"] = st
            # profile_hmm_score_set: nucleotide sequence + its cpg-methylated alternative (variants with methylation types)
            ss = []
            for sg in segs[:4]:
                w = sg["seq"][:30]
                ss.append(ref.hmm_score_set([w, ref.methylate("cpg", w)], ["nucleotide", "cpg"], rd["events"],
                                            sg["e1"], sg["e1"] + 40, 1, 0, rd["shift"], rd["scale"], rd["var"], epb2, 0.9, 0))
            g[p + "score_set"] = np.array(ss, np.float32)
    np.savez_compressed(os.path.join(GOLD, "golden_reads.npz"), **g)

    # ---------------- f2: event detection on synthetic raw signal (reference's own scrappie objects) ----------------
    from nanopolish_amd.synth import synth_raw
    from oracle.oracle_py import ED_DEFAULTS, ED_RNA
    e = dict(read_ids=np.array([0, 1, 2]), read_L=np.array([600, 1500, 300]))
    for rid, L in zip(e["read_ids"], e["read_L"]):
        rd = synth_raw(int(rid), nuc, L=int(L))
        for tag, prm in (("dna", ED_DEFAULTS), ("rna", ED_RNA)):
            ev = ref.detect_events(rd["raw"], **prm)
            q = "r%d_%s_" % (rid, tag)
            for k2 in ("start", "length", "mean", "stdv"):
                e[q + k2] = ev[k2]
        e["r%d_raw_crc" % rid] = np.array([np.frombuffer(rd["raw"].tobytes(), np.uint32).sum(dtype=np.uint64)])
    np.savez_compressed(os.path.join(GOLD, "golden_events.npz"), **e)
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
