#!/usr/bin/env python3
"""Generates tests/golden/golden_reflevel.npz by running the REFERENCE'S OWN read-level code -- SquiggleRead::load_from_raw,
EventAlignmentRecord, calculate_methylation_for_read, create_modbam_record, align_read_to_ref -- compiled in place from
/root/reference (oracle/_ref/libnp_ref_full.so, `make -C oracle full`; veneer oracle/ref_full_harness.cpp).
Run in the build container only:

    python tests/gen_golden_reflevel.py

Inputs are seeded synthetic reads: a 6 kb contig, reads sequenced from it with substitutions, insertions, deletions and
soft clips (both strands) together with the BAM record an aligner would report, and their raw current traces; two
identity-aligned reads for the eventalign chain.  Everything the tests need is stored (inputs and the reference's outputs),
so the GPU box needs neither /root/reference nor numpy's RNG to reproduce them.
"""
import os
import sys
import zlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import load_models  # noqa: E402
from oracle.ref_full import FullRef, cigar_words  # noqa: E402
from nanopolish_amd.synth import synth_cigar_read, synth_raw, BASES  # noqa: E402

GOLD = os.path.join(HERE, "golden")
N_CIGAR_READS = 6
EVENTALIGN_READS = (40, 41)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def main():
    models = load_models()
    nuc = models["nucleotide"]
    F = FullRef()
    g = np.random.default_rng(20260924).integers(0, 4, 6000)
    contig = BASES[g].tobytes().decode()
    out = dict(contig=np.frombuffer(contig.encode(), np.uint8), n_reads=np.int64(N_CIGAR_READS))
    for i in range(N_CIGAR_READS):
        # read 4 is noisy enough to fail the calibration gate (var > MIN_CALIBRATION_VAR): no sites, events cleared
        kw = dict(span=1300)
        rd = synth_cigar_read(100 + i, g, nuc, **kw)
        if i == 4:
            rng = np.random.default_rng(7)
            rd["raw"] = np.maximum(rd["raw"] + 9.0 * rng.standard_normal(len(rd["raw"])), 8.0).astype(np.float32)
        cig = cigar_words(rd["cigar_ops"])
        fr = F.read("read%d" % i, rd["seq"], rd["raw"])
        p = "r%d_" % i
        out[p + "seq"] = np.frombuffer(rd["seq"].encode(), np.uint8)
        out[p + "bam_seq"] = np.frombuffer(rd["bam_seq"].encode(), np.uint8)
        out[p + "raw"] = rd["raw"]
        out[p + "cigar"] = cig
        out[p + "rc_pos"] = np.array([int(rd["rc"]), rd["pos"]], np.int64)
        out[p + "n_events"] = np.int64(fr.n_events)
        out[p + "scalings"] = np.array([fr.shift, fr.scale, fr.var, fr.events_per_base], np.float64)
        out[p + "events"] = fr.events()
        ms, mp = fr.event_map()
        out[p + "map_start"] = ms; out[p + "map_stop"] = mp
        ae, rc_flag, stride = fr.event_alignment_record(rd["rc"], rd["pos"], cig, rd["bam_seq"]) if fr.n_events else (np.zeros((0, 2), np.int32), int(rd["rc"]), 1)
        out[p + "aligned_events"] = ae
        out[p + "ear_rc_stride"] = np.array([rc_flag, stride], np.int64)
        res = fr.call_methylation(rd["rc"], rd["pos"], cig, rd["bam_seq"], contig)
        out[p + "site_start"] = res["start"]; out[p + "site_end"] = res["end"]; out[p + "site_n_motif"] = res["n_motif"]
        out[p + "site_ll_unmeth"] = res["ll_unmeth"]; out[p + "site_ll_meth"] = res["ll_meth"]
        out[p + "site_sequence"] = np.array(res["sequence"] if res["sequence"] else [""], dtype="S64")[:len(res["sequence"])]
        out[p + "Mm"] = np.frombuffer(res["Mm"].encode(), np.uint8)
        out[p + "Ml"] = res["Ml"]
        # align_read_to_ref on the same record (realign_read only calls it for reads that have events, eventalign.cpp:567-571)
        ea = fr.eventalign(rd["rc"], rd["pos"], cig, rd["bam_seq"], contig) if fr.n_events else \
            dict(ref_position=np.zeros(0, np.int32), event_idx=np.zeros(0, np.int32), hmm_state=np.zeros(0, np.uint8))
        out[p + "ea_ref_position"] = ea["ref_position"]; out[p + "ea_event_idx"] = ea["event_idx"]; out[p + "ea_hmm_state"] = ea["hmm_state"]
        # ... and as emit_event_alignment_tsv prints it (read index = i); zlib keeps the fixture small
        tsv = fr.eventalign_tsv(rd["rc"], rd["pos"], cig, rd["bam_seq"], contig, i) if fr.n_events else ""
        out[p + "ea_tsv_z"] = np.frombuffer(zlib.compress(tsv.encode(), 9), np.uint8)
        if i < 2:
            # the same records under --methylation gpc (GC -> GM sites, r9.4_450bps gpc model)
            gp = fr.call_methylation(rd["rc"], rd["pos"], cig, rd["bam_seq"], contig, methylation_type="gpc", modbam=False)
            out[p + "gpc_start"] = gp["start"]; out[p + "gpc_n_motif"] = gp["n_motif"]
            out[p + "gpc_ll_unmeth"] = gp["ll_unmeth"]; out[p + "gpc_ll_meth"] = gp["ll_meth"]
        print("read %d rc=%d pos=%d cigar_ops=%d events=%d sites=%d var=%.3f" % (i, rd["rc"], rd["pos"], len(cig), fr.n_events,
                                                                                 len(res["start"]), fr.var))
    # --methylation dam (GATC -> GMTC) and dcm (CCAGG / CCTGG -> CMAGG / CMTGG): a contig with those motifs planted every ~45 bases
    # (single sites, pairs 7 bases apart and one motif cut by the contig end), two reads with indels, one per strand
    rng = np.random.default_rng(424242)
    mc = rng.integers(0, 4, 1800)
    motifs = ["GATC", "CCAGG", "CCTGG", "GATC", "GATCNNNGATC", "CCAGGNNCCTGG"]
    pos_m = 30
    for j in range(36):
        m = motifs[j % len(motifs)]
        for t, ch in enumerate(m):
            if ch != "N":
                mc[pos_m + t] = "ACGT".index(ch)
        pos_m += 45 + int(rng.integers(0, 8))
    for t, ch in enumerate("GAT"):
        mc[len(mc) - 3 + t] = "ACGT".index(ch)
    mcontig = BASES[mc].tobytes().decode()
    out["m_contig"] = np.frombuffer(mcontig.encode(), np.uint8)
    for j in range(2):
        rd = synth_cigar_read(200 + j, mc, nuc, span=1700, rc=bool(j), soft_clip=(0, 6))
        cig = cigar_words(rd["cigar_ops"])
        fr = F.read("mread%d" % j, rd["seq"], rd["raw"])
        p = "m%d_" % j
        out[p + "seq"] = np.frombuffer(rd["seq"].encode(), np.uint8); out[p + "raw"] = rd["raw"]; out[p + "cigar"] = cig
        out[p + "rc_pos"] = np.array([int(rd["rc"]), rd["pos"]], np.int64)
        for alpha in ("dam", "dcm"):
            q = fr.call_methylation(rd["rc"], rd["pos"], cig, rd["bam_seq"], mcontig, methylation_type=alpha, modbam=False)
            out[p + alpha + "_start"] = q["start"]; out[p + alpha + "_end"] = q["end"]; out[p + alpha + "_n_motif"] = q["n_motif"]
            out[p + alpha + "_ll_unmeth"] = q["ll_unmeth"]; out[p + alpha + "_ll_meth"] = q["ll_meth"]
            print("motif read %d %s: %d sites, n_motif max %d" % (j, alpha, len(q["start"]), int(q["n_motif"].max()) if len(q["start"]) else 0))
    # eventalign: two identity-aligned reads (forward and reverse strand), align_read_to_ref's emitted rows
    for j, rid in enumerate(EVENTALIGN_READS):
        rd = synth_raw(rid, nuc, L=1500)
        seq = rd["seq"]
        fr = F.read("ea%d" % j, seq, rd["raw"])
        ref = revcomp(seq) if rd["rc"] else seq
        res = fr.eventalign(rd["rc"], 0, cigar_words([("M", len(seq))]), ref, ref)
        p = "ea%d_" % j
        out[p + "seq"] = np.frombuffer(seq.encode(), np.uint8)
        out[p + "raw"] = rd["raw"]
        out[p + "rc"] = np.int64(rd["rc"])
        out[p + "scalings"] = np.array([fr.shift, fr.scale, fr.var, fr.events_per_base], np.float64)
        out[p + "events"] = fr.events()
        out[p + "ref_position"] = res["ref_position"]; out[p + "event_idx"] = res["event_idx"]; out[p + "hmm_state"] = res["hmm_state"]
        out[p + "model_kmer"] = np.array(res["model_kmer"], dtype="S8")
        print("eventalign read %d rc=%d rows=%d" % (rid, rd["rc"], len(res["event_idx"])))
    # the gpc pore model as the reference loads it (fixture for the gpc parity tests)
    from oracle import RefOracle
    for alpha in ("gpc", "dam", "dcm"):
        gm = RefOracle().model(alpha)
        np.savez_compressed(os.path.join(GOLD, "models_r9.4_450bps_%s.npz" % alpha), **{f: gm[f] for f in ("level_mean", "level_stdv", "level_log_stdv")})
    path = os.path.join(GOLD, "golden_reflevel.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
