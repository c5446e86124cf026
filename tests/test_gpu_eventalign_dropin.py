"""The batched eventalign binding (nanopolish_amd/csrc/np_eventalign_dropin.cpp): realign_read's per-record work
(src/alignment/nanopolish_eventalign.cpp:539-610) for a whole batch in one device pass, against the UNMODIFIED reference on the
same records (goldens' reads: substitutions, indels, clips, both strands, QC failures):
  * the SquiggleRead the binding rebuilds equals the one SquiggleRead(sequence, Fast5Data) builds through load_from_raw -- event
    count, every event's mean / stdv / duration / start time, scalings, events_per_base, the base-to-event map;
  * the EventAlignment rows equal align_read_to_ref's;
  * the text the reference's own emit_event_alignment_tsv prints from the binding's (SquiggleRead, rows) equals, byte for byte, what
    it prints in the unmodified reference."""
import os

import numpy as np
import pytest

from oracle.ref_full import FullRef, have_full, have_batch, realign_batch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (have_full() and have_batch()), reason="reference-backed libraries not built")]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_reflevel.npz")


def _s(a):
    return bytes(a).decode()


def test_batch_realignment_equals_the_unmodified_reference():
    import torch  # noqa: F401
    g = np.load(GOLD)
    contig = _s(g["contig"])
    recs = []
    for i in range(int(g["n_reads"])):
        p = "r%d_" % i
        rc, pos = (int(v) for v in g[p + "rc_pos"])
        recs.append(dict(seq=_s(g[p + "seq"]), raw=g[p + "raw"], rc=rc, pos=pos, cigar=g[p + "cigar"], bam_seq=_s(g[p + "bam_seq"])))
    F = FullRef()
    for order in (list(range(len(recs))), list(range(len(recs)))[::-1]):
        got, status = realign_batch([recs[i] for i in order], contig)
        n_rows = 0
        for q, i in enumerate(order):
            r = recs[i]
            fr = F.read("read%d" % q, r["seq"], r["raw"])
            o = got[q]
            assert status[q] in (0, 1), "record %d took the host path" % i
            assert (status[q] == 1) == (fr.n_events == 0)
            assert o["n_events"] == fr.n_events
            s, e = fr.event_map()
            if fr.map_size:
                assert np.array_equal(o["map_start"], s) and np.array_equal(o["map_stop"], e)
                assert o["events_per_base"] == fr.events_per_base
            if fr.n_events == 0:
                assert len(o["event_idx"]) == 0
                continue
            assert (o["shift"], o["scale"], o["var"]) == (fr.shift, fr.scale, fr.var)
            assert np.array_equal(o["mean"], fr.events())
            ea = fr.eventalign(r["rc"], r["pos"], r["cigar"], r["bam_seq"], contig)
            assert np.array_equal(o["ref_position"], ea["ref_position"]) and np.array_equal(o["event_idx"], ea["event_idx"])
            assert np.array_equal(o["hmm_state"], ea["hmm_state"])
            assert o["tsv"] == fr.eventalign_tsv(r["rc"], r["pos"], r["cigar"], r["bam_seq"], contig, read_idx=q)
            n_rows += len(ea["event_idx"])
            fr.close()
        assert n_rows > 3000


def test_oversegmented_reads_long_segments(ctx, models):
    """Reads with ~3.7 events per base (every k-mer's dwell split into two sub-levels): a 100-base segment then spans 370+ events and its
    back-track takes ~470 steps over many LDS window refills -- the longest lists of walk bursts the chain kernel sees (whether they
    outgrow the LDS list, NP_EA_BCAP bursts, depends on the burst lengths; tests/test_gpu_chain_spill.py forces the spill with a build
    whose list holds 8).  Against the reference's own SquiggleRead + align_read_to_ref on the same raw signal."""
    from nanopolish_amd import api
    from nanopolish_amd.pipeline import build_host_batch_records, CallMethylationBatch
    from nanopolish_amd.synth import BASES, nucleotide_kmer_ranks
    nuc = models["nucleotide"]
    F = FullRef()
    recs, want = [], []
    for rid in range(3):
        rng = np.random.default_rng(4200 + rid)
        codes = rng.integers(0, 4, 1400)
        seq = BASES[codes].tobytes().decode()
        ranks = nucleotide_kmer_ranks(codes, 6)
        shift, scale, var = rng.uniform(-3, 3), rng.uniform(0.95, 1.05), 1.1
        sub = np.tile(np.array([-1.8, 1.8]), len(ranks))                      # two sub-levels per k-mer, 10 samples each
        rk = np.repeat(np.repeat(ranks, 2), 10)
        mu = scale * nuc["level_mean"][rk] + shift + np.repeat(sub, 10) * nuc["level_stdv"][rk]
        raw = np.maximum(mu + 0.4 * var * nuc["level_stdv"][rk] * rng.standard_normal(len(rk)), 8.0).astype(np.float32)
        rc = rid & 1
        ref = api.reverse_complement("nucleotide", seq) if rc else seq
        recs.append(dict(seq=seq, raw=raw, rc=rc, pos=0, cigar=api.cigar_words([("M", len(seq))]), contig=ref))
        fr = F.read("o%d" % rid, seq, raw)
        assert fr.n_events > 0 and fr.events_per_base > 3.3, (fr.n_events, fr.events_per_base)
        want.append(fr.eventalign(rc, 0, recs[-1]["cigar"], ref, ref))
        fr.close()
    hb = build_host_batch_records(models, recs, "")
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True)
    batch.step()
    got = batch.eventalign()
    for g, w in zip(got, want):
        assert g["status"] == 0 and len(w["event_idx"]) > 3000
        assert np.array_equal(g["ref_position"], w["ref_position"]) and np.array_equal(g["event_idx"], w["event_idx"])
        assert np.array_equal(g["hmm_state"], w["hmm_state"])
