"""Direct-RNA reads on the device path (Missing 3): load_from_raw's RNA branch -- kit r9.4_70bps, alphabet u_to_t_rna, k = 5,
the RNA detector parameters, events reversed after the MoM scalings (src/nanopolish_squiggle_read.cpp:206-213,260-263) -- followed by
align_read_to_ref, against the UNMODIFIED reference compiled in place (oracle/_ref/libnp_ref_full.so) on synthetic RNA-like reads (the
strand passes the pore 3' -> 5', ~42 samples per base): through the Python pipeline and through the reference-side binding
np_realign_reads_batch, alone and mixed with DNA reads in one batch."""
import os

import numpy as np
import pytest

from oracle.ref_full import FullRef, have_full, have_batch, realign_batch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_full(), reason="oracle/_ref/libnp_ref_full.so not built")]
HERE = os.path.dirname(os.path.abspath(__file__))


def rna_model():
    z = np.load(os.path.join(HERE, "golden", "models_r9.4_70bps_rna.npz"))
    return dict(k=5, level_mean=z["u_to_t_rna_level_mean"], level_stdv=z["u_to_t_rna_level_stdv"], level_log_stdv=z["u_to_t_rna_level_log_stdv"])


def rna_records(model, ids, L):
    """identity-aligned RNA reads: forward strand (direct RNA is sequenced as the transcript), each against its own sequence"""
    from nanopolish_amd import api
    from nanopolish_amd.synth import synth_raw_rna
    recs = []
    for rid in ids:
        rd = synth_raw_rna(rid, model, L=L)
        recs.append(dict(seq=rd["seq"], raw=rd["raw"], rc=0, pos=0, cigar=api.cigar_words([("M", len(rd["seq"]))]), contig=rd["seq"], bam_seq=rd["seq"]))
    return recs


def test_rna_reads_through_the_pipeline_equal_the_reference(ctx):
    from nanopolish_amd.pipeline import build_host_batch_records, CallMethylationBatch
    m = rna_model()
    if "u_to_t_rna" not in ctx.models:
        ctx.register_model(m, "u_to_t_rna")
    recs = rna_records(m, range(700, 712), 700) + rna_records(m, [720, 721], 2400)
    F = FullRef()
    want = []
    for q, r in enumerate(recs):
        fr = F.read("rna%d" % q, r["seq"], r["raw"], sample_rate=3012.0, rna=True)
        assert fr.n_events > 0, "the synthetic RNA read must survive the reference's QC gates"
        want.append((fr.n_events, fr.events(), (fr.shift, fr.scale, fr.var), fr.events_per_base,
                     fr.eventalign(0, 0, r["cigar"], r["bam_seq"], r["contig"])))
        fr.close()
    hb = build_host_batch_records({"nucleotide": m}, recs, "", k=5, with_jobs=False)
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True, workload="eventalign", rna=True, base_model="u_to_t_rna")
    batch.step()
    got = batch.eventalign_results()
    n_rows = 0
    for i, (w, g) in enumerate(zip(want, got)):
        ne, st, ln, mean, sd = batch.detected(i)
        assert ne == w[0] and np.array_equal(mean, w[1]), "read %d: events (reversed, 5' -> 3')" % i
        assert g["status"] == 0
        assert np.array_equal(g["ref_position"], w[4]["ref_position"]) and np.array_equal(g["event_idx"], w[4]["event_idx"])
        assert np.array_equal(g["hmm_state"], w[4]["hmm_state"])
        n_rows += len(g["event_idx"])
    assert n_rows > 5000


@pytest.mark.skipif(not have_batch(), reason="oracle/_ref/libnp_ref_full_batch.so not built")
def test_rna_and_dna_records_in_one_batch_through_the_binding():
    """np_realign_reads_batch with direct-RNA and DNA records mixed: one device pass per nucleotide type; every rebuilt SquiggleRead
    (events in 5' -> 3' order with the reference's start times, scalings, event map) and every EventAlignment row, and the TSV text the
    reference's own writer prints from them, equal the unmodified reference's."""
    import torch  # noqa: F401
    from nanopolish_amd import api
    from nanopolish_amd.synth import synth_raw
    from oracle import load_models
    m = rna_model()
    rna = rna_records(m, range(730, 736), 800)
    nuc = load_models()["nucleotide"]
    dna = []
    for rid in (740, 742, 744):
        rd = synth_raw(rid, nuc, L=900)
        ref = api.reverse_complement("nucleotide", rd["seq"]) if rd["rc"] else rd["seq"]
        dna.append(dict(seq=rd["seq"], raw=rd["raw"], rc=int(rd["rc"]), pos=0, cigar=api.cigar_words([("M", len(ref))]), contig=ref, bam_seq=ref))
    # one contig: the records' references back to back
    recs, contig, is_rna, pos = [], [], [], 0
    for r, flag in [(rna[0], 1), (dna[0], 0), (rna[1], 1), (rna[2], 1), (dna[1], 0), (rna[3], 1), (dna[2], 0), (rna[4], 1), (rna[5], 1)]:
        recs.append(dict(r, pos=pos)); contig.append(r["contig"]); is_rna.append(flag); pos += len(r["contig"])
    contig = "".join(contig)
    got, status = realign_batch(recs, contig, sample_rate=3012.0, rna=[i for i, f in enumerate(is_rna) if f])
    F = FullRef()
    n_rows = 0
    for q, (r, flag) in enumerate(zip(recs, is_rna)):
        fr = F.read("read%d" % q, r["seq"], r["raw"], sample_rate=3012.0, rna=bool(flag))
        o = got[q]
        assert status[q] == 0 and fr.n_events > 0
        assert o["n_events"] == fr.n_events and (o["shift"], o["scale"], o["var"]) == (fr.shift, fr.scale, fr.var)
        assert np.array_equal(o["mean"], fr.events())
        s, e = fr.event_map()
        assert np.array_equal(o["map_start"], s) and np.array_equal(o["map_stop"], e) and o["events_per_base"] == fr.events_per_base
        ea = fr.eventalign(r["rc"], r["pos"], r["cigar"], r["bam_seq"], contig)
        assert np.array_equal(o["ref_position"], ea["ref_position"]) and np.array_equal(o["event_idx"], ea["event_idx"])
        assert np.array_equal(o["hmm_state"], ea["hmm_state"])
        assert o["tsv"] == fr.eventalign_tsv(r["rc"], r["pos"], r["cigar"], r["bam_seq"], contig, read_idx=q)
        n_rows += len(ea["event_idx"])
        fr.close()
    assert n_rows > 3000
