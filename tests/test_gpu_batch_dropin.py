"""The batched reference-side binding on the GPU: nanopolish_amd/csrc/np_batch_dropin.cpp replaces the body of
call-methylation's per-record loop -- calculate_methylation_for_read_from_bam (src/nanopolish_call_methylation.cpp:163-177)
under BamProcessor's batch boundary (src/common/nanopolish_bam_processor.cpp:90-119) -- by one device pass over the whole batch.
oracle/_ref/libnp_ref_full_batch.so (`make -C oracle batch`) is the reference's read-level build with that file (and the per-call
shim np_dropin.cpp) linked in place of the hot-path translation units; it travels to the GPU box as a built artefact.
The ScoredSite maps it fills must equal, field for field, what the UNMODIFIED reference produced for the same records
(tests/golden/golden_reflevel.npz, tests/gen_golden_reflevel.py)."""
import os

import numpy as np
import pytest

from oracle.ref_full import have_batch, call_methylation_batch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_batch(), reason="libnp_ref_full_batch.so not built")]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_reflevel.npz")


def _s(a):
    return bytes(a).decode()


def test_scored_site_maps_of_one_batch_equal_the_unmodified_reference():
    import torch  # noqa: F401  (one HIP runtime per process: torch's)
    g = np.load(GOLD)
    recs = []
    for i in range(int(g["n_reads"])):
        p = "r%d_" % i
        rc, pos = (int(v) for v in g[p + "rc_pos"])
        recs.append(dict(seq=_s(g[p + "seq"]), raw=g[p + "raw"], rc=rc, pos=pos, cigar=g[p + "cigar"], bam_seq=_s(g[p + "bam_seq"])))
    # the same batch twice (steady state of the cached models / scratch) and in reversed record order
    for order in (list(range(len(recs))), list(range(len(recs)))[::-1]):
        out, status = call_methylation_batch([recs[i] for i in order], _s(g["contig"]))
        n_sites = 0
        for q, i in enumerate(order):
            p = "r%d_" % i
            got = out[q]
            assert status[q] in (0, 1), "record %d took the host path" % i
            assert (status[q] == 1) == (int(g[p + "n_events"]) == 0)          # reads the reference cleared: empty maps
            assert np.array_equal(got["start"], g[p + "site_start"]) and np.array_equal(got["end"], g[p + "site_end"])
            assert np.array_equal(got["n_motif"], g[p + "site_n_motif"])
            assert np.array_equal(got["ll_unmeth"], g[p + "site_ll_unmeth"]) and np.array_equal(got["ll_meth"], g[p + "site_ll_meth"])
            assert got["sequence"] == [x.decode() for x in g[p + "site_sequence"]]
            n_sites += len(got["start"])
        assert n_sites > 150


def test_gpc_batch_equals_the_unmodified_reference():
    import torch  # noqa: F401
    g = np.load(GOLD)
    recs = []
    for i in range(2):
        p = "r%d_" % i
        rc, pos = (int(v) for v in g[p + "rc_pos"])
        recs.append(dict(seq=_s(g[p + "seq"]), raw=g[p + "raw"], rc=rc, pos=pos, cigar=g[p + "cigar"], bam_seq=_s(g[p + "bam_seq"])))
    out, status = call_methylation_batch(recs, _s(g["contig"]), methylation_type="gpc")
    for i in range(2):
        p = "r%d_" % i
        assert np.array_equal(out[i]["start"], g[p + "gpc_start"]) and np.array_equal(out[i]["n_motif"], g[p + "gpc_n_motif"])
        assert np.array_equal(out[i]["ll_unmeth"], g[p + "gpc_ll_unmeth"]) and np.array_equal(out[i]["ll_meth"], g[p + "gpc_ll_meth"])


def _golden_records(g):
    recs = []
    for i in range(int(g["n_reads"])):
        p = "r%d_" % i
        rc, pos = (int(v) for v in g[p + "rc_pos"])
        recs.append(dict(seq=_s(g[p + "seq"]), raw=g[p + "raw"], rc=rc, pos=pos, cigar=g[p + "cigar"], bam_seq=_s(g[p + "bam_seq"])))
    return recs


def _assert_golden(g, i, got):
    p = "r%d_" % i
    assert np.array_equal(got["start"], g[p + "site_start"]) and np.array_equal(got["end"], g[p + "site_end"])
    assert np.array_equal(got["n_motif"], g[p + "site_n_motif"])
    assert np.array_equal(got["ll_unmeth"], g[p + "site_ll_unmeth"]) and np.array_equal(got["ll_meth"], g[p + "site_ll_meth"])


@pytest.mark.parametrize("coalesce", ["8192", "1", "5"])
def test_pipelined_batches_equal_the_unmodified_reference(coalesce, monkeypatch):
    """NpBatchPipeline, batches in flight (persistent buffers, one upload and one read-back per device pass, three streams): the
    records in batches of 3 and of 1 -- every slot reused several times, batches of different sizes back to back.  Round 5: the packer
    merges the batches that are already waiting into one device pass of up to NP_BATCH_COALESCE records (default 8 192: here everything
    that waits goes into one pass; 1: every batch is a pass of its own, the round-4 behaviour; 5: passes of one to five records, groups
    that end in the middle of the queue) -- results per batch, in submission order, whatever was merged."""
    from oracle.ref_full import call_methylation_pipeline
    import torch  # noqa: F401
    monkeypatch.setenv("NP_BATCH_COALESCE", coalesce)
    g = np.load(GOLD)
    recs = _golden_records(g)
    for bs in (3, 1, len(recs), 2):
        out, status = call_methylation_pipeline(recs, _s(g["contig"]), bs)
        for i in range(len(recs)):
            assert status[i] in (0, 1)
            assert (status[i] == 1) == (int(g["r%d_n_events" % i]) == 0)
            _assert_golden(g, i, out[i])


@pytest.mark.parametrize("piece", ["1", "2"])
def test_large_batches_travel_in_pieces(piece, monkeypatch):
    """Round 6: a submitted batch of more than 2 x NP_BATCH_PIECE records (default 512) is cut into pieces that go through the pipeline like
    small batches (slots, device passes, map building, pass buffers freed piece by piece) and collect() hands the BATCH back when its last
    piece is done.  With pieces of 1 and 2 records the golden records, in batches of all of them, of 5, of 3 and of 1, come back batch by batch,
    in order, statuses at the caller's indices, every map equal to the unmodified reference's."""
    from oracle.ref_full import call_methylation_pipeline
    import torch  # noqa: F401
    monkeypatch.setenv("NP_BATCH_PIECE", piece)
    g = np.load(GOLD)
    recs = _golden_records(g)
    for bs in (len(recs), 5, 3, 1):
        out, status = call_methylation_pipeline(recs, _s(g["contig"]), bs)
        for i in range(len(recs)):
            assert status[i] in (0, 1)
            assert (status[i] == 1) == (int(g["r%d_n_events" % i]) == 0)
            _assert_golden(g, i, out[i])
    out, status = call_methylation_pipeline(recs, _s(g["contig"]), len(recs), rna=[1])          # a host-path record inside a piece
    for i in range(len(recs)):
        if i == 1:
            assert status[i] == 2 and len(out[i]["start"]) == 0
        else:
            _assert_golden(g, i, out[i])


def test_two_contexts_on_one_device_deal_batches_round_robin():
    """The multi-GPU form of NpBatchPipeline (devices = {0, 0}: two library contexts of the pipeline's own, here on one GPU; batches dealt
    round-robin, three slots per context, results in submission order): batches of 1, 2 and 3 records, i.e. up to 12 batches through 6
    slots, every result equal to the unmodified reference's; then three contexts; then the process-wide context again."""
    from oracle.ref_full import call_methylation_pipeline
    import torch  # noqa: F401
    g = np.load(GOLD)
    recs = _golden_records(g)
    for contexts, bs in ((2, 1), (2, 2), (2, 3), (3, 1), (0, 2)):
        out, status = call_methylation_pipeline(recs, _s(g["contig"]), bs, contexts=contexts)
        for i in range(len(recs)):
            assert status[i] in (0, 1)
            assert (status[i] == 1) == (int(g["r%d_n_events" % i]) == 0)
            _assert_golden(g, i, out[i])


def test_event_capacity_overflow_and_rna_reads_take_the_host_path():
    """A read whose detected events exceed the device capacity (driven here by shrinking the capacity estimate) and a read flagged
    RNA come back NP_BATCH_HOST_PATH with NO sites; the other records of the same batches are unaffected, and so is the next run."""
    from oracle.ref_full import call_methylation_pipeline
    import torch  # noqa: F401
    g = np.load(GOLD)
    recs = _golden_records(g)
    out, status = call_methylation_pipeline(recs, _s(g["contig"]), 3, event_cap_divisor=64)
    assert all(int(s) == 2 for s in status) and all(len(o["start"]) == 0 for o in out)
    out, status = call_methylation_pipeline(recs, _s(g["contig"]), 3, rna=[1])
    for i in range(len(recs)):
        if i == 1:
            assert status[i] == 2 and len(out[i]["start"]) == 0
        else:
            assert status[i] in (0, 1)
            _assert_golden(g, i, out[i])


def test_adc_input_gives_the_results_of_the_converted_samples():
    """Records handed over as int16 ADC counts + the channel's conversion (half the bytes to pack and upload; converted on the device
    with the loader's fp32 expression, src/io/nanopolish_fast5_loader.cpp:96-103) against the same records handed over as the pA
    values those counts convert to."""
    from oracle.ref_full import call_methylation_pipeline
    from nanopolish_amd.synth import adc_quantise, ADC_OFFSET, ADC_UNIT
    import torch  # noqa: F401
    g = np.load(GOLD)
    recs = _golden_records(g)
    for r in recs:
        r["adc"], r["raw"] = adc_quantise(r["raw"])
    want, s0 = call_methylation_pipeline(recs, _s(g["contig"]), 4)
    got, s1 = call_methylation_pipeline(recs, _s(g["contig"]), 4, adc=(float(ADC_OFFSET), float(ADC_UNIT)))
    assert np.array_equal(s0, s1)
    n = 0
    for a, b in zip(want, got):
        for k in ("start", "end", "n_motif", "ll_unmeth", "ll_meth"):
            assert np.array_equal(a[k], b[k])
        n += len(a["start"])
    assert n > 150
