"""The batched binding of the variant callers' scoring loops (nanopolish_amd/csrc/np_variants_dropin.cpp) against
the UNMODIFIED reference: score_variant_thresholded (src/common/nanopolish_variant.cpp:765-799) over the screening loop of
generate_candidate_single_base_edits (src/nanopolish_call_variants.cpp:288-352) and score_variant_group (variant.cpp:182-262).
Three configurations on the same reads (SquiggleReads the reference itself loads from raw signal, aligned to a contig with
substitutions, indels and clips, both strands):
    reference            oracle/_ref/libnp_ref_full.so        the reference's own functions on the host, one OpenMP thread
    per-call shim        oracle/_ref/libnp_ref_full_batch.so  the same reference functions, every profile_hmm_score through np_dropin.cpp
    batched binding      oracle/_ref/libnp_ref_full_batch.so  np_score_variants_thresholded / np_score_variant_group: one device batch
All three must agree bit for bit (qualities are sums of float scores in read order, accumulated in double)."""
import os

import numpy as np
import pytest

from oracle.ref_full import FullRef, have_full, have_batch, cigar_words
from nanopolish_amd.synth import synth_cigar_read, BASES

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (have_full() and have_batch()), reason="reference-backed libraries not built")]


@pytest.fixture(scope="module")
def setup(models):
    import torch  # noqa: F401  (one HIP runtime per process: torch's)
    rng = np.random.default_rng(77)
    contig_codes = rng.integers(0, 4, 1600)
    contig = BASES[contig_codes].tobytes().decode()
    ref, dev = FullRef(), FullRef(batch=True)
    rr, dr, recs = [], [], []
    for rid in range(14):
        rd = synth_cigar_read(500 + rid, contig_codes, models["nucleotide"], span=1300)
        a, b = ref.read("r%d" % rid, rd["seq"], rd["raw"]), dev.read("r%d" % rid, rd["seq"], rd["raw"])
        assert a.n_events == b.n_events and a.shift == b.shift and a.scale == b.scale and a.var == b.var
        if a.n_events == 0:
            continue
        rr.append(a); dr.append(b)
        recs.append(dict(rc=rd["rc"], pos=rd["pos"], cigar=cigar_words(rd["cigar_ops"]), bam_seq=rd["bam_seq"]))
    assert len(rr) >= 10
    return ref, dev, rr, dr, recs, contig


@pytest.mark.parametrize("meth,threshold", [("", 1000000), ("cpg", 1000000), ("cpg", 30), ("cpg,dam", 60)])
def test_screening_qualities_equal_the_reference(setup, meth, threshold):
    ref, dev, rr, dr, recs, contig = setup
    positions = [650, 651, 652, 653, 700, 701, 760, 811, 900, 901]
    want, w_win, sets = ref.score_variants(0, rr, recs, contig, positions, methylation_types=meth, score_threshold=threshold)
    shim, s_win, _ = dev.score_variants(0, dr, recs, contig, positions[:3], methylation_types=meth, score_threshold=threshold)
    got, g_win, gsets = dev.score_variants(1, dr, recs, contig, positions, methylation_types=meth, score_threshold=threshold)
    assert len(want) > 60 and sets == gsets and sets > 500
    assert np.array_equal(w_win, g_win) and np.array_equal(got, want)
    assert np.array_equal(shim, want[:len(shim)])
    if threshold < 1000:
        assert (np.abs(want) >= threshold).any(), "the early-out never triggered: the case does not test it"


def test_variant_group_scores_equal_the_reference(setup):
    ref, dev, rr, dr, recs, contig = setup
    for positions, meth in (([700, 704, 709], "cpg"), ([820, 823], ""), ([640, 645, 650, 655], "cpg")):
        want = ref.score_variant_group(0, rr, recs, contig, positions, methylation_types=meth)
        got = dev.score_variant_group(1, dr, recs, contig, positions, methylation_types=meth)
        assert want.shape == got.shape and want.shape[0] == 2 ** len(positions) and want.shape[1] >= 8
        assert np.array_equal(got, want)
