"""Drop-in boundary on the GPU: the reference's own objects and harness, with src/hmm/nanopolish_profile_hmm.cpp and
src/nanopolish_raw_loader.cpp replaced by nanopolish_amd/csrc/np_dropin.cpp (same C++ signatures, forwarding to the
C ABI -> HIP).  oracle/_ref/libnp_ref_dropin.so is built by `make -C oracle dropin` in the build container and travels
to the GPU box as a built artefact.  Every call below goes reference-signature -> shim -> libnp_hip.so -> MI355X and
must reproduce the vectors the unmodified reference produced (tests/golden/, tests/gen_golden.py)."""
import os

import numpy as np
import pytest

from cases import K, HAF_PRE, HAF_POST, methylation_jobs, eventalign_segments, synth_read

DROPIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libnp_ref_dropin.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(DROPIN), reason="libnp_ref_dropin.so not built")]


@pytest.fixture(scope="module")
def dropin():
    import torch  # noqa: F401  (one HIP runtime per process: torch's)
    from oracle import RefOracle
    return RefOracle(DROPIN)


def test_reference_harness_through_the_shim_reproduces_goldens(dropin, orc, models):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_reads.npz"))
    n_scores = 0
    for rid, L in zip(g["read_ids"], g["read_L"]):
        rd = synth_read(int(rid), models["nucleotide"], L=int(L))
        p = "r%d_" % rid
        assert dropin.estimate_scalings_mom(rd["seq"], rd["events"]) == tuple(g[p + "mom"])
        sh, sc = g[p + "mom"]
        pairs = dropin.event_align(rd["events"], rd["seq"], sh, sc)          # adaptive_banded_simple_event_align
        assert np.array_equal(pairs, g[p + "pairs"])
        if p + "score_meth" not in g.files:
            continue
        epb, jobs = methylation_jobs(orc, rd, pairs)
        su = [dropin.hmm_score("cpg", j["subseq"], j["rc_subseq"], rd["events"], j["e1"], j["e2"], j["stride"], j["rc"],
                               rd["shift"], rd["scale"], rd["var"], epb, 1.0, HAF_PRE | HAF_POST) for j in jobs]
        sm = [dropin.hmm_score("cpg", j["m_subseq"], j["rc_m_subseq"], rd["events"], j["e1"], j["e2"], j["stride"], j["rc"],
                               rd["shift"], rd["scale"], rd["var"], epb, 1.0, HAF_PRE | HAF_POST) for j in jobs]
        assert np.array_equal(np.array(su, np.float32), g[p + "score_unmeth"])
        assert np.array_equal(np.array(sm, np.float32), g[p + "score_meth"])
        n_scores += len(su)
        fv = []
        for j in jobs[:6]:
            for flags in (0, HAF_PRE, HAF_POST):
                for bias in (1.0, 0.9):            # hmm_indel_bias_factor is read per call by the shim
                    fv.append(dropin.hmm_score("cpg", j["subseq"], j["rc_subseq"], rd["events"], j["e1"], j["e2"], j["stride"],
                                               j["rc"], rd["shift"], rd["scale"], rd["var"], epb, bias, flags))
        assert np.array_equal(np.array(fv, np.float32), g[p + "score_flagvar"])
        if not rd["rc"]:
            epb2, segs = eventalign_segments(orc, rd, pairs)
            for si, sg in enumerate(segs[:6]):
                q = p + "seg%d_" % si
                ev, km, lf, st = dropin.hmm_align("nucleotide", sg["seq"], None, rd["events"], sg["e1"], sg["e2"], 1, 0,
                                                  rd["shift"], rd["scale"], rd["var"], epb2)          # profile_hmm_align
                assert np.array_equal(ev, g[q + "event_idx"]) and np.array_equal(km, g[q + "kmer_idx"])
                assert np.array_equal(lf, g[q + "l_fm"]) and np.array_equal(st, g[q + "state"])
            ss = [dropin.hmm_score_set([sg["seq"][:30], orc.methylate("cpg", sg["seq"][:30])], ["nucleotide", "cpg"], rd["events"],
                                       sg["e1"], sg["e1"] + 40, 1, 0, rd["shift"], rd["scale"], rd["var"], epb2, 0.9, 0)
                  for sg in segs[:4]]                                                                   # profile_hmm_score_set
            assert np.array_equal(np.array(ss, np.float32), g[p + "score_set"])
    assert n_scores > 200
