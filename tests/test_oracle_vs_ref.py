"""Pins the portable oracle against the reference's own code compiled in place (oracle/_ref/libnp_ref.so).
Only runs where that library exists (the build container; it also travels to the GPU box as a built .so)."""
import numpy as np
import pytest

from oracle import have_ref
from cases import K, HAF_PRE, HAF_POST, methylation_jobs, eventalign_segments, synth_read

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libnp_ref.so not built (needs /root/reference)")


@pytest.fixture(scope="module")
def ref():
    from oracle import RefOracle
    return RefOracle()


def test_models_and_table_match_fixture(ref, orc, models):
    for a in ("nucleotide", "cpg"):
        m = ref.model(a)
        for f in ("level_mean", "level_stdv", "level_log_stdv"):
            assert np.array_equal(m[f], models[a][f])
    assert np.array_equal(ref.flogsum_table(), orc.flogsum_table())


def test_alphabets_random_strings(ref, orc):
    rng = np.random.default_rng(7)
    for a in ("nucleotide", "cpg", "gpc", "dam", "dcm"):
        for _ in range(300):
            n = int(rng.integers(1, 30))
            s = "".join(rng.choice(list("ACGT"), n))
            m = ref.methylate(a, s)
            assert orc.methylate(a, s) == m
            for t in (s, m):
                assert orc.reverse_complement(a, t) == ref.reverse_complement(a, t)
                assert orc.unmethylate(a, t) == ref.unmethylate(a, t)
                for i in range(max(n - 1, 0)):
                    assert orc.is_motif_match(a, t, i) == ref.is_motif_match(a, t, i)
                if n >= 6:
                    assert orc.kmer_rank(a, t[:6]) == ref.kmer_rank(a, t[:6])


@pytest.mark.parametrize("rid,L", [(100, 300), (101, 300), (102, 1000), (103, 1000), (104, 2600), (105, 2600)])
def test_read_pipeline_bit_equal(ref, orc, models, rid, L):
    mn = orc.model(models["nucleotide"]); mc = orc.model(models["cpg"])
    rd = synth_read(rid, models["nucleotide"], L=L)
    assert orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"]) == ref.estimate_scalings_mom(rd["seq"], rd["events"])
    sh, sc = ref.estimate_scalings_mom(rd["seq"], rd["events"])
    p_ref = ref.event_align(rd["events"], rd["seq"], sh, sc)
    p = orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"])
    assert len(p_ref) > 0 and np.array_equal(p, p_ref)
    epb, jobs = methylation_jobs(orc, rd, p)
    S = orc.scalings(rd["shift"], rd["scale"], rd["var"])
    for j in jobs:
        for s, r in ((j["subseq"], j["rc_subseq"]), (j["m_subseq"], j["rc_m_subseq"])):
            ranks = orc.sequence_kmer_ranks("cpg", s, r, K, j["rc"])
            for flags, bias in ((HAF_PRE | HAF_POST, 1.0), (0, 0.9)):
                a = orc.hmm_score(mc, S, rd["events"], ranks, j["e1"], j["e2"], j["stride"], epb, bias, flags)
                b = ref.hmm_score("cpg", s, r, rd["events"], j["e1"], j["e2"], j["stride"], j["rc"], rd["shift"], rd["scale"],
                                  rd["var"], epb, bias, flags)
                assert a == b
    if not rd["rc"]:
        epb2, segs = eventalign_segments(orc, rd, p)
        for sg in segs:
            ranks = orc.sequence_kmer_ranks("nucleotide", sg["seq"], None, K, 0)
            a = orc.hmm_align(mn, S, rd["events"], ranks, sg["e1"], sg["e2"], 1, epb2)
            b = ref.hmm_align("nucleotide", sg["seq"], None, rd["events"], sg["e1"], sg["e2"], 1, 0, rd["shift"], rd["scale"],
                              rd["var"], epb2)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_qc_failure_matches(ref, orc, models):
    mn = orc.model(models["nucleotide"])
    rd = synth_read(40, models["nucleotide"], L=600); other = synth_read(41, models["nucleotide"], L=600)
    sh, sc = ref.estimate_scalings_mom(other["seq"], rd["events"])
    assert len(ref.event_align(rd["events"], other["seq"], sh, sc)) == 0
    assert len(orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], other["ranks"])) == 0


def test_event_detection_bit_equal(ref, orc, models):
    """f2: the restated detector vs the reference's scrappie objects on seeded raw signal, incl. degenerate inputs."""
    from nanopolish_amd.synth import synth_raw
    from oracle.oracle_py import ED_DEFAULTS, ED_RNA
    rng = np.random.default_rng(5)
    raws = [synth_raw(r, models["nucleotide"], L=L)["raw"] for r, L in ((3, 400), (4, 2000), (5, 120))]
    raws.append((80 + 10 * rng.standard_normal(3000)).astype(np.float32))            # white noise
    raws.append(np.repeat(rng.uniform(60, 120, 60), 25).astype(np.float32))           # noiseless steps (zero variance windows)
    raws.append(synth_raw(6, models["nucleotide"], L=300)["raw"][:40])                # very short
    from test_gpu_events import quiet_stretch_raw
    raws.append(quiet_stretch_raw(models))                                            # long event-free stretch with tiny t-statistics
    for raw in raws:
        for prm in (ED_DEFAULTS, ED_RNA):
            a, b = orc.detect_events(raw, **prm), ref.detect_events(raw, **prm)
            for k in a:
                assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_vector_overload_is_the_fp32_sum_in_index_order(ref, orc, models):
    """a8: profile_hmm_score(sequence, vector<HMMInputData>) of the unmodified reference == its own single-read scores added
    up in fp32, front to back (src/hmm/nanopolish_profile_hmm.cpp:14-21) -- what the GPU drop-in test then expects."""
    from cases import vector_overload_case
    for rid in (102, 103):
        rd = synth_read(rid, models["nucleotide"], L=1000)
        sh, sc = ref.estimate_scalings_mom(rd["seq"], rd["events"])
        pairs = ref.event_align(rd["events"], rd["seq"], sh, sc)
        epb, seq, rc_seq, datas = vector_overload_case(orc, rd, pairs)
        assert len(datas) >= 3
        for flags in (0, HAF_PRE | HAF_POST):
            singles = [ref.hmm_score("cpg", seq, None, d["events"], d["e_start"], d["e_stop"], d["stride"], d["rc"], d["shift"], d["scale"],
                                     d["var"], d["events_per_base"], 1.0, flags) for d in datas]
            want = np.float32(0.0)
            for s in singles:
                want = np.float32(want + np.float32(s))
            got = ref.hmm_score_vec("cpg", seq, datas, 1.0, flags)
            assert np.isfinite(got) and np.float32(got) == want
