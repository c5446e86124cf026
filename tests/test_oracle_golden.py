"""The CPU oracle (oracle/np_oracle.c) against the committed vectors the REFERENCE ITSELF produced
(tests/gen_golden.py -> tests/golden/*.npz).  Runs anywhere (no GPU, no /root/reference)."""
import os
import numpy as np
import pytest

from cases import K, HAF_PRE, HAF_POST, methylation_jobs, eventalign_segments, synth_read

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tables():
    return np.load(os.path.join(GOLD, "golden_tables.npz"))


@pytest.fixture(scope="module")
def greads():
    return np.load(os.path.join(GOLD, "golden_reads.npz"))


def test_flogsum_table_and_add_logs(orc, tables):
    assert np.array_equal(orc.flogsum_table(), tables["flogsum_table"])
    got = np.array([orc.flogsum(float(a), float(b)) for a, b in zip(tables["addlogs_a"], tables["addlogs_b"])], np.float32)
    assert np.array_equal(got, tables["addlogs_out"], equal_nan=True)


def test_alphabet_kats(orc, tables):
    for a, s, rc, me, un, mo, r6 in zip(tables["kat_alphabet"], tables["kat_in"], tables["kat_rc"], tables["kat_meth"],
                                        tables["kat_unmeth"], tables["kat_motif"], tables["kat_rank6"]):
        a, s = str(a), str(s)
        assert orc.reverse_complement(a, s) == str(rc)
        base = s.replace("M", "A") if a == "dam" else s.replace("M", "C")
        assert orc.methylate(a, base) == str(me)
        assert orc.unmethylate(a, s) == str(un)
        assert "".join("1" if orc.is_motif_match(a, s, i) else "0" for i in range(max(len(s) - 1, 0))) == str(mo)
        if r6 >= 0:
            assert orc.kmer_rank(a, s[:6]) == r6
    # the reference's own unit-test vector, src/test/nanopolish_test.cpp:239-265
    assert orc.kmer_rank("nucleotide", "GATGA") == 568


def test_emission_kats(orc, models, tables):
    mn = orc.model(models["nucleotide"]); mc = orc.model(models["cpg"])
    sh, sc, dr, var = tables["emis_scal"]
    S = orc.scalings(sh, sc, var, dr)
    got = [orc.log_probability_match_r9(mn, S, int(r), float(x)) for r, x in zip(tables["emis_rank"], tables["emis_level"])]
    assert np.array_equal(np.array(got, np.float32), tables["emis_lp"])
    S2 = orc.scalings(-3.5, 0.97, 1.11)
    got = [orc.log_probability_match_r9(mc, S2, int(r), float(x)) for r, x in zip(tables["emis_cpg_rank"], tables["emis_level"])]
    assert np.array_equal(np.array(got, np.float32), tables["emis_cpg_lp"])


def test_reads_align_score_viterbi(orc, models, greads):
    g = greads
    mn = orc.model(models["nucleotide"]); mc = orc.model(models["cpg"])
    n_scores = n_states = 0
    for rid, L in zip(g["read_ids"], g["read_L"]):
        rd = synth_read(int(rid), models["nucleotide"], L=int(L))
        p = "r%d_" % rid
        sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        assert (sh, sc) == tuple(g[p + "mom"])
        pairs = orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"])
        assert np.array_equal(pairs, g[p + "pairs"])
        if p + "score_meth" not in g.files:
            continue
        epb, jobs = methylation_jobs(orc, rd, pairs)
        assert epb == float(g[p + "epb"])
        assert [j["first"] for j in jobs] == list(g[p + "job_first"])
        assert [j["e1"] for j in jobs] == list(g[p + "job_e1"]) and [j["e2"] for j in jobs] == list(g[p + "job_e2"])
        S = orc.scalings(rd["shift"], rd["scale"], rd["var"])
        su, sm, fv = [], [], []
        for ji, j in enumerate(jobs):
            ru = orc.sequence_kmer_ranks("cpg", j["subseq"], j["rc_subseq"], K, j["rc"])
            rm = orc.sequence_kmer_ranks("cpg", j["m_subseq"], j["rc_m_subseq"], K, j["rc"])
            su.append(orc.hmm_score(mc, S, rd["events"], ru, j["e1"], j["e2"], j["stride"], epb, 1.0, HAF_PRE | HAF_POST))
            sm.append(orc.hmm_score(mc, S, rd["events"], rm, j["e1"], j["e2"], j["stride"], epb, 1.0, HAF_PRE | HAF_POST))
            if ji < 6:
                for flags in (0, HAF_PRE, HAF_POST):
                    for bias in (1.0, 0.9):
                        fv.append(orc.hmm_score(mc, S, rd["events"], ru, j["e1"], j["e2"], j["stride"], epb, bias, flags))
        assert np.array_equal(np.array(su, np.float32), g[p + "score_unmeth"])
        assert np.array_equal(np.array(sm, np.float32), g[p + "score_meth"])
        assert np.array_equal(np.array(fv, np.float32), g[p + "score_flagvar"])
        n_scores += len(su)
        if not rd["rc"]:
            epb2, segs = eventalign_segments(orc, rd, pairs)
            for si, sg in enumerate(segs[:6]):
                q = p + "seg%d_" % si
                ranks = orc.sequence_kmer_ranks("nucleotide", sg["seq"], None, K, 0)
                ev, km, lf, st = orc.hmm_align(mn, S, rd["events"], ranks, sg["e1"], sg["e2"], 1, epb2)
                assert np.array_equal(ev, g[q + "event_idx"]) and np.array_equal(km, g[q + "kmer_idx"])
                assert np.array_equal(lf, g[q + "l_fm"]) and np.array_equal(st, g[q + "state"])
                n_states += len(ev)
            ss = []
            for sg in segs[:4]:
                w = sg["seq"][:30]
                s0 = orc.hmm_score(mn, S, rd["events"], orc.sequence_kmer_ranks("nucleotide", w, None, K, 0),
                                   sg["e1"], sg["e1"] + 40, 1, epb2, 0.9, 0)
                s1 = orc.hmm_score(mc, S, rd["events"], orc.sequence_kmer_ranks("cpg", orc.methylate("cpg", w), None, K, 0),
                                   sg["e1"], sg["e1"] + 40, 1, epb2, 0.9, 0)
                ss.append(orc.combine_score_set([s0, s1]))
            assert np.array_equal(np.array(ss, np.float32), g[p + "score_set"])
    assert n_scores > 200 and n_states > 1000


def test_event_detection_matches_reference_goldens(orc, models):
    """f2: npo_detect_events vs the events the reference's own scrappie objects produced (tests/gen_golden.py)."""
    import os
    from nanopolish_amd.synth import synth_raw
    from oracle.oracle_py import ED_DEFAULTS, ED_RNA
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_events.npz"))
    for rid, L in zip(g["read_ids"], g["read_L"]):
        rd = synth_raw(int(rid), models["nucleotide"], L=int(L))
        assert np.frombuffer(rd["raw"].tobytes(), np.uint32).sum(dtype=np.uint64) == g["r%d_raw_crc" % rid][0], "synthetic raw changed"
        for tag, prm in (("dna", ED_DEFAULTS), ("rna", ED_RNA)):
            ev = orc.detect_events(rd["raw"], **prm)
            for k in ("start", "length", "mean", "stdv"):
                assert np.array_equal(ev[k], g["r%d_%s_%s" % (rid, tag, k)]), (rid, tag, k)
        assert len(ev["mean"]) > 50
