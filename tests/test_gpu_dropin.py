"""Drop-in boundary on the GPU: the reference's own objects and harness, with src/hmm/nanopolish_profile_hmm.cpp and
src/nanopolish_raw_loader.cpp replaced by nanopolish_amd/csrc/np_dropin.cpp (same C++ signatures, forwarding to the
C ABI -> HIP).  oracle/_ref/libnp_ref_dropin.so is built by `make -C oracle dropin` in the build container and travels
to the GPU box as a built artefact.  Every call below goes reference-signature -> shim -> libnp_hip.so -> MI355X and
must reproduce the vectors the unmodified reference produced (tests/golden/, tests/gen_golden.py)."""
import os

import numpy as np
import pytest

from cases import K, HAF_PRE, HAF_POST, methylation_jobs, eventalign_segments, synth_read

DROPIN = os.environ.get("NP_REF_DROPIN_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libnp_ref_dropin.so")
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


def test_vector_overload_through_the_shim(dropin, orc, models):
    """a8: profile_hmm_score(sequence, std::vector<HMMInputData>, flags) of np_dropin.cpp (one device batch over the reads, fp32
    sum in index order) against the single-call goldens semantics pinned on the unmodified reference in
    tests/test_oracle_vs_ref.py::test_vector_overload_is_the_fp32_sum_in_index_order."""
    from cases import vector_overload_case
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_reads.npz"))
    n = 0
    for rid, L in zip(g["read_ids"], g["read_L"]):
        p = "r%d_" % rid
        if p + "score_meth" not in g.files:
            continue
        rd = synth_read(int(rid), models["nucleotide"], L=int(L))
        epb, seq, rc_seq, datas = vector_overload_case(orc, rd, g[p + "pairs"])
        if len(datas) < 3:
            continue
        mc = orc.model(models["cpg"])
        for flags in (0, HAF_PRE | HAF_POST):
            want = np.float32(0.0)
            for d in datas:
                ranks = orc.sequence_kmer_ranks("cpg", seq, rc_seq, K, d["rc"])
                s = orc.hmm_score(mc, orc.scalings(d["shift"], d["scale"], d["var"]), d["events"], ranks, d["e_start"], d["e_stop"], d["stride"],
                                  d["events_per_base"], 1.0, flags)
                want = np.float32(want + np.float32(s))
            got = dropin.hmm_score_vec("cpg", seq, datas, 1.0, flags)
            assert np.float32(got) == want
            n += 1
    assert n >= 4


def test_model_overwritten_in_place_is_refreshed_on_the_device(dropin, orc, models):
    """The shim caches device models by PoreModel address; the reference overwrites registered models in place
    (pore_model_set.cpp:70, methyltrain).  After such an overwrite the next call must score against the NEW parameters."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_reads.npz"))
    rd = synth_read(2, models["nucleotide"], L=900)
    pairs = g["r2_pairs"]
    epb, jobs = methylation_jobs(orc, rd, pairs)
    j = jobs[0]

    def device_score():
        return dropin.hmm_score("cpg", j["subseq"], j["rc_subseq"], rd["events"], j["e1"], j["e2"], j["stride"], j["rc"], rd["shift"],
                                rd["scale"], rd["var"], epb, 1.0, HAF_PRE | HAF_POST)

    def oracle_score():
        mc = orc.model(dropin.model("cpg"))          # the parameters the reference-side PoreModel holds right now
        ranks = orc.sequence_kmer_ranks("cpg", j["subseq"], j["rc_subseq"], K, j["rc"])
        return orc.hmm_score(mc, orc.scalings(rd["shift"], rd["scale"], rd["var"]), rd["events"], ranks, j["e1"], j["e2"], j["stride"], epb,
                             1.0, HAF_PRE | HAF_POST)

    s0 = device_score()
    assert s0 == oracle_score()
    dropin.shift_model("cpg", 1.5)
    try:
        s1 = device_score()
        assert s1 == oracle_score() and s1 != s0
    finally:
        dropin.shift_model("cpg", -1.5)
    assert device_score() == oracle_score()


def test_concurrent_callers_are_combined_and_score_the_same(dropin, orc, models):
    """profile_hmm_score called per work item from an OpenMP loop over reads (the reference's calling pattern, scorereads.cpp:388,
    bam_processor.cpp:99) through the shim: concurrent callers share device rounds (flat combining, np_dropin.cpp), and every score
    equals the single-thread run's and the oracle's."""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from bench_percall_dropin import jobs_of
    J = jobs_of(orc, models, range(300, 316), 900)
    args = ("cpg", J["events"], J["event_off"], J["shift"], J["scale"], J["var"], J["epb"], J["job_off"], J["seqs"], J["rc_seqs"],
            J["e_start"], J["e_stop"], J["stride"], J["rc"], 3)
    one = dropin.score_many_reads(*args, 1)
    r0, c0 = C.c_long(0), C.c_long(0)
    dropin.L.np_dropin_combiner_stats(C.byref(r0), C.byref(c0))
    many = dropin.score_many_reads(*args, 8)
    r1, c1 = C.c_long(0), C.c_long(0)
    dropin.L.np_dropin_combiner_stats(C.byref(r1), C.byref(c1))
    assert len(one) > 500 and np.array_equal(one, many)
    assert c1.value - c0.value == len(one) and r1.value - r0.value < c1.value - c0.value        # fewer device rounds than calls
    mc = orc.model(models["cpg"])
    from cases import synth_read
    rd = synth_read(300, models["nucleotide"], L=900)
    S = orc.scalings(rd["shift"], rd["scale"], rd["var"])
    for j in range(J["job_off"][0], min(J["job_off"][1], 12)):
        ranks = orc.sequence_kmer_ranks("cpg", J["seqs"][j], J["rc_seqs"][j], K, J["rc"][j])
        assert one[j] == orc.hmm_score(mc, S, rd["events"], ranks, J["e_start"][j], J["e_stop"][j], J["stride"][j], J["epb"][0], 1.0, 3)
    dropin.L.np_dropin_error_count.restype = C.c_long
    assert dropin.L.np_dropin_error_count() == 0


_ERR_SCRIPT = r'''
import os, sys
sys.path.insert(0, os.environ["NP_REPO"]); sys.path.insert(0, os.path.join(os.environ["NP_REPO"], "tests"))
import numpy as np
import torch  # noqa: F401
from oracle import RefOracle, load_models
from cases import synth_read
d = RefOracle(os.environ["NP_REF_DROPIN_LIB"])
rd = synth_read(5, load_models()["nucleotide"], L=1600)
ok = d.hmm_score("nucleotide", rd["seq"][:40], None, rd["events"], 10, 70, 1, 0, rd["shift"], rd["scale"], rd["var"], 1.5, 1.0, 0)
print("OK %r" % (ok,), flush=True)
# 1 200 bases = 1 195 k-mers: more than NP_MAX_KMERS, the library refuses the work item (NP_ERR_UNSUPPORTED)
bad = d.hmm_score("nucleotide", rd["seq"][:1200], None, rd["events"], 10, 1700, 1, 0, rd["shift"], rd["scale"], rd["var"], 1.5, 1.0, 0)
print("BAD %r" % (bad,), flush=True)
'''


def test_a_failed_device_call_cannot_end_in_exit_status_0(tmp_path):
    """the reference's callers link UNCHANGED, so none of them reads np_dropin_error_count().  A work item the library
    refuses (here: more k-mers than NP_MAX_KMERS) must therefore end the process with a non-zero status -- at once by default (the
    reference's own convention for unrecoverable conditions, raw_loader.cpp:124-131), or, for a caller adapted to check the count
    (NP_DROPIN_ERRORS_IN_BAND=1), as -inf in band AND a non-zero status when the process exits."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "err.py"
    script.write_text(_ERR_SCRIPT)
    env = dict(os.environ, NP_REPO=root, NP_REF_DROPIN_LIB=DROPIN)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert "OK " in r.stdout and "BAD" not in r.stdout, r.stdout + r.stderr          # the process ended inside the failing call
    assert r.returncode != 0 and "nanopolish_amd:" in r.stderr and "failed" in r.stderr
    r = subprocess.run([sys.executable, str(script)], env=dict(env, NP_DROPIN_ERRORS_IN_BAND="1"), capture_output=True, text=True, timeout=600)
    assert "OK " in r.stdout and "BAD -inf" in r.stdout, r.stdout + r.stderr         # the in-band "no result" ...
    assert r.returncode != 0 and "device call(s) failed" in r.stderr                  # ... and still no exit status 0
