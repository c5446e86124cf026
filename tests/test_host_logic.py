"""CPU-only checks of the product's host side: the C-ABI library loads, exports every symbol include/np_hmm.h
declares, refuses to run without a GPU (no CPU fallback), and its host helpers agree with the oracle / goldens."""
import ctypes as C
import os
import sys
import re

import numpy as np
import pytest

from cases import K, methylation_jobs, synth_read, revcomp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from nanopolish_amd.lib import load_library, build_library, library_path
    if not os.path.exists(library_path()):
        build_library()
    return load_library()


def test_library_exports_every_declared_symbol(L):
    from nanopolish_amd.lib import SYMBOLS
    hdr = open(os.path.join(ROOT, "include", "np_hmm.h")).read()
    declared = set(re.findall(r"\b(np_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(SYMBOLS), (declared ^ set(SYMBOLS))
    for s in declared:
        assert hasattr(L, s), s


def test_struct_layouts_match_header(L):
    from nanopolish_amd import lib
    assert C.sizeof(lib.ReadDev) == 136 and C.sizeof(lib.HmmJobDev) == 32 and C.sizeof(lib.Pair) == 8
    assert C.sizeof(lib.HmmState) == 24 and C.sizeof(lib.HmmJob) == 88 and C.sizeof(lib.AlignJob) == 56


def test_no_cpu_fallback_without_device(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from nanopolish_amd.api import Context
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Context(0)


def test_alphabet_helpers_match_goldens(L):
    from nanopolish_amd import api
    t = np.load(os.path.join(ROOT, "tests", "golden", "golden_tables.npz"))
    for a, s, rc, me, un, mo, r6 in zip(t["kat_alphabet"], t["kat_in"], t["kat_rc"], t["kat_meth"], t["kat_unmeth"],
                                        t["kat_motif"], t["kat_rank6"]):
        a, s = str(a), str(s)
        assert api.reverse_complement(a, s) == str(rc)
        base = s.replace("M", "A") if a == "dam" else s.replace("M", "C")
        assert api.methylate(a, base) == str(me)
        assert api.unmethylate(a, s) == str(un)
        assert "".join("1" if api.is_motif_match(a, s, i) else "0" for i in range(max(len(s) - 1, 0))) == str(mo)
        if r6 >= 0:
            assert api.kmer_rank(a, s[:6]) == r6


def test_transitions_and_logf_match_host_libm(L, orc):
    """np_logf.h restates glibc logf; calculate_transitions must equal the oracle's (which calls libm logf)."""
    from nanopolish_amd import api
    rng = np.random.default_rng(3)
    epbs = np.concatenate([rng.uniform(0.0, 6.0, 20000), np.linspace(1.0, 3.0, 20001), [0.0, 1.25, 5.0, 50.0]])
    for bias in (1.0, 0.9, 0.8):
        for e in epbs[:: (1 if bias == 1.0 else 7)]:
            assert np.array_equal(api.calculate_transitions(float(e), bias), orc.calculate_transitions(float(e), bias)), (e, bias)


def test_mom_ranks_and_jobs_match_oracle(L, orc, models):
    from nanopolish_amd import api
    mn = orc.model(models["nucleotide"])
    for rid in (0, 1, 2, 3):
        rd = synth_read(rid, models["nucleotide"], L=1500)
        assert api.estimate_scalings_using_mom(models["nucleotide"], rd["ranks"], rd["events"]) == \
            orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        ref_seq = revcomp(rd["seq"]) if rd["rc"] else rd["seq"]
        assert api.reverse_complement("nucleotide", rd["seq"]) == revcomp(rd["seq"])
        jb = api.cm_build_jobs_identity(ref_seq, rd["rc"])
        f, l, c = orc.scan_motif_groups("cpg", ref_seq)
        f2, l2, c2 = api.scan_motif_groups("cpg", ref_seq)
        assert np.array_equal(f, f2) and np.array_equal(l, l2) and np.array_equal(c, c2)
        # every oracle work item appears in the product's list with identical k-mer ranks and bounding k-mers
        sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        pairs = orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"])
        start, stop, epb = orc.build_base_to_event_map(pairs, len(rd["ranks"]))
        epb2, jobs = methylation_jobs(orc, rd, pairs)
        firsts = list(jb["first"])
        assert len(jobs) > 10
        for j in jobs:
            i = firsts.index(j["first"])
            lo, hi = jb["rank_off"][i], jb["rank_off"][i + 1]
            assert np.array_equal(jb["ranks_unmeth"][lo:hi], orc.sequence_kmer_ranks("cpg", j["subseq"], j["rc_subseq"], K, j["rc"]))
            assert np.array_equal(jb["ranks_meth"][lo:hi], orc.sequence_kmer_ranks("cpg", j["m_subseq"], j["rc_m_subseq"], K, j["rc"]))
            assert orc.get_closest_event_to(start, jb["kpos"][i, 0]) == j["e1"]
            assert orc.get_closest_event_to(start, jb["kpos"][i, 1]) == j["e2"]


def test_fill_read_host_constants(L):
    from nanopolish_amd import lib
    import math
    r = lib.ReadDev()
    L.np_fill_read_host(C.byref(r), 1.5, 1.05, 1.2, 10, 8000, 20, 5445)
    epk = 8000 / 5445
    p_stay = 1 - (1 / (epk + 1))
    assert r.lp_skip == math.log(1e-10) and r.lp_stay == math.log(p_stay) and r.lp_trim == math.log(0.01)
    assert r.lp_step == math.log(1.0 - math.exp(r.lp_skip) - math.exp(r.lp_stay))
    assert (r.shift, r.scale, r.var, r.log_var) == (1.5, 1.05, 1.2, math.log(1.2))
    assert (r.event_off, r.n_events, r.rank_off, r.n_kmers) == (10, 8000, 20, 5445)


def test_oracle_recalibrate_solves_the_weighted_normal_equations(orc, models):
    """f1 restatement (its 2x2 Eigen solve is the one unpinned step, see tests/test_oracle_vs_ref_full.py for the rest): check npo_recalibrate against an
    independent numpy solve of the same weighted least squares on the same 'M' entries, and that it recovers the
    planted scalings of a synthetic read; < 200 'M' entries -> not recalibrated."""
    from cases import synth_read
    mn = orc.model(models["nucleotide"])
    nuc = models["nucleotide"]
    rd = synth_read(7, nuc, L=3000)
    sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
    pairs = orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"])
    start, stop, _ = orc.build_base_to_event_map(pairs, len(rd["ranks"]))
    got = orc.recalibrate(mn, rd["events"], rd["ranks"], start, stop)
    assert got is not None
    # independent: 'M' entries = first event of each k-mer that has events and whose rank differs from the previous such k-mer
    ks = [k for k in range(len(start)) if start[k] != -1]
    m_k = [k for q, k in enumerate(ks) if q == 0 or rd["ranks"][k] != rd["ranks"][ks[q - 1]]]
    mu = nuc["level_mean"][rd["ranks"][m_k]]; sd = nuc["level_stdv"][rd["ranks"][m_k]]
    e = rd["events"][start[m_k]].astype(np.float64)
    w = 1.0 / sd ** 2
    A = np.array([[w.sum(), (mu * w).sum()], [(mu * w).sum(), (mu * mu * w).sum()]])
    b = np.array([(e * w).sum(), (mu * e * w).sum()])
    x = np.linalg.solve(A, b)
    var = np.sqrt((((e - x[0] - x[1] * mu) / sd) ** 2).sum() / len(m_k))
    assert np.allclose(got, (x[0], x[1], var), rtol=1e-9, atol=1e-9)
    assert abs(got[1] - rd["scale"]) < 0.05 and abs(got[0] - rd["shift"]) < 3.0
    short = synth_read(8, nuc, L=150)
    sh, sc = orc.estimate_scalings_mom(mn, short["ranks"], short["events"])
    p2 = orc.event_align(mn, orc.scalings(sh, sc, 1.0), short["events"], short["ranks"])
    s2, t2, _ = orc.build_base_to_event_map(p2, len(short["ranks"]))
    assert orc.recalibrate(mn, short["events"], short["ranks"], s2, t2) is None


def test_restated_glibc_log_exp_match_host_libm(L):
    """csrc/np_log.h restates glibc's double log/exp (FMA build); the device computes the aligner's per-read constants and
    set4's log(var) with it.  Checked against this host's libm (scalar calls through ctypes) and, through
    np_aligner_constants, against np_fill_read_host, which calls libm as the reference does."""
    import ctypes as C
    from nanopolish_amd import lib as _l
    m = C.CDLL("libm.so.6")
    m.log.restype = C.c_double; m.log.argtypes = [C.c_double]; m.exp.restype = C.c_double; m.exp.argtypes = [C.c_double]
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(1e-12, 3, 60000), np.exp(rng.uniform(-30, 30, 30000)), 1 + rng.uniform(-0.07, 0.07, 60000),
                        [1.0, 0.5, 2.0, 1e-10, 0.01, 1 - 1e-10, 0.9375, 1.06469]])
    ol = np.zeros_like(x); oe = np.zeros_like(x)
    L.np_restated_log_exp(x.ctypes.data_as(_l.c_f64p), len(x), ol.ctypes.data_as(_l.c_f64p), oe.ctypes.data_as(_l.c_f64p))
    assert np.array_equal(ol, np.array([m.log(float(v)) for v in x]))
    dom = x < 512.0                                           # np_exp_glibc covers |x| < 512 (log-probabilities)
    assert np.array_equal(oe[dom], np.array([m.exp(-float(v)) for v in x[dom]])) and np.isnan(oe[~dom]).all()
    rd = _l.ReadDev(); out = np.zeros(4)
    for ne in range(50, 30000, 61):
        for nk in (97, 1000, 5445):
            L.np_fill_read_host(C.byref(rd), 0.0, 1.0, 1.0, 0, ne, 0, nk)
            L.np_aligner_constants(ne, nk, out.ctypes.data_as(_l.c_f64p))
            assert (out[0], out[1], out[2], out[3]) == (rd.lp_skip, rd.lp_stay, rd.lp_step, rd.lp_trim), (ne, nk)


def test_restated_libm_matches_this_hosts_libm():
    """The device computes log / exp / logf with restatements of glibc 2.35's x86-64 FMA variants (csrc/np_log.h, np_logf.h).
    On a host with another libm the REFERENCE would compute its constants with that libm; this test says whether the two agree
    here (they do in the image this repository is verified in: 0 mismatches over 3 x 2M arguments)."""
    import ctypes as C
    from nanopolish_amd import lib as _l
    bad = C.c_uint64(1)
    assert _l.load_library().np_selftest_libm(2000000, 20260924, C.byref(bad)) == 0
    assert bad.value == 0, "%d of 6M restated log/exp/logf values differ from this host's libm" % bad.value


def test_host_batches_concatenate_and_tile_like_one_build(models):
    """bench.py prepares its reads in chunks on a pool of processes and tiles the pool: joining chunks (concat_host_batches)
    must give what one build over the whole id range gives, and a tiled batch must be its copies back to back with every
    offset shifted (tile_host_batch)."""
    from nanopolish_amd.pipeline import build_host_batch, concat_host_batches, tile_host_batch
    ids = np.arange(6)
    whole = build_host_batch(models, ids, L=400, raw=True, adc=True)
    parts = concat_host_batches([build_host_batch(models, ids[:2], L=400, raw=True, adc=True),
                                 build_host_batch(models, ids[2:5], L=400, raw=True, adc=True),
                                 build_host_batch(models, ids[5:], L=400, raw=True, adc=True)])
    assert parts["n"] == whole["n"] == 6
    for key in ("events", "ranks", "job_ranks", "kpos", "event_off", "rank_off", "job_off", "raw", "raw_off", "adc", "adc_offset", "adc_unit"):
        assert np.array_equal(parts[key], whole[key]), key
    for key in ("reads_a", "reads_b", "jobs"):
        assert parts[key].tobytes() == whole[key].tobytes(), key
    assert parts["ref_seqs"] == whole["ref_seqs"]

    t = tile_host_batch(whole, 3)
    assert t["n"] == 18 and len(t["events"]) == 3 * len(whole["events"]) and len(t["jobs"]) == 3 * len(whole["jobs"])
    ne, nr = len(whole["events"]), len(whole["ranks"])
    for c in range(3):
        a = t["reads_a"][6 * c:6 * (c + 1)]
        assert np.array_equal(a["event_off"], whole["reads_a"]["event_off"] + c * ne)
        assert np.array_equal(a["rank_off"], whole["reads_a"]["rank_off"] + c * nr)
        assert np.array_equal(t["events"][c * ne:(c + 1) * ne], whole["events"])
        j = t["jobs"][len(whole["jobs"]) * c:len(whole["jobs"]) * (c + 1)]
        assert np.array_equal(j["read"], whole["jobs"]["read"] + 6 * c)
    assert np.array_equal(t["event_off"], np.concatenate([whole["event_off"][:-1] + c * ne for c in range(3)] + [[3 * ne]]))


def test_bench_host_helpers():
    """the bench's read-length distribution (deterministic in the read id, clipped, the requested mean) and the CPU count it
    sizes the reference's OpenMP run with (affinity mask capped by the cgroup quota)"""
    import bench
    from nanopolish_amd.hostinfo import usable_cores
    ids = np.arange(20000)
    a = bench.ragged_lengths(ids, 5500)
    assert np.array_equal(a, bench.ragged_lengths(ids, 5500)) and np.array_equal(a[100:200], bench.ragged_lengths(ids[100:200], 5500))
    assert a.min() >= 600 and a.max() <= 40000 and abs(a.mean() - 5500) < 120 and a.std() > 2000
    visible, quota, eff = usable_cores()
    assert visible >= 1 and 1 <= eff <= visible and (quota is None or eff <= max(1, int(quota + 0.5)))


def test_align_kernel_isa_guard_accepts_the_shipped_build_and_rejects_a_spilling_queue(tmp_path):
    """tools/check_align_isa.py (run by the Makefile on every build of np_align_kernel.hip): the shipped kernel passes; the build with a
    trace prefetch queue deeper than the register budget holds (NP_BT_DEPTH = 12: the compiler copies a queue entry right after the
    inline-asm load that fills it -- the configuration that once walked garbage on the GPU) is rejected."""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "nanopolish_amd", "csrc", "np_align_kernel.hip")
    flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-S", "--cuda-device-only"]
    for extra, want_ok in (([], True), (["-DNP_BT_DEPTH=12"], False)):
        out = str(tmp_path / ("a%d.s" % len(extra)))
        subprocess.run([hipcc] + flags + extra + [src, "-o", out], check=True, capture_output=True)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_align_isa.py"), out], capture_output=True, text=True)
        assert (r.returncode == 0) == want_ok, r.stderr[-600:]
        if not want_ok:
            assert "may still be in flight" in r.stderr


def test_every_option_and_environment_switch_of_the_library_is_documented():
    """np_set_option's names against include/np_hmm.h, and every NP_* variable the library or the shims read against the header and
    INTEGRATION.md / DESIGN.md / README.md: a knob a maintainer can trip over has a sentence somewhere."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "nanopolish_amd", "csrc", "np_capi.hip")).read()
    hdr = open(os.path.join(root, "include", "np_hmm.h")).read()
    opts = re.findall(r'k == "([a-z_0-9]+)"\) c->', src)
    assert len(opts) >= 10
    assert [o for o in opts if '"%s"' % o not in hdr] == []
    envs = set()
    for f in sum((glob.glob(os.path.join(root, "nanopolish_amd", "csrc", e)) for e in ("*.hip", "*.cpp", "*.h")), []):
        envs |= set(re.findall(r'getenv\("(NP_[A-Z_0-9]+)"\)', open(f).read()))
    docs = hdr + "".join(open(os.path.join(root, n)).read() for n in ("INTEGRATION.md", "DESIGN.md", "README.md"))
    assert len(envs) >= 15
    assert sorted(e for e in envs if e not in docs) == []
