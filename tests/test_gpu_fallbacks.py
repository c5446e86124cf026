"""Shipped fallbacks that a normal r9.4 run never takes (VERDICT r4 Missing 2b): the event aligner's ALL-GENERIC path.

Kernel A's FAST pair loop rests on every DP cell being <= 0 (Suzuki's move rule as one unsigned compare of the two band ends,
csrc/np_align_kernel.hip "R.nonpos").  An emission is cl - a^2/2 with cl = log(1/sqrt(2 pi)) - log(sigma'): for a scaled sigma
below 0.399 pA cl is positive, cells can be positive, and the read takes the generic band step for EVERY band -- what
src/nanopolish_raw_loader.cpp:179-195,259-274 does unconditionally.  With the r9.4 kit that only ever happened for the ~300
bands at a read's ends; here whole reads go through it, in the three ways a production run could get there:
  * a read whose scaled variance is small (set4's `var`), compared with the REFERENCE's own aligner compiled in place and the port;
  * a narrow custom model (np_register_model admits level_stdv down to 1/16), compared with the port run on the same model;
  * a mix of both kinds of reads in one launch (the flag is per read: neighbours in the persistent queue must not leak state).
"""
import numpy as np
import pytest

from cases import synth_read

pytestmark = pytest.mark.gpu

NARROW = 0.15      # scaled sigma = 0.15 x the kit's 1.0 ... 4.5 pA: cl > 0 for three quarters of the k-mers


def _narrow_read(read_id, model, L, var_n=NARROW):
    """A synthetic read whose event noise is var_n x the model's level_stdv, so that it aligns under that variance -- and in which
    EVERY k-mer has at least one event: the reference's QC sums the emission of every cell of the path (raw_loader.cpp:338-341),
    the cell a k-mer skip lands on included, and under a 0.2 pA sigma one such cell costs hundreds of log units."""
    rd = synth_read(read_id, model, L=L)
    rng = np.random.default_rng(0xFA11 + read_id)
    K = len(rd["ranks"])
    cnt = (1 + (rng.random(K) < 0.45) + (rng.random(K) < 0.15)).astype(np.int64)
    rk = np.repeat(rd["ranks"], cnt)
    mu = rd["scale"] * model["level_mean"][rk] + rd["shift"]
    ev = (mu + var_n * model["level_stdv"][rk] * rng.standard_normal(len(rk))).astype(np.float32)
    return dict(rd, events=ev, var=var_n)


def _cl_max(model, var):
    return float(np.max(np.log(0.3989422804014327) - (model["level_log_stdv"] + np.log(var))))


@pytest.mark.parametrize("L", [700, 2500, 8000])
def test_all_generic_aligner_small_variance_vs_reference_and_port(ctx, orc, models, L):
    from oracle import RefOracle, have_ref
    mn = orc.model(models["nucleotide"])
    assert _cl_max(models["nucleotide"], NARROW) > 0.5            # the reads below cannot take the FAST loop
    ref = RefOracle() if have_ref() else None
    reads = [_narrow_read(900 + 10 * (L // 700) + i, models["nucleotide"], L) for i in range(4)]
    jobs, moms = [], []
    for rd in reads:
        sh, sc = rd["shift"], rd["scale"]        # the generator's own scalings: under a 0.2 pA sigma the method-of-moments estimate is
        jobs.append(dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=NARROW))   # pA off and QC fails
        moms.append((sh, sc))
    got = ctx.adaptive_banded_simple_event_align(jobs)
    n_ok = 0
    for rd, (sh, sc), g in zip(reads, moms, got):
        want = orc.event_align(mn, orc.scalings(sh, sc, NARROW), rd["events"], rd["ranks"])
        assert want is not None
        assert g.shape == want.shape and np.array_equal(g, want), "read %d vs the port" % rd["read_id"]
        if ref is not None:
            w2 = ref.event_align(rd["events"], rd["seq"], sh, sc, var=NARROW)
            assert np.array_equal(g, w2), "read %d vs the reference" % rd["read_id"]
        n_ok += len(want) > 0
    assert n_ok >= 3                                               # they align (QC passes): the comparison is not of empty results


def test_all_generic_aligner_narrow_custom_model(orc, models):
    """level_stdv x 0.12 registered as its own model: var = 1, sigma' < 0.4 for most k-mers -- the route a custom kit would take."""
    from nanopolish_amd.api import Context
    m = models["nucleotide"]
    f = 0.12
    assert float(m["level_stdv"].min()) * f >= 0.0625              # inside np_register_model's verified range
    narrow = dict(k=6, level_mean=m["level_mean"].copy(), level_stdv=m["level_stdv"] * f, level_log_stdv=np.log(m["level_stdv"] * f))
    c = Context(0)
    try:
        mid = c.register_model(narrow, "narrow")
        mo = orc.model(narrow)
        reads = [_narrow_read(960 + i, m, L, var_n=f) for i, L in enumerate((700, 1300, 2500, 5450))]
        jobs, want = [], []
        for rd in reads:
            sh, sc = rd["shift"], rd["scale"]
            jobs.append(dict(events=rd["events"], ranks=rd["ranks"], model=mid, scale=sc, shift=sh, var=1.0))
            want.append(orc.event_align(mo, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"]))
        got = c.adaptive_banded_simple_event_align(jobs)
        for rd, g, w in zip(reads, got, want):
            assert w is not None and len(w) > 0
            assert np.array_equal(g, w), rd["read_id"]
    finally:
        c.close()


def test_generic_and_fast_reads_share_a_launch(ctx, orc, models):
    """nonpos is a property of the READ: generic and FAST reads interleaved in one persistent launch, more reads than a wave's
    worth of tickets per workgroup, each compared with the port."""
    mn = orc.model(models["nucleotide"])
    reads, var = [], []
    for i in range(24):
        L = 600 + 97 * i
        if i % 3 == 1:
            reads.append(_narrow_read(1000 + i, models["nucleotide"], L)); var.append(NARROW)
        else:
            reads.append(synth_read(1000 + i, models["nucleotide"], L=L)); var.append(1.0)
    jobs, want = [], []
    for rd, v in zip(reads, var):
        sh, sc = (rd["shift"], rd["scale"]) if v != 1.0 else orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        jobs.append(dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=v))
        want.append(orc.event_align(mn, orc.scalings(sh, sc, v), rd["events"], rd["ranks"]))
    got = ctx.adaptive_banded_simple_event_align(jobs)
    n_ok = 0
    for rd, g, w in zip(reads, got, want):
        assert w is not None
        assert np.array_equal(g, w), rd["read_id"]
        n_ok += len(w) > 0
    assert n_ok >= 20
