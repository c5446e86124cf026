"""Edge cases through the C ABI on the GPU: empty and ragged batches, minimum/maximum sizes, argument errors."""
import numpy as np
import pytest

from cases import K, HAF_PRE, HAF_POST, synth_read

pytestmark = pytest.mark.gpu


def test_empty_batches(ctx):
    assert len(ctx.profile_hmm_score([])) == 0
    assert ctx.adaptive_banded_simple_event_align([]) == []
    assert ctx.profile_hmm_align([]) == []


def test_ragged_event_align_batch(ctx, orc, models):
    """reads from 7 to 6000 bases in one call (tiny reads never fill the 100-wide band)."""
    mn = orc.model(models["nucleotide"])
    reads = [synth_read(200 + i, models["nucleotide"], L=L) for i, L in enumerate((60, 6000, 101, 7, 900, 250, 13, 3300))]
    jobs, want = [], []
    for rd in reads:
        sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        jobs.append(dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=1.0))
        want.append(orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"]))
    got = ctx.adaptive_banded_simple_event_align(jobs)
    n_ok = 0
    for rd, g, w in zip(reads, got, want):
        if w is None:                      # the reference would read its trace out of bounds here
            assert len(g) == 0
            continue
        assert np.array_equal(g, w), len(rd["seq"])
        n_ok += len(w) > 0
    assert n_ok >= 5


def test_event_align_band_phase_boundaries(ctx, orc, models):
    """Reads that push the band along unusual paths, so that kernel A's three phases (generic / FAST middle / generic) hand
    over at odd places: over-segmented reads (3 events per k-mer: the band runs along the event axis), under-segmented
    reads (fewer events than k-mers), reads barely longer than the band, a 20k-base read, and event streams whose
    second half belongs to another read (the alignment leaves the diagonal and fails QC)."""
    mn = orc.model(models["nucleotide"])
    rng = np.random.default_rng(9)
    cases = []
    for i, L in enumerate((1500, 1500, 1500, 106, 140, 20000)):
        rd = synth_read(400 + i, models["nucleotide"], L=L)
        ev = rd["events"]
        if i == 0:
            ev = (np.repeat(ev, 2) + rng.normal(0, 0.3, 2 * len(ev))).astype(np.float32)         # ~3 events per k-mer
        elif i == 1:
            ev = ev[::2].copy()                                                                   # ~0.7 events per k-mer
        elif i == 2:
            other = synth_read(450, models["nucleotide"], L=L)["events"]
            ev = np.concatenate([ev[:len(ev) // 2], other[len(other) // 2:]]).astype(np.float32)
        cases.append((rd, ev))
    jobs, want = [], []
    for rd, ev in cases:
        sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], ev)
        jobs.append(dict(events=ev, ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=1.0))
        want.append(orc.event_align(mn, orc.scalings(sh, sc, 1.0), ev, rd["ranks"]))
    got = ctx.adaptive_banded_simple_event_align(jobs)
    n_ok = 0
    for (rd, ev), g, w in zip(cases, got, want):
        if w is None:
            assert len(g) == 0
            continue
        assert np.array_equal(g, w), (len(rd["seq"]), len(ev))
        n_ok += len(w) > 0
    assert n_ok >= 3


def test_hmm_window_extremes(ctx, orc, models):
    """1- and 2-event windows, the largest supported sequence (1024 k-mers), events_per_base extremes, both strides."""
    mn = orc.model(models["nucleotide"])
    rd = synth_read(300, models["nucleotide"], L=3000)
    S = orc.scalings(rd["shift"], rd["scale"], rd["var"])
    cases = []
    for n_k, e1, e2, stride, epb in ((1, 10, 10, 1, 1.5), (1, 10, 11, 1, 1.5), (16, 100, 100, 1, 1.5), (16, 120, 101, -1, 1.5),
                                     (5, 50, 90, 1, 0.4), (40, 400, 470, 1, 6.5), (1000, 100, 1700, 1, 1.6),
                                     (1024, 1900, 150, -1, 1.7), (700, 10, 20, 1, 1.5)):
        ranks = rd["ranks"][30:30 + n_k]
        for flags in (0, HAF_PRE | HAF_POST):
            cases.append((ranks, e1, e2, stride, epb, flags))
    jobs = [dict(events=rd["events"], ranks=r, e_start=e1, e_stop=e2, stride=st, model=ctx.models["nucleotide"], scale=rd["scale"],
                 shift=rd["shift"], var=rd["var"], events_per_base=epb, flags=fl) for (r, e1, e2, st, epb, fl) in cases]
    want = np.array([orc.hmm_score(mn, S, rd["events"], r.astype(np.uint32), e1, e2, st, epb, 1.0, fl)
                     for (r, e1, e2, st, epb, fl) in cases], np.float32)
    got = ctx.profile_hmm_score(jobs)
    assert np.array_equal(got, want, equal_nan=True)
    assert np.isfinite(want).all()       # the K (skip) states keep every window reachable, however lopsided


def test_viterbi_on_lopsided_windows(ctx, orc, models):
    """100 k-mers against 11 events (long K-state chains) and 3 k-mers against 300 events (long stays), no clipping."""
    mn = orc.model(models["nucleotide"])
    rd = synth_read(301, models["nucleotide"], L=600)
    S = orc.scalings(rd["shift"], rd["scale"], rd["var"])
    for ranks, e1, e2 in ((rd["ranks"][10:110], 20, 30), (rd["ranks"][50:53], 100, 399), (rd["ranks"][5:7], 9, 10)):
        want = orc.hmm_align(mn, S, rd["events"], ranks.astype(np.uint32), e1, e2, 1, 1.5)
        got = ctx.profile_hmm_align([dict(events=rd["events"], ranks=ranks, e_start=e1, e_stop=e2, stride=1, model=ctx.models["nucleotide"],
                                          scale=rd["scale"], shift=rd["shift"], var=rd["var"], events_per_base=1.5, flags=0)])[0]
        assert want is not None and all(np.array_equal(a, b) for a, b in zip(got, want))


def test_argument_errors(ctx, models):
    rd = synth_read(302, models["nucleotide"], L=400)
    base = dict(events=rd["events"], ranks=rd["ranks"][:20], e_start=10, e_stop=40, stride=1, model=ctx.models["nucleotide"],
                scale=1.0, shift=0.0, var=1.0, events_per_base=1.5, flags=0)
    with pytest.raises(RuntimeError):
        ctx.profile_hmm_score([dict(base, stride=-1)])                    # stride disagrees with e_start/e_stop
    with pytest.raises(RuntimeError):
        ctx.profile_hmm_score([dict(base, e_stop=10 ** 6)])               # beyond the read's events
    with pytest.raises(RuntimeError):
        ctx.profile_hmm_score([dict(base, ranks=np.zeros(1025, np.uint16))])   # > NP_MAX_KMERS
    with pytest.raises(RuntimeError):
        ctx.profile_hmm_score([dict(base, model=99)])
    assert np.isfinite(ctx.profile_hmm_score([base])[0])                  # the context survives the errors


def test_host_entry_points_are_thread_safe(ctx, orc, models):
    """the reference calls profile_hmm_score / adaptive_banded_simple_event_align concurrently from its OpenMP workers
    (bam_processor.cpp:99); the single-call entry points of one context must give the serial answers from 8 threads at once"""
    from concurrent.futures import ThreadPoolExecutor
    from cases import synth_read, methylation_jobs, K
    from nanopolish_amd import api
    mn = orc.model(models["nucleotide"])
    work = []
    for rid in range(8):
        rd = synth_read(700 + rid, models["nucleotide"], L=900)
        sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        pairs = orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"])
        epb, jobs = methylation_jobs(orc, rd, pairs)
        hj = [dict(events=rd["events"], ranks=api.sequence_kmer_ranks("cpg", j["subseq"], j["rc_subseq"], K, j["rc"]), e_start=j["e1"],
                   e_stop=j["e2"], stride=j["stride"], model=ctx.models["cpg"], scale=rd["scale"], shift=rd["shift"], var=rd["var"],
                   events_per_base=epb, flags=3) for j in jobs]
        aj = dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=1.0)
        work.append((hj, aj))
    serial = [(ctx.profile_hmm_score(hj), ctx.adaptive_banded_simple_event_align([aj])[0]) for hj, aj in work]

    def one(i):
        hj, aj = work[i % len(work)]
        return ctx.profile_hmm_score(hj), ctx.adaptive_banded_simple_event_align([aj])[0]
    with ThreadPoolExecutor(8) as ex:
        par = list(ex.map(one, range(32)))
    for i, (s, p) in enumerate(par):
        assert np.array_equal(s, serial[i % len(work)][0]) and np.array_equal(p, serial[i % len(work)][1])
