"""Long and ultra-long reads (parity was untested beyond 6 000-base reads): the event aligner bit-exact against the
oracle -- and against the reference's own adaptive_banded_simple_event_align compiled in place (oracle/_ref/libnp_ref.so, when it
travelled with the repo) -- for reads of 33 k and 100 k bases and for one read of more than a million events, the last one inside
a batch large enough to drive the 48 GB scratch budget (np_capi.hip:run_event_align; the reference allocates
(n_events + n_kmers + 2) x 100 x 5 bytes per read, src/nanopolish_raw_loader.cpp:123-138); and the whole call-methylation pass
(work items on the device, alignment, calibration, scoring) on 33 k-base reads against the oracle's per-read pass."""
import numpy as np
import pytest

from cases import synth_read, call_methylation_read

pytestmark = pytest.mark.gpu


def _want(orc, mn, rd):
    sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
    return sh, sc, orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"])


def _ref_pairs(rd, sh, sc):
    from oracle import RefOracle, have_ref
    if not have_ref():
        return None
    ref = RefOracle()
    eo = np.array([0, len(rd["events"])], np.int64)
    pairs, pair_off, n_pairs = ref.align_many([rd["seq"]], rd["events"], eo, np.array([sh]), np.array([sc]), 1)
    return pairs[:n_pairs[0]]


def test_33k_and_100k_base_reads_bit_exact(ctx, orc, models):
    mn = orc.model(models["nucleotide"])
    reads = [synth_read(7000 + L, models["nucleotide"], L=L) for L in (33000, 100000, 33001)]
    jobs, want = [], []
    for rd in reads:
        sh, sc, w = _want(orc, mn, rd)
        jobs.append(dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=1.0))
        want.append((sh, sc, w))
    got = ctx.adaptive_banded_simple_event_align(jobs)
    for rd, g, (sh, sc, w) in zip(reads, got, want):
        assert w is not None and len(w) > len(rd["ranks"])
        assert np.array_equal(g, w), len(rd["seq"])
        r = _ref_pairs(rd, sh, sc)
        if r is not None:
            assert np.array_equal(g, r), "differs from the reference's own aligner (%d bases)" % len(rd["seq"])


def test_million_event_read_inside_a_batch_that_drives_the_scratch_budget(ctx, orc, models):
    """One read of ~1.03 M events (700 k bases) and 1 150 short reads in ONE call: the per-wave trace / parameter slabs are sized
    by the longest read (~83 MB per resident wave), so the full persistent grid would need far more than the 48 GB budget and
    the launch shrinks its grid; every read must still come out bit-exact."""
    mn = orc.model(models["nucleotide"])
    big = synth_read(7777, models["nucleotide"], L=700000)
    assert len(big["events"]) > 1000000
    small = [synth_read(8000 + i, models["nucleotide"], L=150 + (i % 7) * 40) for i in range(1150)]
    reads = [small[0], big] + small[1:]
    jobs, want = [], []
    for rd in reads:
        sh, sc, w = _want(orc, mn, rd)
        jobs.append(dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=1.0))
        want.append((sh, sc, w))
    got = ctx.adaptive_banded_simple_event_align(jobs)
    full_grid = min(ctx.get_stat("align_blocks_max"), (len(reads) + 3) // 4)
    assert ctx.get_stat("align_blocks") < full_grid, "the batch did not reach the scratch budget"
    assert ctx.get_stat("align_scratch_bytes") <= 48 << 30
    n_ok = 0
    for rd, g, (sh, sc, w) in zip(reads, got, want):
        if w is None:
            assert len(g) == 0
            continue
        assert np.array_equal(g, w), len(rd["seq"])
        n_ok += len(w) > 0
    assert n_ok > 1000 and len(got[1]) > 1000000
    r = _ref_pairs(big, want[1][0], want[1][1])
    if r is not None:
        assert np.array_equal(got[1], r)


def test_call_methylation_pass_on_33k_base_reads(ctx, orc, models):
    """The device-resident pass on a ragged batch with 33 k-base reads (~48 k events: ~6x the bench's read): pairs bit-exact and
    every CpG group's two scores equal to the oracle's per-read pass (windows deep inside a long read: event indices ~48 000)."""
    from nanopolish_amd.pipeline import build_host_batch, CallMethylationBatch
    ids = [9100, 9101, 9102, 9103]
    L = [33000, 1200, 33000, 5450]
    hb = build_host_batch(models, ids, L=L, with_jobs=False)
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=False, jobs_on_device=True)
    batch.step()
    mn = orc.model(models["nucleotide"]); mc = orc.model(models["cpg"])
    n_groups = 0
    for i, rd in enumerate(hb["reads"]):
        want = call_methylation_read(orc, mn, mc, rd)
        assert np.array_equal(batch.pairs_of(i), want["pairs"]), L[i]
        first, n_motif, su, sm = batch.groups_of(i)
        got = {int(f): (u, m) for f, u, m in zip(first, su, sm) if np.isfinite(u)}
        assert len(want["first"]) > 0
        for f, u, m in zip(want["first"], want["unmeth"], want["meth"]):
            assert got[int(f)] == (u, m), (L[i], f)
            n_groups += 1
        assert len(got) == len(want["first"])
    assert n_groups > 1500
