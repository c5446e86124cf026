"""GPU: randomised stress of the two hot kernels and the Viterbi path against the CPU oracle -- many small problems with
unusual shapes (events per base from 0.4 to 4, reads of 12 to 900 bases, heavy noise that trips the aligner's QC, windows
of 6 to 300 k-mers over 2 to 600 events, both strides, every clip-flag combination).  Bit-exact or fail."""
import numpy as np
import pytest

from nanopolish_amd.synth import nucleotide_kmer_ranks, BASES

pytestmark = pytest.mark.gpu
K = 6


def _random_read(rng, nuc, L, rate, noise):
    codes = rng.integers(0, 4, L)
    ranks = nucleotide_kmer_ranks(codes, K)
    n_ev = rng.poisson(rate, len(ranks))
    shift = rng.uniform(-8, 8); scale = rng.uniform(0.8, 1.2); var = rng.uniform(0.8, 1.6)
    rk = np.repeat(ranks, n_ev)
    ev = (scale * nuc["level_mean"][rk] + shift + noise * var * nuc["level_stdv"][rk] * rng.standard_normal(len(rk))).astype(np.float32)
    return dict(seq=BASES[codes].tobytes().decode(), ranks=ranks, events=ev, shift=shift, scale=scale, var=var)


def test_event_align_fuzz(ctx, orc, models):
    nuc = models["nucleotide"]
    mn = orc.model(nuc)
    rng = np.random.default_rng(2024)
    reads = []
    while len(reads) < 400:
        L = int(rng.integers(12, 900))
        rd = _random_read(rng, nuc, L, rate=float(rng.choice([0.4, 0.8, 1.5, 2.5, 4.0])), noise=float(rng.choice([0.5, 1.0, 1.0, 3.0, 8.0])))
        if len(rd["events"]) >= 2:
            reads.append(rd)
    jobs, moms = [], []
    for rd in reads:
        sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        moms.append((sh, sc))
        jobs.append(dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=1.0))
    got = ctx.adaptive_banded_simple_event_align(jobs)
    n_ok = n_fail = 0
    for rd, (sh, sc), g in zip(reads, moms, got):
        want = orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"])
        if want is None:                       # the reference walks off its band there (known deviation 1)
            assert len(g) == 0
            continue
        assert g.shape == want.shape and np.array_equal(g, want), (len(rd["events"]), len(rd["ranks"]))
        n_ok += len(want) > 0; n_fail += len(want) == 0
    assert n_ok > 80 and n_fail > 20           # both outcomes of the QC are exercised


def test_event_align_more_reads_than_resident_waves(ctx, orc, models):
    """A batch larger than the persistent grid: waves take reads from the queue one after the other, longest first
    (align_lpt) or in index order -- same pairs either way, and the oracle's.  (The grid is shrunk to 1024 waves so that
    1300 short reads are such a batch; two reads are far longer than the rest.)"""
    nuc = models["nucleotide"]
    mn = orc.model(nuc)
    rng = np.random.default_rng(4242)
    reads = []
    while len(reads) < 1300:
        L = 4000 if len(reads) in (7, 700) else int(rng.integers(12, 700))
        rd = _random_read(rng, nuc, L, rate=float(rng.choice([0.8, 1.5, 2.5])), noise=float(rng.choice([0.5, 1.0, 1.0, 3.0, 8.0])))
        if len(rd["events"]) >= 2:
            reads.append(rd)
    jobs = []
    for rd in reads:
        sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        rd["mom"] = (sh, sc)
        jobs.append(dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=1.0))
    try:
        ctx.set_option("align_blocks_per_cu", 1)
        ctx.set_option("align_lpt", 0)
        in_order = ctx.adaptive_banded_simple_event_align(jobs)
        ctx.set_option("align_lpt", 1)
        longest_first = ctx.adaptive_banded_simple_event_align(jobs)
    finally:
        ctx.set_option("align_blocks_per_cu", 8); ctx.set_option("align_lpt", 1)
    n_ok = 0
    for i, (a, b) in enumerate(zip(in_order, longest_first)):
        assert a.shape == b.shape and np.array_equal(a, b), i
        n_ok += len(a) > 0
    assert 300 < n_ok < 1300
    for i in list(range(0, 1300, 9)) + [7, 700]:
        rd = reads[i]
        want = orc.event_align(mn, orc.scalings(rd["mom"][0], rd["mom"][1], 1.0), rd["events"], rd["ranks"])
        if want is None:
            assert len(longest_first[i]) == 0
            continue
        assert longest_first[i].shape == want.shape and np.array_equal(longest_first[i], want), i


def test_hmm_score_and_align_fuzz(ctx, orc, models):
    nuc, cpg = models["nucleotide"], models["cpg"]
    mn, mc = orc.model(nuc), orc.model(cpg)
    rng = np.random.default_rng(77)
    rd = _random_read(rng, nuc, 2500, 1.6, 1.0)
    ev = rd["events"]; E = len(ev)
    S = orc.scalings(rd["shift"], rd["scale"], rd["var"])
    jobs, want, vjobs, vwant = [], [], [], []
    for t in range(600):
        n = int(rng.choice([6, 7, 15, 16, 17, 18, 23, 24, 25, 31, 32, 33, 64, 65, 129, 300])) if t % 3 == 0 else int(rng.integers(6, 120))
        e = int(rng.integers(2, 600)) if t % 5 == 0 else int(rng.integers(2, 3 * n + 12))
        e = min(e, E - 2)
        e1 = int(rng.integers(0, E - e))
        stride = 1 if rng.random() < 0.5 else -1
        a, b = (e1, e1 + e - 1) if stride == 1 else (e1 + e - 1, e1)
        flags = int(rng.integers(0, 4))
        epb = float(rng.uniform(0.8, 4.0)); bias = float(rng.choice([1.0, 0.9]))
        use_cpg = t % 2 == 0
        ranks = rng.integers(0, 15625 if use_cpg else 4096, n).astype(np.uint32)
        m, mid = (mc, ctx.models["cpg"]) if use_cpg else (mn, ctx.models["nucleotide"])
        job = dict(events=ev, ranks=ranks.astype(np.uint16), e_start=a, e_stop=b, stride=stride, model=mid, scale=rd["scale"], shift=rd["shift"],
                   var=rd["var"], events_per_base=epb, flags=flags, indel_bias=bias)
        jobs.append(job); want.append(orc.hmm_score(m, S, ev, ranks, a, b, stride, epb, bias, flags))
        if t % 4 == 0 and e >= 2:
            vjobs.append(dict(job, flags=0)); vwant.append(orc.hmm_align(m, S, ev, ranks, a, b, stride, epb, bias, 0))
    got = ctx.profile_hmm_score(jobs)
    want = np.array(want, np.float32)
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    assert same.all(), np.flatnonzero(~same)[:5]
    assert np.isfinite(want).sum() > 300
    res = ctx.profile_hmm_align(vjobs)
    for r, w in zip(res, vwant):
        if w is None:
            assert len(r[0]) == 0
            continue
        assert np.array_equal(r[0], w[0]) and np.array_equal(r[1], w[1]) and np.array_equal(r[3], w[3])
        assert np.array_equal(r[2], np.asarray(w[2], np.float64))


def test_record_pipeline_fuzz(ctx, orc, models):
    """24 reads with random substitutions / indels / clips on both strands: the whole device chain from raw signal and BAM
    records (work items on the device) and the eventalign segment chain, against the oracle's restatement of the reference's
    per-read pass (itself pinned to the reference, tests/test_oracle_vs_ref_full.py)"""
    from nanopolish_amd import api
    from nanopolish_amd.pipeline import build_host_batch_records, CallMethylationBatch
    from nanopolish_amd.synth import synth_cigar_read
    from oracle.workloads import call_methylation_record, eventalign_record
    nuc = models["nucleotide"]
    mn, mc = orc.model(nuc), orc.model(models["cpg"])
    g = np.random.default_rng(31337).integers(0, 4, 9000)
    contig = BASES[g].tobytes().decode()
    rng = np.random.default_rng(5)
    recs = []
    for rid in range(24):
        rd = synth_cigar_read(800 + rid, g, nuc, span=int(rng.integers(300, 1500)), p_sub=float(rng.choice([0.0, 0.02, 0.06])),
                              p_ins=float(rng.choice([0.0, 0.02, 0.05])), p_del=float(rng.choice([0.0, 0.02, 0.05])),
                              max_indel=int(rng.integers(1, 12)), soft_clip=(0, int(rng.integers(0, 40))))
        recs.append(dict(seq=rd["seq"], raw=rd["raw"], rc=rd["rc"], pos=rd["pos"], cigar=api.cigar_words(rd["cigar_ops"])))
    hb = build_host_batch_records(models, recs, contig)
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True, jobs_on_device=True)
    batch.step()
    ea = batch.eventalign()
    n_sites = n_rows = 0
    for i, r in enumerate(recs):
        want = call_methylation_record(orc, mn, mc, r["seq"], r["raw"], r["rc"], r["pos"], r["cigar"], contig)
        first, nm, u, m = batch.groups_of(i)
        got = {int(f) + r["pos"]: (float(a), float(b)) for f, a, b in zip(first, u, m) if a == a}
        assert got == {s["start"]: (s["ll_unmeth"], s["ll_meth"]) for s in want["sites"]}, i
        n_sites += len(got)
        if want["n_events"]:
            ev = want["events"]; S = orc.scalings(*want["scalings"])

            def cpu(fwd, rc_s, e1, e2, stride, do_rc):
                return orc.hmm_align(mn, S, ev, orc.sequence_kmer_ranks("nucleotide", fwd, rc_s, K, do_rc), e1, e2, stride, want["epb"])
            rows, _ = eventalign_record(orc, r["seq"], r["rc"], r["pos"], r["cigar"], contig, want["map_start"], cpu)
        else:
            rows = []
        assert ea[i]["status"] == 0
        assert list(zip(ea[i]["ref_position"].tolist(), ea[i]["event_idx"].tolist(), ea[i]["hmm_state"].tolist())) == rows, i
        n_rows += len(rows)
    assert n_sites > 300 and n_rows > 20000
