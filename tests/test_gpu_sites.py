"""f4 on the device (np_site_table_dev) and profile_hmm_score_set's combination on the device (np_hmm_score_set_combine_dev),
through the C ABI on the MI355X, against the text path (printf "%.2lf" + the frequency script's rules, nanopolish_amd/sites.py
host mirror and nanopolish_amd/output.py) and against np_hmm_score_set_host / the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scores_with_boundaries(rng, n):
    """(unmethylated, methylated) float32 pairs whose differences sit on and around the text round trip's decision points:
    x.xx5 ties (exactly representable eighths), the call threshold 2.0 * n_motif, sign changes, plus random values."""
    u = rng.uniform(-400, -50, n).astype(np.float32)
    d = rng.normal(0, 6, n)
    special = np.array([0.125, -0.125, 0.375, 2.125, -2.125, 1.995, 2.0, -2.0, 2.005, 1.9949999, 3.995, 4.0, 4.005, -3.995, 5.995, 6.0,
                        0.0, 0.004999, -0.004999, 0.005, 7.625, -7.625, 1.875, 2.375, 12.5, -12.5])
    k = len(special)
    d[:k] = special
    d[k:2 * k] = special + rng.choice([-1, 1], k) * 2.0 ** -15         # one float ulp (at |score| in [64,128)) off the boundary
    u[:2 * k] = -64.0 - rng.integers(0, 32, 2 * k)                      # exact in fp32, so that m - u == d exactly
    m = (u.astype(np.float64) + d).astype(np.float32)
    return u, m


def test_site_table_on_the_device_equals_the_text_path(ctx):
    import torch
    from nanopolish_amd.sites import site_table, site_table_dev
    rng = np.random.default_rng(11)
    n, n_pos = 20000, 3000
    u, m = _scores_with_boundaries(rng, n)
    skip = rng.random(n) < 0.1
    u[skip] = np.nan; m[skip] = np.nan
    first = rng.integers(0, n_pos, n).astype(np.int32)
    nm = rng.integers(1, 4, n).astype(np.int32)
    sc = np.stack([u, m], 1).reshape(-1)
    dev = torch.device("cuda:0")
    t = site_table_dev(ctx, torch, torch.from_numpy(sc).to(dev), torch.from_numpy(first).to(dev), torch.from_numpy(nm).to(dev), n_pos)
    ctx.sync()
    llr = torch.from_numpy(m.astype(np.float64) - u.astype(np.float64))
    want = site_table(torch, torch.from_numpy(first.astype(np.int64)), torch.from_numpy(nm.astype(np.int64)), llr, n_pos)
    assert want[:, 0].sum() > 5000
    assert np.array_equal(t.cpu().numpy(), want.numpy())
    # the text path itself: TSV lines in the reference writer's format -> calculate_methylation_frequency.py's rules
    from nanopolish_amd.output import methylation_tsv_header, format_methylation_tsv, calculate_methylation_frequency
    keep = ~skip
    lines = [methylation_tsv_header()]
    for f, k, uu, mm in zip(first[keep], nm[keep], u[keep], m[keep]):
        lines += format_methylation_tsv([dict(chromosome="c", start_position=int(f), end_position=int(f), n_motif=int(k), sequence="CG",
                                              ll_methylated=[float(mm), 0.0], ll_unmethylated=[float(uu), 0.0], strands_scored=1)], "r", False)
    freq = calculate_methylation_frequency(lines)[1:]
    tt = t.cpu().numpy()
    assert len(freq) == int((tt[:, 0] > 0).sum())
    for ln in freq:
        f = ln.split("\t")
        assert tt[int(f[1]), 1] == int(f[4]) and tt[int(f[1]), 2] == int(f[5])


def test_site_table_with_read_offsets(ctx):
    import torch
    from nanopolish_amd.pipeline import JOB_DT
    from nanopolish_amd.sites import site_table_dev
    dev = torch.device("cuda:0")
    sc = np.array([-100, -90, -100, -99.5, np.nan, np.nan, -80, -90], np.float32)       # llr 10, 0.5 (ambiguous), skipped, -10
    first = np.array([5, 5, 7, 5], np.int32); nm = np.array([1, 1, 1, 2], np.int32)
    jobs = np.zeros(8, JOB_DT); jobs["read"] = [0, 0, 0, 0, 1, 1, 1, 1]
    base = np.array([100, 200], np.int64)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    t = site_table_dev(ctx, torch, up(sc).view(torch.float32), up(first).view(torch.int32), up(nm).view(torch.int32), 300,
                       jobs=up(jobs), read_base=up(base).view(torch.int64))
    ctx.sync()
    t = t.cpu().numpy()
    assert tuple(t[105]) == (1, 1, 1) and tuple(t[205]) == (1, 2, 0) and int(t.sum()) == 6


def test_score_set_combination_on_the_device(ctx, orc, models):
    """profile_hmm_score_set with members scored model by model on the device and combined by np_hmm_score_set_combine_dev ==
    np_hmm_score_set_host == the oracle's combine_score_set."""
    import torch
    from cases import K, synth_read, eventalign_segments
    rd = synth_read(7, models["nucleotide"], L=1500)
    mn = orc.model(models["nucleotide"])
    sh, sc_ = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
    pairs = orc.event_align(mn, orc.scalings(sh, sc_, 1.0), rd["events"], rd["ranks"])
    epb, segs = eventalign_segments(orc, rd, pairs)
    sets, member_scores = [], []
    for sg in segs[:12]:
        seqs = [sg["seq"][:30], orc.methylate("cpg", sg["seq"][:30]), sg["seq"][1:31]]
        alph = ["nucleotide", "cpg", "nucleotide"]
        n_members = 2 + (len(sets) % 2)
        jobs = []
        for s, a in list(zip(seqs, alph))[:n_members]:
            jobs.append(dict(events=rd["events"], ranks=orc.sequence_kmer_ranks(a, s, None, K, 0), e_start=sg["e1"], e_stop=sg["e1"] + 40, stride=1,
                             model=ctx.models[a], scale=rd["scale"], shift=rd["shift"], var=rd["var"], events_per_base=epb, flags=0,
                             indel_bias=0.9))
        sets.append(jobs)
    want = ctx.profile_hmm_score_set(sets)
    flat = [j for s in sets for j in s]
    singles = ctx.profile_hmm_score(flat) if len({j["model"] for j in flat}) == 1 else np.concatenate(
        [ctx.profile_hmm_score([j]) for j in flat])
    # members in a shuffled score array + an index, and in set order without one
    off = np.zeros(len(sets) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sets])
    perm = np.random.default_rng(3).permutation(len(flat))
    shuffled = np.zeros(len(flat), np.float32); shuffled[perm] = singles
    dev = torch.device("cuda:0")
    p = lambda t: C.c_void_p(t.data_ptr())
    d_off = torch.from_numpy(off).to(dev)
    for scores, idx in ((singles, None), (shuffled, perm.astype(np.int64))):
        d_sc = torch.from_numpy(scores).to(dev); d_idx = torch.from_numpy(idx).to(dev) if idx is not None else None
        d_out = torch.zeros(len(sets), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ctx._chk(ctx.L.np_hmm_score_set_combine_dev(ctx.h, None, len(sets), p(d_off), p(d_idx) if d_idx is not None else None,
                                                            p(d_sc), p(d_out)), "np_hmm_score_set_combine_dev")
        ctx.sync()
        assert np.array_equal(d_out.cpu().numpy(), want)
    for q, s in enumerate(sets):
        assert want[q] == np.float32(orc.combine_score_set(singles[off[q]:off[q + 1]]))


def _genome_batch(ctx, models, n_reads, G, span, seed=0x9E0, tile=1, first_id=0):
    from nanopolish_amd import api
    from nanopolish_amd.synth import synth_cigar_read_fast, BASES
    from nanopolish_amd.pipeline import build_host_batch_records, tile_host_batch, CallMethylationBatch
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, G).astype(np.uint8)
    for i in rng.integers(0, G - 1, G // 12):                  # CpG-rich: clusters of several sites, so that reads end inside clusters
        genome[i] = 1; genome[i + 1] = 2
    contig = BASES[genome].tobytes().decode()
    recs = []
    for rid in range(first_id, first_id + n_reads):
        r = synth_cigar_read_fast(rid, genome, models["nucleotide"], span=span)
        recs.append(dict(seq=r["seq"], events=r["events"], shift=r["shift"], scale=r["scale"], var=r["var"], rc=int(r["rc"]), pos=int(r["pos"]),
                         cigar=api.cigar_words(r["cigar_ops"])))
    hb = build_host_batch_records(models, recs, contig, with_jobs=False)
    b = CallMethylationBatch(ctx, tile_host_batch(hb, tile), "cuda:0", calibrate=True, from_raw=False, jobs_on_device=True, map_stop=False)
    return contig, recs, hb, b


def test_genome_keyed_site_table_on_the_device(ctx, orc, models):
    """Round 6.  Reads with a PLACE on a genome (BAM-style records: position, CIGAR with substitutions / indels / clips, both
    strands), from pre-detected events, through the step the bench times -- work items by CIGAR on the device, alignment, recalibration, scoring --
    then np_site_table_genome_dev.  (a) every read's scored sites (genome start / end, n_motif, both log-likelihoods) equal the oracle's
    calculate_methylation_for_read restatement on the same record, bit for bit; (b) the device table equals the host mirror fed with the ORACLE's
    sites -- the mirror itself is pinned to the reference's frequency script by tests/test_output.py -- including the keys of reads that stop
    inside a cluster (columns 3-5); (c) a tiled batch (every read twice) gives exactly twice the table; nothing overflows."""
    import torch
    from oracle.workloads import call_methylation_record
    from nanopolish_amd.sites import site_table_genome, genome_table_rows
    G = 24000
    contig, recs, hb, b = _genome_batch(ctx, models, 36, G, 1500)
    b.step()
    table, ovf = b.genome_site_table()
    ctx.sync()
    mn, mc = orc.model(models["nucleotide"]), orc.model(models["cpg"])
    S, E, NM, LLR = [], [], [], []
    n_sites = 0
    for i, r in enumerate(recs):
        want = call_methylation_record(orc, mn, mc, r["seq"], None, r["rc"], r["pos"], r["cigar"], contig, events=r["events"])
        first, nm, u, m = b.groups_of(i)
        keep = np.isfinite(u)
        got = [(int(f) + r["pos"], int(k), float(uu), float(mm)) for f, k, uu, mm in zip(first[keep], nm[keep], u[keep], m[keep])]
        exp = [(s["start"], s["n_motif"], s["ll_unmeth"], s["ll_meth"]) for s in want["sites"]]
        assert got == exp, "read %d" % i
        n_sites += len(exp)
        for s in want["sites"]:
            S.append(s["start"]); E.append(s["end"]); NM.append(s["n_motif"]); LLR.append(np.float64(np.float32(s["ll_meth"])) - np.float64(np.float32(s["ll_unmeth"])))
    assert n_sites > 1500
    contig_off = np.array([0, G], np.int64)
    want_t, want_ovf = site_table_genome(torch, torch.tensor(S), torch.tensor(E), torch.tensor(NM), torch.tensor(LLR, dtype=torch.float64),
                                         contig.encode(), contig_off)
    assert int(ovf.cpu()[0]) == 0 and want_ovf == 0
    t = table.cpu().numpy()
    assert np.array_equal(t, want_t.numpy())
    assert (t[:, 0] > 1).sum() > 50                      # sites seen by several overlapping reads
    # (no key of the second form here: a group cut by its read's end lies within min_flank of that end, and calculate_methylation_for_read's
    #  window rules drop it -- basemods.cpp:334, alignment_db.cpp:697-708.  test_genome_table_kernel_on_cut_groups feeds the kernel such groups.)
    rows = genome_table_rows(table, contig.encode(), contig_off)
    assert sum(r[2] for r in rows) == int(t[:, 0].sum() + t[:, 3].sum())
    del b
    contig2, recs2, hb2, b2 = _genome_batch(ctx, models, 36, G, 1500, tile=2)
    b2.step()
    t2, ovf2 = b2.genome_site_table()
    ctx.sync()
    assert np.array_equal(t2.cpu().numpy(), 2 * t) and int(ovf2.cpu()[0]) == 0


def test_genome_table_kernel_on_cut_groups(ctx):
    """np_site_table_genome_dev on the golden generator's calls (tests/gen_golden_frequency.py:synthetic_genome_calls): reads on TWO contigs whose
    groups are cut by the read's start or end -- keys (start', cluster end) and (cluster start, end'), the second form in columns 3-5 -- and
    LLRs on the text round trip's decision points.  The device table equals the host mirror and, key by key, the REFERENCE SCRIPT's output
    for the same calls (tests/golden/golden_frequency_genome.tsv)."""
    import os
    import torch
    from gen_golden_frequency import synthetic_genome_calls
    from nanopolish_amd.pipeline import JOB_DT
    from nanopolish_amd.sites import site_table_genome, site_table_genome_dev, genome_table_rows
    lines, recs, contigs = synthetic_genome_calls()
    contig_off = np.concatenate([[0], np.cumsum([len(c) for c in contigs])]).astype(np.int64)
    genome = "".join(contigs).encode()
    dev = torch.device("cuda:0")
    # every record is its own "read" whose segment starts 7 bases before the group (read_base + segment-relative positions, as the builder writes them)
    base = np.array([contig_off[r["contig"]] + r["start_position"] - 7 for r in recs], np.int64)
    first = np.full(len(recs), 7, np.int32)
    last = np.array([7 + r["end_position"] - r["start_position"] for r in recs], np.int32)
    nm = np.array([r["n_motif"] for r in recs], np.int32)
    # device scores are floats: unmethylated 0, methylated = the TSV's own two-decimal text as a float -- its "%.2lf" round trip is that text again,
    # so the device decides on exactly the numbers the reference script parsed
    text = np.array([float(ln.split("\t")[5]) for ln in lines[1:]], np.float64)
    assert len(text) == len(recs)
    uf = np.zeros(len(recs), np.float32); mf = text.astype(np.float32)
    llr = mf.astype(np.float64) - uf.astype(np.float64)
    assert all("%.2f" % a == "%.2f" % b for a, b in zip(llr, text))
    jobs = np.zeros(2 * len(recs), JOB_DT); jobs["read"] = np.repeat(np.arange(len(recs), dtype=np.uint32), 2)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    sc = np.stack([uf, mf], 1).reshape(-1)
    table, ovf = site_table_genome_dev(ctx, torch, up(sc).view(torch.float32), up(first).view(torch.int32), up(last).view(torch.int32), up(nm).view(torch.int32),
                                       up(jobs), up(base).view(torch.int64), up(np.frombuffer(genome, np.uint8)), up(contig_off).view(torch.int64))
    ctx.sync()
    t = lambda a, dt: torch.tensor(a, dtype=dt)
    want, want_ovf = site_table_genome(torch, t(base + 7, torch.int64), t(base + last, torch.int64), t(nm, torch.int64), t(llr, torch.float64), genome, contig_off)
    got = table.cpu().numpy()
    assert np.array_equal(got, want.numpy()) and int(ovf.cpu()[0]) == want_ovf == 0
    assert (got[:, 3] > 0).sum() >= 3 and (got[:, 0] > 0).sum() > 100
    gold = []
    for ln in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_frequency_genome.tsv")).read().splitlines()[1:]:
        f = ln.split("\t")
        b = int(contig_off[int(f[0][len("contig"):]) - 1])
        gold.append((b + int(f[1]), b + int(f[2]), int(f[4]), int(f[5])))
    assert [(s_, e_, c_, m_) for s_, e_, _, c_, m_ in genome_table_rows(table, genome, contig_off)] == sorted(gold)
    # ---- one row per motif SITE (np_genome_site_index_dev + np_site_table_genome_indexed_dev): the same keys, the rows of the sites only
    from nanopolish_amd.sites import genome_site_index_dev, motif_sites
    d_genome = up(np.frombuffer(genome, np.uint8)); d_off = up(contig_off).view(torch.int64)
    mask, rank, n_sites = genome_site_index_dev(ctx, torch, d_genome, d_off)
    hit = motif_sites(genome, contig_off)
    assert n_sites == int(hit.sum())
    bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")[:len(hit)].astype(bool)
    assert np.array_equal(bits, hit)                                                      # (a CG across the contig boundary is not a site)
    words = np.add.reduceat(hit.astype(np.int64), np.arange(0, len(hit), 64))
    assert np.array_equal(rank.cpu().numpy().view(np.uint32), np.concatenate([[0], np.cumsum(words)]).astype(np.uint32))
    compact, ovf_c = site_table_genome_dev(ctx, torch, up(sc).view(torch.float32), up(first).view(torch.int32), up(last).view(torch.int32),
                                           up(nm).view(torch.int32), up(jobs), up(base).view(torch.int64), d_genome, d_off, index=(mask, rank, n_sites))
    ctx.sync()
    assert compact.shape == (n_sites, 6) and int(ovf_c.cpu()[0]) == 0
    assert np.array_equal(compact.cpu().numpy(), got[hit]) and int(got[~hit].sum()) == 0
    want_c, _ = site_table_genome(torch, t(base + 7, torch.int64), t(base + last, torch.int64), t(nm, torch.int64), t(llr, torch.float64), genome, contig_off,
                                  compact=True)
    assert np.array_equal(compact.cpu().numpy(), want_c.numpy())
    assert genome_table_rows(compact, genome, contig_off) == genome_table_rows(table, genome, contig_off)


@pytest.mark.parametrize("n_pos", [1, 63, 64, 65, 2048 * 64 - 1, 2048 * 64 * 3 + 77, 140_000_011])
def test_genome_site_index_sizes(ctx, n_pos):
    """np_genome_site_index_dev across the scan's seams: one word, one chunk of 2 048 words, several chunks, more chunks than the second pass's
    1 024 threads (140 Mb = 1 069 chunks), several contigs with sites cut by their boundaries, for every alphabet."""
    import torch
    from nanopolish_amd.sites import genome_site_index_dev, motif_sites
    rng = np.random.default_rng(n_pos)
    g = rng.choice(np.frombuffer(b"ACGT", np.uint8), n_pos)
    cuts = np.unique(np.concatenate([[0, n_pos], rng.integers(0, n_pos + 1, 5)])).astype(np.int64)
    dev = torch.device("cuda:0")
    d_g = torch.from_numpy(g).to(dev); d_off = torch.from_numpy(cuts).to(dev)
    for alphabet in (("cpg", "gpc", "dam", "dcm") if n_pos < 10_000_000 else ("cpg",)):
        mask, rank, n_sites = genome_site_index_dev(ctx, torch, d_g, d_off, alphabet=alphabet)
        hit = motif_sites(g.tobytes(), cuts, alphabet)
        assert n_sites == int(hit.sum())
        bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")[:n_pos].astype(bool)
        assert np.array_equal(bits, hit)
        words = np.add.reduceat(hit.astype(np.int64), np.arange(0, n_pos, 64))
        assert np.array_equal(rank.cpu().numpy().view(np.uint32), np.concatenate([[0], np.cumsum(words)]).astype(np.uint32))
