"""world_size-2 run of the N>1 path on CPU (gloo): read sharding + the final site-table all-reduce give the same
table as a single process over all reads.  LLRs come from the CPU oracle here (no GPU in this test)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nanopolish_amd.shard import shard_read_ids, reduce_site_table
from nanopolish_amd.sites import site_table

N_READS, READ_LEN = 6, 700


def _rank_table(lo, hi):
    from oracle import Oracle, load_models
    from cases import call_methylation_read, synth_read
    models = load_models(); orc = Oracle()
    mn, mc = orc.model(models["nucleotide"]), orc.model(models["cpg"])
    first, nm, llr = [], [], []
    for rid in range(lo, hi):
        r = call_methylation_read(orc, mn, mc, synth_read(rid, models["nucleotide"], L=READ_LEN))
        first += list(r["first"]); nm += [j["n_motif"] for j in r["jobs"]]
        llr += list(r["meth"].astype(np.float64) - r["unmeth"])
    return site_table(torch, torch.tensor(first, dtype=torch.int64), torch.tensor(nm, dtype=torch.int64),
                      torch.tensor(llr, dtype=torch.float64), READ_LEN)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_read_ids(N_READS, rank, world)
    t = reduce_site_table(_rank_table(lo, hi))
    if rank == 0:
        torch.save(t, out)
    dist.barrier()
    dist.destroy_process_group()


def test_shards_cover_everything_once():
    for n in (0, 1, 7, 100000):
        for w in (1, 2, 3, 8):
            r = [shard_read_ids(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_two_rank_site_reduction_equals_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "table.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    want = _rank_table(0, N_READS)
    assert want[:, 0].sum() > 10
    assert torch.equal(got, want)


# ---- the genome-keyed reduction (round 6): ranks hold reads that OVERLAP on the same contigs ---------------------------------------------
def _genome_rank_table(lo, hi):
    """per-rank table of the golden generator's reads lo..hi-1 (records grouped by read), keyed (contig, start, end), one row per motif site of the
    genome (the bench's layout: np_site_table_genome_indexed_dev) -- sites.site_table_genome(compact=True)"""
    from gen_golden_frequency import synthetic_genome_calls
    from nanopolish_amd.sites import site_table_genome
    lines, recs, contigs = synthetic_genome_calls()
    contig_off = np.concatenate([[0], np.cumsum([len(c) for c in contigs])]).astype(np.int64)
    # the generator emits records read by read: recover the read of every record from the TSV lines (column 4 = read name)
    names = [ln.split("\t")[4] for ln in lines[1:]]
    assert len(names) == len(recs)
    mine = [r for r, nme in zip(recs, names) if lo <= int(nme.split("_")[1]) < hi]
    t = lambda f, dt: torch.tensor([f(r) for r in mine], dtype=dt)
    table, ovf = site_table_genome(torch, t(lambda r: int(contig_off[r["contig"]]) + r["start_position"], torch.int64),
                                   t(lambda r: int(contig_off[r["contig"]]) + r["end_position"], torch.int64), t(lambda r: r["n_motif"], torch.int64),
                                   t(lambda r: (r["ll_methylated"][0] + r["ll_methylated"][1]) - (r["ll_unmethylated"][0] + r["ll_unmethylated"][1]), torch.float64),
                                   "".join(contigs).encode(), contig_off, compact=True)
    assert ovf == 0
    return table, contigs, contig_off


def _gworker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_read_ids(70, rank, world)
    t = reduce_site_table(_genome_rank_table(lo, hi)[0])
    if rank == 0:
        torch.save(t, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_genome_keyed_reduction_equals_the_reference_script(tmp_path):
    """70 reads that overlap on two contigs, sharded over two ranks by read id; each rank's table is keyed by GENOME position
    ((contig, start, end), the two-column-block layout of np_site_table_genome_dev), one all-reduce(sum) -- the N > 1 line's only collective.
    The reduced table equals the single-process one and, key by key, what scripts/calculate_methylation_frequency.py printed for all 70
    reads' calls (tests/golden/golden_frequency_genome.tsv): sites covered by reads of BOTH ranks add up."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from nanopolish_amd.sites import genome_table_rows
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "gtable.pt")
    mp.spawn(_gworker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    want, contigs, contig_off = _genome_rank_table(0, 70)
    assert torch.equal(got, want)
    a, _, _ = _genome_rank_table(0, 35); b, _, _ = _genome_rank_table(35, 70)
    assert int(((a[:, 0] > 0) & (b[:, 0] > 0)).sum()) > 20            # keys both ranks contribute to
    gold = []
    for ln in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_frequency_genome.tsv")).read().splitlines()[1:]:
        f = ln.split("\t")
        base = int(contig_off[int(f[0][len("contig"):]) - 1])
        gold.append((base + int(f[1]), base + int(f[2]), int(f[4]), int(f[5])))
    rows = [(s_, e_, c_, m_) for s_, e_, _, c_, m_ in genome_table_rows(got, "".join(contigs).encode(), contig_off)]
    assert rows == sorted(gold)


# ---- variants (BASELINE config 4): per-variant totals over reads, reads sharded over ranks -----------------------------------
N_VREADS = 6


def _variant_scores(lo, hi):
    """scores [haplotype, read] of a few screened positions for reads lo..hi-1 of one 300-base reference (CPU oracle)"""
    from oracle import Oracle, load_models
    from oracle.workloads import variant_window_items, HAF_PRE, HAF_POST, K
    from nanopolish_amd.synth import synth_read_from_codes
    models = load_models(); orc = Oracle()
    mn = orc.model(models["nucleotide"])
    ref_codes = np.random.default_rng(3).integers(0, 4, 300)
    ref_seq = "".join("ACGT"[c] for c in ref_codes)
    reads_pairs = []
    for rid in range(lo, hi):
        rd = synth_read_from_codes(ref_codes, rid, models["nucleotide"], rc=bool(rid & 1))
        sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        reads_pairs.append((rd, orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"])))
    items = variant_window_items(orc, ref_seq, reads_pairs, range(60, 240, 45))
    rows, base_of = [], []
    for it in items:
        b = len(rows)
        for seq in it["seqs"]:
            row = np.full(hi - lo, np.nan, np.float32)
            for (ri, e1, e2, stride, rc, epb) in it["per_read"]:
                rd = reads_pairs[ri][0]
                ranks = orc.sequence_kmer_ranks("nucleotide", seq, None, K, rc)
                row[ri] = orc.combine_score_set([orc.hmm_score(mn, orc.scalings(rd["shift"], rd["scale"], rd["var"]), rd["events"], ranks,
                                                               e1, e2, stride, epb, 0.9, HAF_PRE | HAF_POST)])
            rows.append(row); base_of.append(b)
    return torch.tensor(np.stack(rows)), torch.tensor(base_of, dtype=torch.int64)


def _vworker(rank, world, port, out):
    from nanopolish_amd.variants import variant_quality, reduce_variant_quality
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_read_ids(N_VREADS, rank, world)
    sc, base_of = _variant_scores(lo, hi)
    q = reduce_variant_quality(variant_quality(torch, sc, base_of))
    if rank == 0:
        torch.save(q, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_variant_qualities_equal_single_process(tmp_path):
    from nanopolish_amd.variants import variant_quality
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "q.pt")
    mp.spawn(_vworker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    sc, base_of = _variant_scores(0, N_VREADS)
    want = variant_quality(torch, sc, base_of)
    assert want.abs().max() > 1.0 and float(want[base_of.unique()].abs().max()) == 0.0          # base haplotypes score 0 against themselves
    assert torch.allclose(got, want, rtol=0, atol=1e-9)                                          # summation order differs across ranks
    # the true base usually wins: most single-base edits of a correct reference have negative quality
    assert (want < 0).sum() > (want > 0).sum()
