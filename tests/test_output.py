"""f4: TSV record formatting and per-site frequency aggregation vs the reference's own script
(tests/golden/golden_frequency*.tsv, produced by scripts/calculate_methylation_frequency.py in tests/gen_golden_frequency.py),
and the device-side site table (nanopolish_amd/sites.py, here on CPU tensors) vs the text path."""
import ctypes as C
import os
import numpy as np

from nanopolish_amd.output import (methylation_tsv_header, format_methylation_tsv, calculate_methylation_frequency,
                                   site_records_from_scores)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_frequency_matches_reference_script_output():
    calls = open(os.path.join(GOLD, "golden_calls.tsv")).readlines()
    for tag, split in (("", False), ("_split", True)):
        want = open(os.path.join(GOLD, "golden_frequency%s.tsv" % tag)).read().splitlines()
        assert calculate_methylation_frequency(calls, split_groups=split) == want


def test_tsv_generator_is_stable_and_formats_like_printf():
    from gen_golden_frequency import synthetic_calls
    assert "".join(synthetic_calls()) == open(os.path.join(GOLD, "golden_calls.tsv")).read()
    assert methylation_tsv_header().split("\t")[5] == "log_lik_ratio"
    libc = C.CDLL(None)
    libc.snprintf.restype = C.c_int
    rng = np.random.default_rng(2)
    vals = np.concatenate([rng.normal(0, 50, 3000), np.arange(-200, 200) * 0.005, [0.125, 0.135, 2.675, -0.005, 1e-9, -1e-9]])
    buf = C.create_string_buffer(64)
    for v in vals:
        libc.snprintf(buf, C.c_size_t(64), b"%.2lf", C.c_double(float(v)))
        assert buf.value.decode() == "%.2f" % float(v)


def test_site_records_follow_scored_site_rules():
    ref = "TTACGTTTTTACGACGTTTTTTTTTTACGTTT"
    recs = site_records_from_scores("chr", ref, 1000, [3, 11, 27], [3, 14, 27], [1, 2, 1], [-10.0, float("nan"), -30.0], [-8.0, -1.0, -33.5], k=6)
    assert [r["start_position"] for r in recs] == [1003, 1027] and recs[1]["end_position"] == 1027
    assert recs[0]["sequence"] == ref[3 - 5:3 + 6] and recs[0]["n_motif"] == 1
    line = format_methylation_tsv(recs, "r1", False)[0].rstrip("\n").split("\t")
    assert line[:4] == ["chr", "+", "1003", "1003"] and line[5:8] == ["2.00", "-8.00", "-10.00"]


def test_device_site_table_equals_text_aggregation():
    import torch
    from gen_golden_frequency import synthetic_calls
    from nanopolish_amd.sites import site_table
    lines, recs = synthetic_calls(with_records=True)
    first = torch.tensor([r["start_position"] for r in recs], dtype=torch.int64)
    nm = torch.tensor([r["n_motif"] for r in recs], dtype=torch.int64)
    llr = torch.tensor([(r["ll_methylated"][0] + r["ll_methylated"][1]) - (r["ll_unmethylated"][0] + r["ll_unmethylated"][1]) for r in recs],
                       dtype=torch.float64)
    table = site_table(torch, first, nm, llr, 20000).numpy()
    freq = calculate_methylation_frequency(lines)[1:]
    seen = 0
    for ln in freq:
        f = ln.split("\t")
        s, called, meth = int(f[1]), int(f[4]), int(f[5])
        assert table[s, 1] == called and table[s, 2] == meth
        seen += 1
    assert seen > 30 and int((table[:, 1] > 0).sum()) == seen


def _genome_case():
    import torch
    from gen_golden_frequency import synthetic_genome_calls
    lines, recs, contigs = synthetic_genome_calls()
    contig_off = np.concatenate([[0], np.cumsum([len(c) for c in contigs])]).astype(np.int64)
    genome = "".join(contigs).encode()
    start = torch.tensor([contig_off[r["contig"]] + r["start_position"] for r in recs], dtype=torch.int64)
    end = torch.tensor([contig_off[r["contig"]] + r["end_position"] for r in recs], dtype=torch.int64)
    nm = torch.tensor([r["n_motif"] for r in recs], dtype=torch.int64)
    llr = torch.tensor([(r["ll_methylated"][0] + r["ll_methylated"][1]) - (r["ll_unmethylated"][0] + r["ll_unmethylated"][1]) for r in recs],
                       dtype=torch.float64)
    return lines, recs, contigs, contig_off, genome, start, end, nm, llr


def _golden_genome_rows(contig_off):
    want = []
    for ln in open(os.path.join(GOLD, "golden_frequency_genome.tsv")).read().splitlines()[1:]:
        f = ln.split("\t")
        base = int(contig_off[int(f[0][len("contig"):]) - 1])
        want.append((base + int(f[1]), base + int(f[2]), int(f[4]), int(f[5])))
    return sorted(want)


def test_genome_keyed_site_table_equals_reference_script_on_overlapping_reads():
    """Round 6: reads that overlap on a genome, keyed (contig, start, end) as scripts/calculate_methylation_frequency.py:16-23
    keys them.  Reads that stop or start inside a cluster of sites report groups with another end / start: keys of their own in the script,
    columns 3-5 / a different row of the two-table layout here.  The host mirror's rows equal the reference script's output (golden), key by
    key, over two contigs (a CG across the contig boundary is not a site); nothing overflows."""
    import torch
    from nanopolish_amd.sites import site_table_genome, genome_table_rows
    lines, recs, contigs, contig_off, genome, start, end, nm, llr = _genome_case()
    assert "".join(lines) == open(os.path.join(GOLD, "golden_calls_genome.tsv")).read()
    assert calculate_methylation_frequency(lines) == open(os.path.join(GOLD, "golden_frequency_genome.tsv")).read().splitlines()
    table, overflow = site_table_genome(torch, start, end, nm, llr, genome, contig_off)
    assert overflow == 0
    got = [(s, e, called, meth) for s, e, _, called, meth in genome_table_rows(table, genome, contig_off)]
    want = _golden_genome_rows(contig_off)
    assert got == want and len(want) > 100
    t = table.numpy()
    assert (t[:, 3] > 0).sum() >= 3 and (t[:, 0] > 0).sum() > 100          # both key forms occur
    starts = {}
    for s, e, _, _ in want:
        starts.setdefault(s, set()).add(e)
    assert any(len(v) > 1 for v in starts.values())                       # one start, several ends: what a start-keyed table would merge
    # one row per motif SITE (the layout of np_site_table_genome_indexed_dev): the same keys from a table of the sites' rows only
    from nanopolish_amd.sites import motif_sites
    hit = motif_sites(genome, contig_off)
    compact, ovf_c = site_table_genome(torch, start, end, nm, llr, genome, contig_off, compact=True)
    assert ovf_c == 0 and compact.shape[0] == int(hit.sum()) < len(genome) // 4
    assert np.array_equal(compact.numpy(), t[hit]) and int(t[~hit].sum()) == 0
    assert genome_table_rows(compact, genome, contig_off) == genome_table_rows(table, genome, contig_off)
