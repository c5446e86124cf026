#!/usr/bin/env python3
"""Generates tests/golden/golden_calls.tsv and golden_frequency{,_split}.tsv: a seeded synthetic call-methylation TSV and
what the REFERENCE's own scripts/calculate_methylation_frequency.py prints for it.  Build container only (needs
/root/reference):   python tests/gen_golden_frequency.py"""
import os
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from nanopolish_amd.output import methylation_tsv_header, format_methylation_tsv  # noqa: E402

GOLD = os.path.join(HERE, "golden")
SCRIPT = "/root/reference/scripts/calculate_methylation_frequency.py"


def synthetic_calls(seed=7, n_reads=40, n_sites=60, with_records=False):
    rng = np.random.default_rng(seed)
    bases = np.array(list("ACGT"))
    # sites on a synthetic contig: groups of 1..3 CGs with 5 bases of context either side
    groups = []
    pos = 100
    for _ in range(n_sites):
        n_motif = int(rng.integers(1, 4))
        gaps = rng.integers(2, 9, n_motif - 1)
        seq = "".join(rng.choice(bases, 5)) + "CG"
        end = pos
        for g in gaps:
            seq += "".join(rng.choice(np.array(list("AT")), int(g) - 2)) + "CG"
            end += int(g)
        seq += "".join(rng.choice(bases, 5))
        groups.append((pos, end, n_motif, seq))
        pos = end + int(rng.integers(15, 60))
    lines = [methylation_tsv_header()]
    records = []
    for r in range(n_reads):
        cover = rng.random(n_sites) < 0.6
        sites = []
        for (s, e, nm, seq), c in zip(groups, cover):
            if not c:
                continue
            llr = float(rng.normal(0, 3.0 * nm))
            # some values exactly on rounding / threshold edges
            if rng.random() < 0.1:
                llr = float(rng.choice([2.0 * nm, -2.0 * nm, 2.0 * nm - 0.005, 1.995 * nm, -1.9949999 * nm, 0.0]))
            u = float(rng.uniform(-300, -100))
            sites.append(dict(chromosome="contig1", start_position=s, end_position=e, n_motif=nm, sequence=seq,
                              ll_unmethylated=[u, 0.0], ll_methylated=[u + llr, 0.0], strands_scored=1))
        lines += format_methylation_tsv(sites, "read_%03d" % r, bool(r & 1))
        records += sites
    return (lines, records) if with_records else lines


def synthetic_genome_calls(seed=11, n_reads=70, min_separation=10):
    """Reads that OVERLAP on two contigs (round 6: the genome-keyed site table).  Every read covers a window of a contig; its groups are the
    window's motif sites chained by gaps <= min_separation (src/basemods/nanopolish_basemods.cpp:306-320) -- so reads that end or start inside a
    cluster report groups with another end / start than the reads that span it, which the frequency script keys separately.
    Returns (TSV lines, records with raw LLRs and contig indices, contigs)."""
    rng = np.random.default_rng(seed)
    contigs = []
    for n in (2600, 1400):
        b = rng.choice(np.array(list("ACGT")), n)
        for i in rng.integers(0, n - 1, n // 9):              # CpG-rich: clusters of several sites are common
            b[i] = "C"; b[i + 1] = "G"
        contigs.append("".join(b))
    contigs[0] = contigs[0][:-2] + "AC"; contigs[1] = "GT" + contigs[1][2:]      # a CG across the contig boundary is NOT a site
    lines = [methylation_tsv_header()]
    records = []
    for r in range(n_reads):
        ci = int(rng.random() < 0.35)
        seq = contigs[ci]
        ln = int(rng.integers(150, 900))
        lo = int(rng.integers(0, len(seq) - ln)); hi = lo + ln
        if r % 7 == 0:                                         # some reads end exactly inside a cluster: on a CG, one base into it, just before it
            cg = [i for i in range(lo + 40, hi - 1) if seq[i:i + 2] == "CG"]
            if cg:
                hi = cg[len(cg) // 2] + int(rng.integers(0, 3))
        sites = [i for i in range(lo, hi - 1) if seq[i:i + 2] == "CG"]
        groups, cur = [], []
        for x in sites:
            if cur and x - cur[-1] > min_separation:
                groups.append(cur); cur = []
            cur.append(x)
        if cur:
            groups.append(cur)
        recs = []
        for g in groups:
            nm = len(g)
            llr = float(rng.normal(0, 3.0 * nm))
            if rng.random() < 0.1:
                llr = float(rng.choice([2.0 * nm, -2.0 * nm, 2.0 * nm - 0.005, 1.995 * nm, -1.9949999 * nm, 0.0]))
            u = float(rng.uniform(-300, -100))
            recs.append(dict(chromosome="contig%d" % (ci + 1), contig=ci, start_position=g[0], end_position=g[-1], n_motif=nm,
                             sequence=seq[max(0, g[0] - 5):g[-1] + 7], ll_unmethylated=[u, 0.0], ll_methylated=[u + llr, 0.0], strands_scored=1))
        lines += format_methylation_tsv(recs, "read_%03d" % r, bool(r & 1))
        records += recs
    return lines, records, contigs


def main():
    lines, _, _ = synthetic_genome_calls()
    path = os.path.join(GOLD, "golden_calls_genome.tsv")
    open(path, "w").write("".join(lines))
    out = subprocess.run([sys.executable, SCRIPT, path], check=True, capture_output=True, text=True).stdout
    open(os.path.join(GOLD, "golden_frequency_genome.tsv"), "w").write(out)
    print("genome", len(out.splitlines()), "lines")
    lines = synthetic_calls()
    path = os.path.join(GOLD, "golden_calls.tsv")
    open(path, "w").write("".join(lines))
    for tag, extra in (("", []), ("_split", ["-s"])):
        out = subprocess.run([sys.executable, SCRIPT] + extra + [path], check=True, capture_output=True, text=True).stdout
        open(os.path.join(GOLD, "golden_frequency%s.tsv" % tag), "w").write(out)
        print(tag or "default", len(out.splitlines()), "lines")


if __name__ == "__main__":
    main()
