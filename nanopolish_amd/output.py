"""Output side of call-methylation, host code mirroring the reference's writers:

* `methylation_tsv_header()` / `format_methylation_tsv()`: OutputHandles::write_site_header
  (src/basemods/nanopolish_basemods.h:25-30) and write_methylation_results_as_tsv
  (src/nanopolish_call_methylation.cpp:531-550): one line per scored CpG group, log-likelihoods printed with %.2lf.
* `calculate_methylation_frequency()`: scripts/calculate_methylation_frequency.py (per-site aggregation of the TSV, default
  call threshold 2.0, optional --split-groups), returning the lines that script prints.
* `site_records_from_scores()`: ScoredSite assembly of calculate_methylation_for_read (src/basemods/nanopolish_basemods.cpp:
  384-413) for one strand of one read: start/end position, n_motif and the k-mer-padded group sequence.

* `modbam_tags()`: the `Mm` / `Ml` tag payloads create_modbam_record attaches to a read's BAM record
  (src/basemods/nanopolish_basemods.cpp:50-177): per-call probability codes and the run-length "C+m?" delta string.
  Writing the BAM record itself is htslib's job and is not rebuilt.

The device-resident `sites.site_table` computes the
same per-site counts as `calculate_methylation_frequency` without going through text; tests/test_output.py checks the
two against each other and against the output of the reference's own script (tests/golden/golden_frequency.tsv).
"""

TSV_COLUMNS = ["chromosome", "strand", "start", "end", "read_name", "log_lik_ratio", "log_lik_methylated",
               "log_lik_unmethylated", "num_calling_strands", "num_motifs", "sequence"]


def methylation_tsv_header():
    return "\t".join(TSV_COLUMNS) + "\n"


def format_methylation_tsv(sites, read_name, is_rev):
    """sites: iterable of dicts(chromosome, start_position, end_position, n_motif, sequence, ll_methylated[2],
    ll_unmethylated[2], strands_scored), in ascending start_position order (the reference iterates a std::map)."""
    orient = "-" if is_rev else "+"
    out = []
    for ss in sorted(sites, key=lambda s: s["start_position"]):
        sum_ll_m = float(ss["ll_methylated"][0]) + float(ss["ll_methylated"][1])
        sum_ll_u = float(ss["ll_unmethylated"][0]) + float(ss["ll_unmethylated"][1])
        diff = sum_ll_m - sum_ll_u
        out.append("%s\t%s\t%d\t%d\t%s\t%.2f\t%.2f\t%.2f\t%d\t%d\t%s\n" % (
            ss["chromosome"], orient, ss["start_position"], ss["end_position"], read_name, diff, sum_ll_m, sum_ll_u,
            ss["strands_scored"], ss["n_motif"], ss["sequence"]))
    return out


def site_records_from_scores(contig, ref_seq, ref_start_pos, first_site, last_site, n_motif, unmeth, meth, k=6, strand_idx=0,
                             region_start=-1, region_end=-1):
    """One ScoredSite per scored group (NaN scores = group skipped upstream).  first_site/last_site: motif positions in
    ref_seq coordinates (what np_scan_motif_groups returns)."""
    out = []
    for f, l, nm, u, m in zip(first_site, last_site, n_motif, unmeth, meth):
        if u != u or m != m:
            continue
        start_position = int(f) + ref_start_pos
        end_position = int(l) + ref_start_pos
        if (region_start != -1 and start_position < region_start) or (region_end != -1 and end_position >= region_end):
            continue
        ll_u = [0.0, 0.0]; ll_m = [0.0, 0.0]
        ll_u[strand_idx] = float(u); ll_m[strand_idx] = float(m)
        out.append(dict(chromosome=contig, start_position=start_position, end_position=end_position, n_motif=int(nm),
                        sequence=ref_seq[int(f) - k + 1:int(l) + k], ll_unmethylated=ll_u, ll_methylated=ll_m, strands_scored=1))
    return out


def calculate_methylation_frequency(tsv_lines, call_threshold=2.0, split_groups=False):
    """tsv_lines: the call-methylation TSV incl. its header line.  Returns the script's output lines (header first)."""
    lines = [ln.rstrip("\n") for ln in tsv_lines if ln.strip()]
    cols = lines[0].split("\t")
    sites = {}

    def update(key, num_called, is_methylated, sequence):
        st = sites.get(key)
        if st is None:
            st = sites[key] = dict(num_reads=0, called_sites=0, called_sites_methylated=0, group_size=num_called, sequence=sequence)
        st["num_reads"] += 1
        st["called_sites"] += num_called
        if is_methylated > 0:
            st["called_sites_methylated"] += num_called

    for ln in lines[1:]:
        rec = dict(zip(cols, ln.split("\t")))
        num_sites = int(rec["num_motifs"])
        llr = float(rec["log_lik_ratio"])
        if abs(llr) < call_threshold * num_sites:
            continue
        sequence = rec["sequence"]
        is_methylated = llr > 0
        if split_groups and num_sites > 1:
            c = str(rec["chromosome"]); s = int(rec["start"])
            cg_pos = sequence.find("CG")
            first_cg_pos = cg_pos
            while cg_pos != -1:
                key = (c, s + cg_pos - first_cg_pos, s + cg_pos - first_cg_pos)
                update(key, 1, is_methylated, "split-group")
                cg_pos = sequence.find("CG", cg_pos + 1)
        else:
            update((str(rec["chromosome"]), int(rec["start"]), int(rec["end"])), num_sites, is_methylated, sequence)
    out = ["\t".join(["chromosome", "start", "end", "num_motifs_in_group", "called_sites", "called_sites_methylated",
                      "methylated_frequency", "group_sequence"])]
    for key in sorted(sites.keys()):
        st = sites[key]
        if st["called_sites"] > 0:
            f = float(st["called_sites_methylated"]) / st["called_sites"]
            out.append("%s\t%s\t%s\t%d\t%d\t%d\t%.3f\t%s" % (key[0], key[1], key[2], st["group_size"], st["called_sites"],
                                                             st["called_sites_methylated"], f, st["sequence"]))
    return out


def modbam_tags(sites, cigar, pos, bam_seq, is_rev, alphabet="cpg"):
    """Mm / Ml tag payloads of create_modbam_record (src/basemods/nanopolish_basemods.cpp:107-177) for one read.
    sites: dicts(start_position, sequence, ll_methylated[2], ll_unmethylated[2]) as site_records_from_scores returns them
    (any order; the reference iterates a std::map keyed by start position); cigar / pos / bam_seq / is_rev: the read's BAM
    record (CIGAR words, 0-based position, SEQ as stored -- on the reference strand -- and the reverse flag).
    Returns (mm_string, ml_codes: list of ints 0..255)."""
    import math
    from . import api
    if alphabet != "cpg":
        raise ValueError("create_modbam_record asserts the cpg alphabet (basemods.cpp:137)")
    unmodified_symbol = "C"                                   # get_modification_symbols: the base under the M of "MG"
    # calculate_call_vectors (:50-81)
    call_reference_positions, call_probabilities = [], []
    for call in sorted(sites, key=lambda c: c["start_position"]):
        seq = call["sequence"]
        m_seq = api.methylate(alphabet, seq)
        flank_offset = m_seq.find("M")
        assert flank_offset != -1
        llm, llu = float(call["ll_methylated"][0]), float(call["ll_unmethylated"][0])
        em, eu = math.exp(llm), math.exp(llu)
        den = em + eu
        # the reference computes (int)(p * 255) on a double; 0/0 (both likelihoods underflow) converts to INT_MIN there,
        # whose low byte is 0
        p = em / den if den != 0.0 else float("nan")
        code = min(255, int(p * 255)) & 0xFF if p == p else 0
        for j, ch in enumerate(m_seq):
            if ch == "M":
                call_reference_positions.append(call["start_position"] + j - flank_offset)
                call_probabilities.append(code)
    # reference position -> index into the ORIGINAL read sequence (:122-127)
    aligned = api.cigar_aligned_bases(cigar, pos)
    original_sequence = bam_seq if not is_rev else api.reverse_complement("nucleotide", bam_seq)
    n = len(original_sequence)
    ref_to_read = {int(rp): (int(qp) if not is_rev else n - int(qp) - 1) for rp, qp in aligned}
    strand_offset = 1 if is_rev else 0                        # on the opposite strand the read base of interest pairs with the G
    idx, probs = [], []
    for rp, pr in zip(call_reference_positions, call_probabilities):
        ri = ref_to_read.get(rp + strand_offset)
        if ri is not None and original_sequence[ri] == unmodified_symbol:
            idx.append(ri); probs.append(pr)
    if is_rev:
        idx.reverse(); probs.reverse()
    # generate_mm_tag (:83-105)
    parts = [unmodified_symbol + "+m?"]
    count_start = 0
    for i in idx:
        parts.append(",%d" % original_sequence[count_start:i].count(unmodified_symbol))
        count_start = i + 1
    return "".join(parts) + ";", probs
