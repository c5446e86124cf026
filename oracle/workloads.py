"""Reference-side workloads built on the CPU oracle (TEST INFRASTRUCTURE, see oracle/__init__.py):
call-methylation work items and eventalign segments of identity-aligned synthetic reads, and the whole
per-read call-methylation pass (align -> event map -> 2 x profile_hmm_score per CpG group) on the oracle.
"""
import numpy as np

K = 6
HAF_PRE = 1
HAF_POST = 2


def revcomp(seq):
    return seq[::-1].translate(str.maketrans("ACGT", "TGCA"))


def methylation_jobs(orc, read, pairs, min_separation=10, min_flank=10):
    """Work items of calculate_methylation_for_read (src/basemods/nanopolish_basemods.cpp:289-370) for an
    identity-aligned synthetic read.  Returns (events_per_base, list of job dicts)."""
    seq = read["seq"]
    L = len(seq)
    n_kmers = L - K + 1
    rc = read["rc"]
    start, stop, epb = orc.build_base_to_event_map(pairs, n_kmers)
    ref_seq = revcomp(seq) if rc else seq
    aligned_bases = np.stack([np.arange(L), np.arange(L)], 1).astype(np.int32)
    aligned_events = orc.event_alignment_record(aligned_bases, L, K, rc, start)
    first, last, n_motif = orc.scan_motif_groups("cpg", ref_seq, min_separation)
    jobs = []
    for f, l, nm in zip(first, last, n_motif):
        sub_start, sub_end, span = int(f) - min_flank, int(l) + min_flank, int(l) - int(f)
        if sub_start <= min_separation or span > 200:
            continue
        subseq = ref_seq[sub_start:sub_end + 1]
        b = orc.find_by_ref_bounds(aligned_events, sub_start, sub_end) if len(aligned_events) else None
        if b is None or abs(b[1] - b[0]) <= 10:
            continue
        e1, e2 = b
        m_subseq = orc.methylate("cpg", subseq)
        jobs.append(dict(first=int(f), last=int(l), n_motif=int(nm), subseq=subseq, m_subseq=m_subseq,
                         rc_subseq=orc.reverse_complement("cpg", subseq),
                         rc_m_subseq=orc.reverse_complement("cpg", m_subseq),
                         e1=e1, e2=e2, stride=1 if e1 <= e2 else -1, rc=int(rc)))
    return epb, jobs


def eventalign_segments(orc, read, pairs, stride_bp=100):
    """Consecutive ~100-bp segments as align_read_to_ref walks them (src/alignment/nanopolish_eventalign.cpp:668-812),
    simplified to fixed, non-chained windows of a forward identity-aligned read (enough to exercise
    profile_hmm_align at the eventalign problem size: flags = 0)."""
    seq = read["seq"]
    L = len(seq)
    n_kmers = L - K + 1
    start, stop, epb = orc.build_base_to_event_map(pairs, n_kmers)
    segs = []
    for s in range(K, L - stride_bp - K, stride_bp):
        e1 = orc.get_closest_event_to(start, s)
        e2 = orc.get_closest_event_to(start, s + stride_bp - K)
        if e1 < 0 or e2 < 0 or e2 - e1 < 2:
            continue
        segs.append(dict(seq=seq[s:s + stride_bp], e1=e1, e2=e2))
    return epb, segs


def call_methylation_read(orc, mn, mc, read):
    """The reference's per-read pass on the oracle: MoM scalings -> adaptive_banded_simple_event_align ->
    base_to_event_map -> work items -> profile_hmm_score(unmethylated), profile_hmm_score(methylated).
    mn / mc: oracle model handles (nucleotide / cpg)."""
    sh, sc = orc.estimate_scalings_mom(mn, read["ranks"], read["events"])
    pairs = orc.event_align(mn, orc.scalings(sh, sc, 1.0), read["events"], read["ranks"])
    out = dict(mom=(sh, sc), pairs=pairs, epb=0.0, first=np.zeros(0, np.int64), unmeth=np.zeros(0, np.float32),
               meth=np.zeros(0, np.float32), jobs=[])
    if pairs is None or len(pairs) == 0:
        return out
    epb, jobs = methylation_jobs(orc, read, pairs)
    out["epb"] = epb; out["jobs"] = jobs
    if epb > 5.0:          # events-per-base QC, src/nanopolish_squiggle_read.cpp:332
        out["jobs"] = []
        return out
    S = orc.scalings(read["shift"], read["scale"], read["var"])
    u, m = [], []
    for j in jobs:
        ru = orc.sequence_kmer_ranks("cpg", j["subseq"], j["rc_subseq"], K, j["rc"])
        rm = orc.sequence_kmer_ranks("cpg", j["m_subseq"], j["rc_m_subseq"], K, j["rc"])
        u.append(orc.hmm_score(mc, S, read["events"], ru, j["e1"], j["e2"], j["stride"], epb, 1.0, HAF_PRE | HAF_POST))
        m.append(orc.hmm_score(mc, S, read["events"], rm, j["e1"], j["e2"], j["stride"], epb, 1.0, HAF_PRE | HAF_POST))
    out["first"] = np.array([j["first"] for j in jobs], np.int64)
    out["unmeth"] = np.array(u, np.float32); out["meth"] = np.array(m, np.float32)
    return out
