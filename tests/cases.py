"""Shared seeded test cases: synthetic reads, call-methylation work items, eventalign segments.

Everything here is TEST-side: it may use the oracle (oracle/) freely.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nanopolish_amd.synth import synth_read, revcomp  # noqa: E402

K = 6
HAF_PRE = 1
HAF_POST = 2


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
