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


def record_reference_segment(contig, pos, cigar_ops_or_words):
    """What calculate_methylation_for_read fetches for a record (src/basemods/nanopolish_basemods.cpp:259-270):
    contig[pos .. bam_endpos] INCLUSIVE (faidx_fetch_seq's end is inclusive), clipped to the contig."""
    ref_len = 0
    for w in cigar_ops_or_words:
        op, n = (("MIDNSHP=X".index(w[0]), int(w[1])) if isinstance(w, (tuple, list)) else (int(w) & 0xf, int(w) >> 4))
        if op in (0, 2, 3, 7, 8):
            ref_len += n
    endpos = pos + (ref_len if ref_len > 0 else 1)
    return contig[pos:min(endpos + 1, len(contig))]


def methylation_jobs_record(orc, rc, read_length, cigar, pos, ref_seq, map_start, alphabet="cpg", min_separation=10, min_flank=10):
    """Work items of calculate_methylation_for_read (src/basemods/nanopolish_basemods.cpp:289-370) for a read aligned by a
    BAM record: SequenceAlignmentRecord (CIGAR walk) -> EventAlignmentRecord -> motif groups -> _find_by_ref_bounds."""
    aligned_bases = orc.cigar_aligned_bases(cigar, pos)
    if aligned_bases is None:
        raise ValueError("spliced alignment")
    aligned_events = orc.event_alignment_record(aligned_bases, read_length, K, rc, map_start)
    first, last, n_motif = orc.scan_motif_groups(alphabet, ref_seq, min_separation)
    jobs = []
    for f, l, nm in zip(first, last, n_motif):
        sub_start, sub_end, span = int(f) - min_flank, int(l) + min_flank, int(l) - int(f)
        if sub_start <= min_separation or span > 200:
            continue
        subseq = ref_seq[sub_start:sub_end + 1]
        b = orc.find_by_ref_bounds(aligned_events, sub_start + pos, sub_end + pos) if len(aligned_events) else None
        if b is None or abs(b[1] - b[0]) <= 10:
            continue
        e1, e2 = b
        m_subseq = orc.methylate(alphabet, subseq)
        jobs.append(dict(first=int(f) + pos, last=int(l) + pos, n_motif=int(nm), subseq=subseq, m_subseq=m_subseq,
                         rc_subseq=orc.reverse_complement(alphabet, subseq),
                         rc_m_subseq=orc.reverse_complement(alphabet, m_subseq),
                         e1=e1, e2=e2, stride=1 if e1 <= e2 else -1, rc=int(rc),
                         sequence=ref_seq[int(f) - K + 1:int(l) + K]))
    return jobs


def call_methylation_record(orc, mn, m_meth, read_seq, raw, rc, pos, cigar, contig, alphabet="cpg", events=None):
    """The reference's whole per-read pass restated on the oracle, from raw signal and a BAM record:
    SquiggleRead::load_from_raw (src/nanopolish_squiggle_read.cpp:189-336: detect_events, MoM scalings, event alignment,
    base_to_event_map, recalibrate_model, QC gates) then calculate_methylation_for_read.
    mn / m_meth: oracle model handles (nucleotide / the methylation alphabet's).  Returns a dict with the read-level state
    (events, scalings, events_per_base, event map) and the scored sites in ascending start position.
    events: pre-detected event means (raw is then ignored): the same pass from the event table on, as a read loaded from an events file."""
    L = len(read_seq)
    codes = np.frombuffer(read_seq.encode(), np.uint8)
    lut = np.zeros(256, np.int64); lut[ord("C")] = 1; lut[ord("G")] = 2; lut[ord("T")] = 3
    c = lut[codes]
    n_kmers = L - K + 1
    ranks = np.zeros(n_kmers, np.int64)
    for j in range(K):                                           # Alphabet::kmer_rank of every read k-mer (nucleotide)
        ranks = ranks * 4 + c[j:j + n_kmers]
    ranks = ranks.astype(np.uint32)
    if events is None:
        ev = orc.detect_events(raw)
        events = ev["mean"] if isinstance(ev, dict) else ev[0]
    else:
        events = np.ascontiguousarray(events, np.float32)
    out = dict(n_events=0, events=events, scalings=None, epb=0.0, map_start=None, map_stop=None, sites=[], jobs=[])
    sh, sc = orc.estimate_scalings_mom(mn, ranks, events)
    pairs = orc.event_align(mn, orc.scalings(sh, sc, 1.0), events, ranks)
    if pairs is None or len(pairs) == 0:                          # failed alignment: events cleared (:324-329)
        out["scalings"] = (sh, sc, 1.0)
        return out
    start, stop, epb = orc.build_base_to_event_map(pairs, n_kmers)
    out["map_start"], out["map_stop"], out["epb"] = start, stop, epb
    cal = orc.recalibrate(mn, events, ranks, start, stop)
    out["scalings"] = cal if cal is not None else (sh, sc, 1.0)
    if cal is None or cal[2] > 2.5:                               # not recalibrated / MIN_CALIBRATION_VAR (:320-323)
        return out
    if epb > 5.0:                                                 # events-per-base QC (:332)
        return out
    out["n_events"] = len(events)
    ref_seq = record_reference_segment(contig, pos, cigar)
    jobs = methylation_jobs_record(orc, rc, L, cigar, pos, ref_seq, start, alphabet)
    S = orc.scalings(*cal)
    for j in jobs:
        ru = orc.sequence_kmer_ranks(alphabet, j["subseq"], j["rc_subseq"], K, j["rc"])
        rm = orc.sequence_kmer_ranks(alphabet, j["m_subseq"], j["rc_m_subseq"], K, j["rc"])
        u = orc.hmm_score(m_meth, S, events, ru, j["e1"], j["e2"], j["stride"], epb, 1.0, HAF_PRE | HAF_POST)
        m = orc.hmm_score(m_meth, S, events, rm, j["e1"], j["e2"], j["stride"], epb, 1.0, HAF_PRE | HAF_POST)
        out["sites"].append(dict(start=j["first"], end=j["last"], n_motif=j["n_motif"], ll_unmeth=float(np.float32(u)),
                                 ll_meth=float(np.float32(m)), sequence=j["sequence"]))
    out["jobs"] = jobs
    return out


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


def call_methylation_read(orc, mn, mc, read, calibrate=False):
    """The reference's per-read pass on the oracle: MoM scalings -> adaptive_banded_simple_event_align ->
    base_to_event_map -> [calibrate: recalibrate_model, squiggle_read.cpp:304-323] -> work items ->
    profile_hmm_score(unmethylated), profile_hmm_score(methylated).
    mn / mc: oracle model handles (nucleotide / cpg).  calibrate=False scores with the read's given scalings."""
    sh, sc = orc.estimate_scalings_mom(mn, read["ranks"], read["events"])
    pairs = orc.event_align(mn, orc.scalings(sh, sc, 1.0), read["events"], read["ranks"])
    out = dict(mom=(sh, sc), pairs=pairs, epb=0.0, first=np.zeros(0, np.int64), unmeth=np.zeros(0, np.float32),
               meth=np.zeros(0, np.float32), jobs=[], calibrated=False, scalings=None)
    if pairs is None or len(pairs) == 0:
        return out
    epb, jobs = methylation_jobs(orc, read, pairs)
    out["epb"] = epb; out["jobs"] = jobs
    cal = (read["shift"], read["scale"], read["var"])
    if calibrate:
        start, stop, _ = orc.build_base_to_event_map(pairs, len(read["ranks"]))
        cal = orc.recalibrate(mn, read["events"], read["ranks"], start, stop)
        if cal is None or cal[2] > 2.5:       # not recalibrated / MIN_CALIBRATION_VAR: events cleared (:320-323)
            out["scalings"] = cal
            out["jobs"] = []
            return out
    out["calibrated"] = True; out["scalings"] = cal
    if epb > 5.0:          # events-per-base QC, src/nanopolish_squiggle_read.cpp:332
        out["jobs"] = []
        return out
    S = orc.scalings(*cal)
    u, m = [], []
    for j in jobs:
        ru = orc.sequence_kmer_ranks("cpg", j["subseq"], j["rc_subseq"], K, j["rc"])
        rm = orc.sequence_kmer_ranks("cpg", j["m_subseq"], j["rc_m_subseq"], K, j["rc"])
        u.append(orc.hmm_score(mc, S, read["events"], ru, j["e1"], j["e2"], j["stride"], epb, 1.0, HAF_PRE | HAF_POST))
        m.append(orc.hmm_score(mc, S, read["events"], rm, j["e1"], j["e2"], j["stride"], epb, 1.0, HAF_PRE | HAF_POST))
    out["first"] = np.array([j["first"] for j in jobs], np.int64)
    out["unmeth"] = np.array(u, np.float32); out["meth"] = np.array(m, np.float32)
    return out


def eventalign_read(orc, read, pairs, hmm_align_fn, align_stride=100, output_stride=50):
    """align_read_to_ref's segment chain (src/alignment/nanopolish_eventalign.cpp:655-823) for an identity-aligned
    synthetic read, parameterised by the profile_hmm_align implementation:
        hmm_align_fn(fwd_subseq, rc_subseq, e_start, e_stop, stride, rc) -> (event_idx, kmer_idx, l_fm, state) or None
    Returns the emitted (ref_position, event_idx, state) triples and the number of profile_hmm_align calls."""
    seq = read["seq"]
    L = len(seq)
    n_kmers = L - K + 1
    do_base_rc = bool(read["rc"])
    ref_seq = revcomp(seq) if do_base_rc else seq
    rc_ref_seq = revcomp(ref_seq)
    start, stop, epb = orc.build_base_to_event_map(pairs, n_kmers)
    max_kmer_idx = L - K
    aligned = [(p, p) for p in range(L) if p <= max_kmer_idx]          # trim_aligned_pairs_to_kmer (:167-177)
    flip = lambda i: L - i - K                                          # flip_k_strand, squiggle_read.h:229-233
    closest = lambda k_idx: orc.get_closest_event_to(start, k_idx)

    def get_end_pair(ref_pos_max, pair_idx):                            # :196-205
        while pair_idx < len(aligned):
            if aligned[pair_idx][0] > ref_pos_max:
                return pair_idx - 1
            pair_idx += 1
        return len(aligned) - 1

    read_kidx_start, read_kidx_end = aligned[0][1], aligned[-1][1]
    if do_base_rc:
        read_kidx_start, read_kidx_end = flip(read_kidx_start), flip(read_kidx_end)
    first_event, last_event = closest(read_kidx_start), closest(read_kidx_end)
    forward = first_event < last_event
    curr_start_event, curr_start_ref, curr_pair_idx = first_event, aligned[0][0], 0
    out, n_calls = [], 0
    while (forward and curr_start_event < last_event) or (not forward and curr_start_event > last_event):
        end_pair_idx = get_end_pair(curr_start_ref + align_stride, curr_pair_idx)
        curr_end_ref, curr_end_read = aligned[end_pair_idx]
        if do_base_rc:
            curr_end_read = flip(curr_end_read)
        s, l = curr_start_ref, curr_end_ref - curr_start_ref + 1
        fwd_subseq = ref_seq[s:s + l]
        rc_subseq = rc_ref_seq[len(ref_seq) - s - l:len(ref_seq) - s]
        if len(fwd_subseq) < 2 * K:
            break
        e_start, e_stop = curr_start_event, closest(curr_end_read)
        if abs(e_start - e_stop) < 2:
            break
        stride = 1 if e_start < e_stop else -1
        res = hmm_align_fn(fwd_subseq, rc_subseq, e_start, e_stop, stride, do_base_rc)
        n_calls += 1
        if res is None:
            break
        ev, km, lf, st = res
        last_section = end_pair_idx == len(aligned) - 1
        num_output = 0
        last_event_output = last_ref_kmer_output = 0
        for i in range(len(ev)):
            if not (num_output < output_stride or last_section):
                break
            if chr(st[i]) != 'K' and int(ev[i]) != curr_start_event:
                out.append((curr_start_ref + int(km[i]), int(ev[i]), int(st[i])))
                last_event_output = int(ev[i]); last_ref_kmer_output = curr_start_ref + int(km[i])
                num_output += 1
        curr_start_event, curr_start_ref = last_event_output, last_ref_kmer_output
        curr_pair_idx = get_end_pair(curr_start_ref, curr_pair_idx)
        if num_output == 0:
            break
    return out, n_calls, epb


def eventalign_record(orc, read_seq, rc, pos, cigar, contig, map_start, hmm_align_fn, align_stride=100, output_stride=50):
    """align_read_to_ref (src/alignment/nanopolish_eventalign.cpp:612-826) for a read aligned by a BAM record (CIGAR words,
    0-based pos, reverse flag), parameterised by the profile_hmm_align implementation like eventalign_read:
        hmm_align_fn(fwd_subseq, rc_subseq, e_start, e_stop, stride, rc) -> (event_idx, kmer_idx, l_fm, state) or None
    Returns the emitted (ref_position, event_idx, state) triples and the number of profile_hmm_align calls."""
    L = len(read_seq)
    ref_seq = record_reference_segment(contig, pos, cigar)
    rc_ref_seq = revcomp(ref_seq)
    ab = orc.cigar_aligned_bases(cigar, pos)
    max_kmer_idx = L - K
    aligned = [(int(r), int(q)) for r, q in ab]
    while aligned and aligned[-1][1] > max_kmer_idx:                    # trim_aligned_pairs_to_kmer (:167-177)
        aligned.pop()
    if not aligned:
        return [], 0
    flip = lambda i: L - i - K
    closest = lambda k_idx: orc.get_closest_event_to(map_start, k_idx)

    def get_end_pair(ref_pos_max, pair_idx):                            # :196-205
        while pair_idx < len(aligned):
            if aligned[pair_idx][0] > ref_pos_max:
                return pair_idx - 1
            pair_idx += 1
        return len(aligned) - 1

    ks, ke = aligned[0][1], aligned[-1][1]
    if rc:
        ks, ke = flip(ks), flip(ke)
    first_event, last_event = closest(ks), closest(ke)
    forward = first_event < last_event
    curr_start_event, curr_start_ref, curr_pair_idx = first_event, aligned[0][0], 0
    out, n_calls = [], 0
    while (forward and curr_start_event < last_event) or (not forward and curr_start_event > last_event):
        end_pair_idx = get_end_pair(curr_start_ref + align_stride, curr_pair_idx)
        curr_end_ref, curr_end_read = aligned[end_pair_idx]
        if rc:
            curr_end_read = flip(curr_end_read)
        s, l = curr_start_ref - pos, curr_end_ref - curr_start_ref + 1
        fwd_subseq = ref_seq[s:s + l]
        rc_subseq = rc_ref_seq[len(ref_seq) - s - l:len(ref_seq) - s]
        if len(fwd_subseq) < 2 * K:
            break
        e_start, e_stop = curr_start_event, closest(curr_end_read)
        if abs(e_start - e_stop) < 2:
            break
        stride = 1 if e_start < e_stop else -1
        res = hmm_align_fn(fwd_subseq, rc_subseq, e_start, e_stop, stride, bool(rc))
        n_calls += 1
        if res is None:
            break
        ev, km, lf, st = res
        last_section = end_pair_idx == len(aligned) - 1
        num_output = 0
        last_event_output = last_ref_kmer_output = 0
        for i in range(len(ev)):
            if not (num_output < output_stride or last_section):
                break
            if chr(st[i]) != 'K' and int(ev[i]) != curr_start_event:
                out.append((curr_start_ref + int(km[i]), int(ev[i]), int(st[i])))
                last_event_output = int(ev[i]); last_ref_kmer_output = curr_start_ref + int(km[i])
                num_output += 1
        curr_start_event, curr_start_ref = last_event_output, last_ref_kmer_output
        curr_pair_idx = get_end_pair(curr_start_ref, curr_pair_idx)
        if num_output == 0:
            break
    return out, n_calls


def variant_window_items(orc, ref_seq, reads_pairs, positions, flank=10):
    """Work items of generate_candidate_single_base_edits + score_variant_thresholded
    (src/nanopolish_call_variants.cpp:288-361, src/common/nanopolish_variant.cpp:765-799) for identity-aligned reads of one
    reference: per position a 22-bp window, the base haplotype and its single-base substitutions / insertions / deletion;
    per read the event bounds of the window (AlignmentDB::get_event_subsequences, alignment_db.cpp:172-221).
    reads_pairs: list of (read dict, aligner pairs).  Returns a list of dicts(pos, seqs=[base, variants...],
    per_read=[(read index, e1, e2, stride, rc)])."""
    L = len(ref_seq)
    recs = []
    for rd, pairs in reads_pairs:
        n_kmers = L - K + 1
        start, stop, epb = orc.build_base_to_event_map(pairs, n_kmers)
        ab = np.stack([np.arange(L), np.arange(L)], 1).astype(np.int32)
        ae = orc.event_alignment_record(ab, L, K, rd["rc"], start)
        stride = 1 if (len(ae) and ae[0, 1] < ae[-1, 1]) else -1
        recs.append((ae, stride, epb))
    items = []
    for i in positions:
        cs, ce = i - flank, i + 1 + flank
        base = ref_seq[cs:ce + 1]
        seqs = [base]
        o = i - cs
        ref_b = ref_seq[i]
        for b in "ACGT":
            if b != ref_b:
                seqs.append(base[:o] + b + base[o + 1:])                 # substitution
            if b != ref_b:
                seqs.append(base[:o + 1] + b + base[o + 1:])             # insertion ref -> ref+b (not ref+ref)
        if ref_seq[i - 1] != ref_seq[i]:
            seqs.append(base[:o] + base[o + 1:])                         # deletion of base i (ref[i-1:i+1] -> ref[i-1])
        per_read = []
        for ri, (ae, stride, epb) in enumerate(recs):
            b = orc.find_by_ref_bounds(ae, cs, ce) if len(ae) else None
            if b is None:
                continue
            e1, e2 = b
            if abs(e1 - e2) / abs(ce - cs) < 20.0:                       # MAX_EVENT_TO_BP_RATIO heuristic (:206-216)
                per_read.append((ri, e1, e2, stride, int(reads_pairs[ri][0]["rc"]), epb))
        items.append(dict(pos=i, seqs=seqs, per_read=per_read))
    return items
