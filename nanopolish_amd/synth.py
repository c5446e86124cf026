"""Deterministic synthetic R9.4 reads for tests and bench.py.

A read is: L uniform-random bases, K = L-k+1 k-mers; every k-mer emits 0 events w.p. 0.08, else
1 + Bern(.45) + Bern(.15) events; each event mean ~ N(scale*mu_k + shift, (var*sigma_k)^2) from the
r9.4_450bps nucleotide 6-mer table; per-read shift ~ U(-5,5), scale ~ U(.9,1.1), var ~ U(1,1.4), drift 0.
Odd read ids are reverse-strand reads: the reference is revcomp(read sequence).
The generator is numpy-only so that the same arrays feed the CPU oracle and the HIP path.
"""
import numpy as np

BASES = np.frombuffer(b"ACGT", np.uint8)
SEED0 = 0xC0FFEE


def nucleotide_kmer_ranks(codes, k=6):
    """ranks of all k-mers of a base-code array (A0 C1 G2 T3), as Alphabet::kmer_rank orders them"""
    codes = np.asarray(codes, np.int64)
    n = len(codes) - k + 1
    r = np.zeros(n, np.int64)
    for j in range(k):
        r = r * 4 + codes[j:j + n]
    return r.astype(np.uint32)


def revcomp(seq):
    return seq[::-1].translate(str.maketrans("ACGT", "TGCA"))


def synth_read(read_id, model, L=5000, k=6, seed0=SEED0):
    rng = np.random.default_rng(seed0 + int(read_id))
    codes = rng.integers(0, 4, L)
    seq = BASES[codes].tobytes().decode()
    ranks = nucleotide_kmer_ranks(codes, k)
    K = len(ranks)
    n_ev = np.where(rng.random(K) < 0.08, 0, 1 + (rng.random(K) < 0.45) + (rng.random(K) < 0.15)).astype(np.int64)
    shift = rng.uniform(-5, 5)
    scale = rng.uniform(0.9, 1.1)
    var = rng.uniform(1.0, 1.4)
    rk = np.repeat(ranks, n_ev)
    mu = scale * model["level_mean"][rk] + shift
    sd = var * model["level_stdv"][rk]
    events = (mu + sd * rng.standard_normal(len(rk))).astype(np.float32)
    return dict(read_id=int(read_id), seq=seq, codes=codes.astype(np.uint8), ranks=ranks, events=events,
                shift=float(shift), scale=float(scale), var=float(var), rc=bool(read_id & 1))


def synth_batch(read_ids, model, L=5000, k=6, seed0=SEED0):
    """CSR batch: events/event_off, ranks/rank_off, per-read scalars."""
    reads = [synth_read(r, model, L, k, seed0) for r in read_ids]
    event_off = np.zeros(len(reads) + 1, np.int64)
    rank_off = np.zeros(len(reads) + 1, np.int64)
    event_off[1:] = np.cumsum([len(r["events"]) for r in reads])
    rank_off[1:] = np.cumsum([len(r["ranks"]) for r in reads])
    return dict(reads=reads,
                events=np.concatenate([r["events"] for r in reads]),
                event_off=event_off,
                ranks=np.concatenate([r["ranks"] for r in reads]).astype(np.uint32),
                rank_off=rank_off,
                shift=np.array([r["shift"] for r in reads]), scale=np.array([r["scale"] for r in reads]),
                var=np.array([r["var"] for r in reads]), rc=np.array([r["rc"] for r in reads], np.uint8))


def synth_read_from_codes(ref_codes, read_id, model, rc=False, k=6, seed0=SEED0):
    """A read of a GIVEN reference (base codes A0 C1 G2 T3): the read strand is the reference (rc=False) or its reverse
    complement (rc=True); events are emitted along the read strand.  Used for the variants-shaped cases, where many
    reads must cover the same reference."""
    rng = np.random.default_rng(seed0 + 7919 * int(read_id) + 13)
    codes = np.asarray(ref_codes, np.int64)
    if rc:
        codes = 3 - codes[::-1]
    seq = BASES[codes].tobytes().decode()
    ranks = nucleotide_kmer_ranks(codes, k)
    K = len(ranks)
    n_ev = np.where(rng.random(K) < 0.08, 0, 1 + (rng.random(K) < 0.45) + (rng.random(K) < 0.15)).astype(np.int64)
    shift = rng.uniform(-5, 5); scale = rng.uniform(0.9, 1.1); var = rng.uniform(1.0, 1.4)
    rk = np.repeat(ranks, n_ev)
    events = (scale * model["level_mean"][rk] + shift + var * model["level_stdv"][rk] * rng.standard_normal(len(rk))).astype(np.float32)
    return dict(read_id=int(read_id), seq=seq, codes=codes.astype(np.uint8), ranks=ranks, events=events,
                shift=float(shift), scale=float(scale), var=float(var), rc=bool(rc))


ADC_OFFSET = np.float32(10.0)
ADC_UNIT = np.float32(1400.0) / np.float32(8192.0)          # range / digitisation of a MinION channel, in fp32 as the loaders compute it


def adc_quantise(raw):
    """pA samples -> (int16 ADC counts, the pA values those counts convert back to): the signal loaders' conversion
    rawptr[i] = ((float)count + offset) * raw_unit (src/io/nanopolish_fast5_loader.cpp:96-103), all in fp32."""
    adc = np.clip(np.rint(np.asarray(raw, np.float32) / ADC_UNIT - ADC_OFFSET), -32768, 32767).astype(np.int16)
    return adc, ((adc.astype(np.float32) + ADC_OFFSET) * ADC_UNIT).astype(np.float32)


def synth_raw(read_id, model, L=5000, k=6, seed0=SEED0, samples_per_kmer=8.9, noise=1.0, adc=False):
    """Synthetic RAW current trace (pA, float32) of a read, for the event-detection stage:
    the read of synth_read(read_id) dwells on every k-mer for 1 + Poisson(samples_per_kmer - 1) samples (4 kHz sampling
    at 450 bases/s is ~8.9 samples per base) at its scaled model level, with white noise of the k-mer's scaled stdv.
    Values are kept >= 8 pA, like real open-channel-normalised signal (and comfortably inside the range where the
    detector's double-precision prefix sums are exact).  adc=True: also the int16 ADC counts (rd["adc"]), with rd["raw"]
    the pA values they convert to."""
    rd = synth_read(read_id, model, L, k, seed0)
    rng = np.random.default_rng(seed0 + 104729 * (int(read_id) + 1))
    K = len(rd["ranks"])
    dwell = 1 + rng.poisson(samples_per_kmer - 1.0, K)
    rk = np.repeat(rd["ranks"], dwell)
    mu = rd["scale"] * model["level_mean"][rk] + rd["shift"]
    sd = noise * rd["var"] * model["level_stdv"][rk]
    raw = np.maximum(mu + sd * rng.standard_normal(len(rk)), 8.0).astype(np.float32)
    rd = dict(rd); rd["raw"] = raw; rd["dwell"] = dwell
    if adc:          # the trace as a sequencer stores it: int16 counts; "raw" becomes exactly what those counts convert to
        rd["adc"], rd["raw"] = adc_quantise(raw)
    return rd


def synth_raw_from_codes(codes, read_id, model, k=6, seed0=SEED0, samples_per_kmer=8.9, noise=1.0):
    """synth_raw for a GIVEN read sequence (base codes A0 C1 G2 T3, the read's own strand)."""
    rng = np.random.default_rng(seed0 + 15485863 * (int(read_id) + 1))
    codes = np.asarray(codes, np.int64)
    seq = BASES[codes].tobytes().decode()
    ranks = nucleotide_kmer_ranks(codes, k)
    shift = rng.uniform(-5, 5); scale = rng.uniform(0.9, 1.1); var = rng.uniform(1.0, 1.4)
    dwell = 1 + rng.poisson(samples_per_kmer - 1.0, len(ranks))
    rk = np.repeat(ranks, dwell)
    mu = scale * model["level_mean"][rk] + shift
    sd = noise * var * model["level_stdv"][rk]
    raw = np.maximum(mu + sd * rng.standard_normal(len(rk)), 8.0).astype(np.float32)
    return dict(read_id=int(read_id), seq=seq, codes=codes.astype(np.uint8), ranks=ranks, raw=raw, dwell=dwell,
                shift=float(shift), scale=float(scale), var=float(var))


def synth_cigar_read(read_id, contig_codes, model, span=1200, k=6, seed0=SEED0, p_sub=0.02, p_ins=0.015, p_del=0.015,
                     max_indel=4, soft_clip=(0, 12), rc=None, events=False):
    """A read sequenced from a window of a contig with substitutions, insertions, deletions and soft clips, together with
    the BAM record an aligner would report for it: pos, CIGAR (ops as [(char, length)]), SEQ on the reference strand, the
    reverse flag.  The read's own sequence (what the basecaller emitted, and what the signal is generated from) is SEQ
    for a forward read and its reverse complement for a reverse read.
    events=True: the read carries pre-detected EVENTS of its own sequence (synth_read's event model) instead of a raw trace -- the shape of
    the call-methylation bench's reads, here with a genome position."""
    rng = np.random.default_rng(seed0 + 32452843 * (int(read_id) + 1))
    contig_codes = np.asarray(contig_codes, np.int64)
    G = len(contig_codes)
    span = min(span, G)
    pos = int(rng.integers(0, G - span + 1))
    rc = bool(read_id & 1) if rc is None else bool(rc)
    out, ops = [], []

    def push(op, n):
        if n <= 0:
            return
        if ops and ops[-1][0] == op:
            ops[-1][1] += n
        else:
            ops.append([op, n])

    lead = int(rng.integers(soft_clip[0], soft_clip[1] + 1)); tail = int(rng.integers(soft_clip[0], soft_clip[1] + 1))
    out.extend(rng.integers(0, 4, lead).tolist()); push("S", lead)
    i, end = pos, pos + span
    first = True
    while i < end:
        u = rng.random()
        if not first and u < p_ins:
            n = int(rng.integers(1, max_indel + 1)); out.extend(rng.integers(0, 4, n).tolist()); push("I", n)
        elif not first and u < p_ins + p_del and i + max_indel + 1 < end:
            n = int(rng.integers(1, max_indel + 1)); i += n; push("D", n)
        b = int(contig_codes[i])
        if rng.random() < p_sub:
            b = (b + int(rng.integers(1, 4))) & 3
        out.append(b); push("M", 1); i += 1
        first = False
    out.extend(rng.integers(0, 4, tail).tolist()); push("S", tail)
    bam_codes = np.array(out, np.int64)
    read_codes = 3 - bam_codes[::-1] if rc else bam_codes
    rd = synth_read_from_codes(read_codes, read_id, model, rc=False, k=k, seed0=seed0) if events else synth_raw_from_codes(read_codes, read_id, model, k, seed0)
    rd.update(rc=rc, pos=pos, cigar_ops=[(o, n) for o, n in ops], bam_seq=BASES[bam_codes].tobytes().decode(), ref_span=span)
    return rd


def synth_raw_rna(read_id, model, L=1200, k=5, seed0=SEED0, samples_per_kmer=42.0, noise=1.0):
    """Synthetic raw trace of a direct-RNA-like read: the strand passes the pore 3' -> 5' at ~70 bases/s (3 kHz sampling: ~43
    samples per base), so the trace dwells on the k-mers of the basecalled (5' -> 3') sequence from the LAST to the first;
    model: the r9.4_70bps u_to_t_rna 5-mer table (tests/golden/models_r9.4_70bps_rna.npz).  rd["seq"] is the 5' -> 3' sequence
    (T for U, as load_from_raw stores it), rd["ranks"] its k-mer ranks."""
    rd = synth_read(read_id, model, L, k, seed0)
    rng = np.random.default_rng(seed0 + 179424673 * (int(read_id) + 1))
    ranks = rd["ranks"][::-1]
    dwell = np.maximum(4, rng.poisson(samples_per_kmer, len(ranks)))
    rk = np.repeat(ranks, dwell)
    mu = rd["scale"] * model["level_mean"][rk] + rd["shift"]
    sd = noise * rd["var"] * model["level_stdv"][rk]
    raw = np.maximum(mu + sd * rng.standard_normal(len(rk)), 8.0).astype(np.float32)
    rd = dict(rd); rd["raw"] = raw; rd["dwell"] = dwell
    return rd


def synth_cigar_read_fast(read_id, contig_codes, model, span=1200, k=6, seed0=SEED0, p_sub=0.02, p_ins=0.015, p_del=0.015, max_indel=4,
                          soft_clip=(0, 12), events=True):
    """synth_cigar_read's kind of read -- a window of a contig with substitutions, insertions, deletions and soft clips, its BAM record, and
    (events=True) pre-detected events of the read's own sequence -- generated edit by edit instead of base by base (~0.5 ms per 5 kb read):
    what bench.py draws its genome-placed batches from (250 000 reads per rank at N > 1).  Another random stream than synth_cigar_read:
    the two do not produce the same reads for the same id."""
    rng = np.random.default_rng(seed0 + 49979687 * (int(read_id) + 1))
    contig_codes = np.asarray(contig_codes)
    G = len(contig_codes)
    span = min(span, G)
    pos = int(rng.integers(0, G - span + 1))
    rc = bool(read_id & 1)
    ref = contig_codes[pos:pos + span].astype(np.int64)
    sub = rng.random(span) < p_sub
    ref = np.where(sub, (ref + rng.integers(1, 4, span)) & 3, ref)
    # edit sites: strictly inside the window, at least max_indel + 2 reference bases apart, so that operations never touch
    n_ed = int(rng.poisson(span * (p_ins + p_del)))
    gap = max_indel + 2
    sites = np.unique(rng.integers(gap, max(gap + 1, span - 2 * gap), n_ed) // gap * gap) if n_ed else np.zeros(0, np.int64)
    is_ins = rng.random(len(sites)) < p_ins / (p_ins + p_del)
    lens = rng.integers(1, max_indel + 1, len(sites))
    lead = int(rng.integers(soft_clip[0], soft_clip[1] + 1)); tail = int(rng.integers(soft_clip[0], soft_clip[1] + 1))
    parts, ops = [rng.integers(0, 4, lead)], ([["S", lead]] if lead else [])
    cur = 0
    for x, ins, n in zip(sites.tolist(), is_ins.tolist(), lens.tolist()):
        parts.append(ref[cur:x]); ops.append(["M", x - cur]); cur = x
        if ins:
            parts.append(rng.integers(0, 4, n)); ops.append(["I", n])
        else:
            ops.append(["D", n]); cur = x + n
    parts.append(ref[cur:span]); ops.append(["M", span - cur])
    if tail:
        parts.append(rng.integers(0, 4, tail)); ops.append(["S", tail])
    bam_codes = np.concatenate(parts).astype(np.int64)
    read_codes = 3 - bam_codes[::-1] if rc else bam_codes
    rd = synth_read_from_codes(read_codes, read_id, model, rc=False, k=k, seed0=seed0) if events else synth_raw_from_codes(read_codes, read_id, model, k, seed0)
    rd.update(rc=rc, pos=pos, cigar_ops=[(o, n) for o, n in ops if n > 0], ref_span=span)
    return rd
