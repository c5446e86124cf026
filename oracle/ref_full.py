"""ctypes binding of oracle/_ref/libnp_ref_full.so (TEST INFRASTRUCTURE, see oracle/__init__.py): the reference's own
read-level code -- SquiggleRead::load_from_raw, EventAlignmentRecord, calculate_methylation_for_read,
create_modbam_record, align_read_to_ref -- compiled in place from /root/reference (oracle/Makefile target `full`,
veneer oracle/ref_full_harness.cpp).  Exists only where /root/reference does; it pins the portable restatement and
generates tests/golden/golden_reflevel.npz (tests/gen_golden_reflevel.py)."""
import ctypes as C
import os
import subprocess
import time
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_FULL = os.path.join(_HERE, "_ref", "libnp_ref_full.so")
# NP_REF_BATCH_LIB: another build of the batch configuration (the sanitizer build of the shims, tests/test_gpu_sanitizers.py)
_BATCH = os.environ.get("NP_REF_BATCH_LIB") or os.path.join(_HERE, "_ref", "libnp_ref_full_batch.so")

_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)

CIGAR_OPS = "MIDNSHP=X"


def have_full():
    return os.path.exists(_FULL)


def build_full():
    subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "ref", "full"])


def cigar_words(ops):
    """[(op_char, length), ...] -> uint32 BAM cigar words (length << 4 | op)"""
    return np.array([(int(l) << 4) | CIGAR_OPS.index(o) for o, l in ops], np.uint32)


def _p(a, t):
    return a.ctypes.data_as(t)


class FullRead:
    """A SquiggleRead built by the reference from raw samples (SquiggleRead(sequence, Fast5Data, flags))."""

    def __init__(self, lib, name, sequence, raw, sample_rate=4000.0, rna=False):
        self.L = lib
        raw = np.ascontiguousarray(raw, np.float32)
        self.sequence = sequence
        make = lib.npfull_read_create_rna if rna else lib.npfull_read_create       # rna: load_from_raw's direct-RNA branch
        make.restype = C.c_void_p
        make.argtypes = [C.c_char_p, C.c_char_p, _f32p, C.c_size_t, C.c_double]
        self.h = make(name.encode(), sequence.encode(), _p(raw, _f32p), len(raw), float(sample_rate))
        n_ev, mp = C.c_int(), C.c_int()
        sh, sc, va, epb = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        lib.npfull_read_summary(self.h, C.byref(n_ev), C.byref(sh), C.byref(sc), C.byref(va), C.byref(epb), C.byref(mp))
        self.n_events, self.map_size = n_ev.value, mp.value
        self.shift, self.scale, self.var, self.events_per_base = sh.value, sc.value, va.value, epb.value

    def close(self):
        if self.h:
            self.L.npfull_read_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def events(self):
        out = np.zeros(self.n_events, np.float32)
        if self.n_events:
            self.L.npfull_read_events(self.h, _p(out, _f32p))
        return out

    def event_map(self):
        s, e = np.zeros(self.map_size, np.int32), np.zeros(self.map_size, np.int32)
        if self.map_size:
            self.L.npfull_read_event_map(self.h, _p(s, _i32p), _p(e, _i32p))
        return s, e

    def closest_event(self, k_idx):
        return int(self.L.npfull_closest_event(self.h, int(k_idx)))

    def event_alignment_record(self, is_rev, pos, cigar, bam_seq):
        cig = np.ascontiguousarray(cigar, np.uint32)
        cap = len(bam_seq) + 8
        rp, ev = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        rc, stride = C.c_int(), C.c_int()
        n = self.L.npfull_event_alignment_record(self.h, int(is_rev), int(pos), _p(cig, _u32p), len(cig), bam_seq.encode(), cap,
                                                 _p(rp, _i32p), _p(ev, _i32p), C.byref(rc), C.byref(stride))
        return np.stack([rp[:n], ev[:n]], 1), rc.value, stride.value

    def call_methylation(self, is_rev, pos, cigar, bam_seq, contig_seq, methylation_type="cpg", modbam=True, cap=4096):
        """calculate_methylation_for_read: dict of per-site arrays (+ the Mm / Ml tag payloads of create_modbam_record)"""
        cig = np.ascontiguousarray(cigar, np.uint32)
        st, en, nm = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        lu, lm = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
        seqs = C.create_string_buffer(cap * 256)
        mm_cap = 16 * cap + 64
        mm = C.create_string_buffer(mm_cap)
        ml = np.zeros(4 * cap, np.uint8)
        n_ml = C.c_int(-1)
        want_mod = modbam and methylation_type == "cpg"
        n = self.L.npfull_call_methylation(self.h, int(is_rev), int(pos), _p(cig, _u32p), len(cig), bam_seq.encode(),
                                           contig_seq.encode(), methylation_type.encode(), cap, _p(st, _i32p), _p(en, _i32p),
                                           _p(nm, _i32p), _p(lu, _f64p), _p(lm, _f64p), seqs,
                                           mm if want_mod else None, mm_cap, _p(ml, _u8p), len(ml), C.byref(n_ml))
        assert n <= cap
        raw = seqs.raw
        out = dict(start=st[:n].copy(), end=en[:n].copy(), n_motif=nm[:n].copy(), ll_unmeth=lu[:n].copy(), ll_meth=lm[:n].copy(),
                   sequence=[raw[i * 256:(i + 1) * 256].split(b"\0", 1)[0].decode() for i in range(n)])
        if want_mod:
            out["Mm"] = mm.value.decode()
            out["Ml"] = ml[:n_ml.value].copy()
        return out

    def eventalign(self, is_rev, pos, cigar, bam_seq, contig_seq, cap=None):
        cig = np.ascontiguousarray(cigar, np.uint32)
        cap = cap or (4 * max(self.n_events, 1) + 64)
        rp, ev = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        st = C.create_string_buffer(cap)
        rk, mk = C.create_string_buffer(cap * 8), C.create_string_buffer(cap * 8)
        n = self.L.npfull_eventalign(self.h, int(is_rev), int(pos), _p(cig, _u32p), len(cig), bam_seq.encode(), contig_seq.encode(),
                                     cap, _p(rp, _i32p), _p(ev, _i32p), st, rk, mk)
        assert n <= cap
        ks = lambda b: [b.raw[i * 8:(i + 1) * 8].split(b"\0", 1)[0].decode() for i in range(n)]
        return dict(ref_position=rp[:n].copy(), event_idx=ev[:n].copy(), hmm_state=np.frombuffer(st.raw[:n], np.uint8).copy(),
                    ref_kmer=ks(rk), model_kmer=ks(mk))


    def eventalign_tsv(self, is_rev, pos, cigar, bam_seq, contig_seq, read_idx=0):
        """align_read_to_ref printed by the reference's emit_event_alignment_tsv (default options)"""
        cig = np.ascontiguousarray(cigar, np.uint32)
        cap = 160 * (4 * max(self.n_events, 1) + 64)
        buf = C.create_string_buffer(cap)
        n = self.L.npfull_eventalign_tsv(self.h, int(is_rev), int(pos), _p(cig, _u32p), len(cig), bam_seq.encode(), contig_seq.encode(),
                                         int(read_idx), buf, cap)
        assert n < cap
        return buf.value.decode()


class FullRef:
    def __init__(self, batch=False):
        """batch=False: the unmodified reference (libnp_ref_full.so, CPU).  batch=True: the reference with the product's bindings linked
        in place of its hot-path translation units (libnp_ref_full_batch.so, needs a GPU): the same entry points then run through
        np_dropin.cpp, and the np_* batched bindings are reachable (mode=1 of score_variants / score_variant_group)."""
        if not have_full():
            raise RuntimeError("oracle/_ref/libnp_ref_full.so is not built (needs /root/reference; `make -C oracle full`)")
        L = C.CDLL(_BATCH if batch else _FULL)
        L.npfull_read_create.restype = C.c_void_p
        L.npfull_read_create.argtypes = [C.c_char_p, C.c_char_p, _f32p, C.c_size_t, C.c_double]
        L.npfull_read_destroy.argtypes = [C.c_void_p]
        L.npfull_read_summary.argtypes = [C.c_void_p, C.POINTER(C.c_int), _f64p, _f64p, _f64p, _f64p, C.POINTER(C.c_int)]
        L.npfull_read_events.argtypes = [C.c_void_p, _f32p]
        L.npfull_read_event_map.argtypes = [C.c_void_p, _i32p, _i32p]
        L.npfull_closest_event.argtypes = [C.c_void_p, C.c_int]
        L.npfull_aligned_bases.argtypes = [C.c_int, C.c_int, _u32p, C.c_int, C.c_char_p, C.c_int, _i32p, _i32p]
        L.npfull_event_alignment_record.argtypes = [C.c_void_p, C.c_int, C.c_int, _u32p, C.c_int, C.c_char_p, C.c_int, _i32p, _i32p,
                                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.npfull_find_by_ref_bounds.argtypes = [_i32p, _i32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.npfull_call_methylation.argtypes = [C.c_void_p, C.c_int, C.c_int, _u32p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int,
                                              _i32p, _i32p, _i32p, _f64p, _f64p, C.c_char_p, C.c_char_p, C.c_int, _u8p, C.c_int,
                                              C.POINTER(C.c_int)]
        L.npfull_eventalign.argtypes = [C.c_void_p, C.c_int, C.c_int, _u32p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, _i32p, _i32p,
                                        C.c_char_p, C.c_char_p, C.c_char_p]
        L.npfull_eventalign_tsv.argtypes = [C.c_void_p, C.c_int, C.c_int, _u32p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.npfull_many_identity.argtypes = [C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int64), _f32p, C.POINTER(C.c_int64), _u8p, C.c_double,
                                           C.c_int, _i32p]
        self.L = L

    def _variant_args(self, reads, records):
        n = len(reads)
        hs = (C.c_void_p * n)(*[r.h for r in reads])
        cig = np.concatenate([np.ascontiguousarray(r["cigar"], np.uint32) for r in records])
        cig_off = np.zeros(n + 1, np.int64); cig_off[1:] = np.cumsum([len(r["cigar"]) for r in records])
        is_rev = np.array([int(r["rc"]) for r in records], np.int32); pos = np.array([int(r["pos"]) for r in records], np.int32)
        bseqs = (C.c_char_p * n)(*[r["bam_seq"].encode() for r in records])
        return n, hs, is_rev, pos, cig, cig_off, bseqs

    def score_variants(self, mode, reads, records, contig_seq, positions, flank=10, score_threshold=1000000, methylation_types="",
                       indel_bias=0.9, cap=1 << 16):
        """The screening loop of generate_candidate_single_base_edits over `positions` (npfull_score_variants): mode 0 = the
        reference's score_variant_thresholded per candidate (one OpenMP thread), mode 1 = np_score_variants_thresholded, all windows
        in one device batch (batch=True only).  reads: FullRead objects of THIS library; records: dicts(rc, pos, cigar, bam_seq).
        Returns (quality per candidate, window index per candidate, number of profile_hmm_score_set evaluations)."""
        n, hs, is_rev, pos, cig, cig_off, bseqs = self._variant_args(reads, records)
        P = np.ascontiguousarray(positions, np.int32)
        q = np.zeros(cap, np.float64); w = np.zeros(cap, np.int32); sets = C.c_int64(0)
        self.L.npfull_score_variants.restype = C.c_int
        m = self.L.npfull_score_variants(int(mode), n, hs, _p(is_rev, _i32p), _p(pos, _i32p), _p(cig, _u32p), _p(cig_off, C.POINTER(C.c_int64)),
                                         bseqs, contig_seq.encode(), len(P), _p(P, _i32p), int(flank), int(score_threshold),
                                         methylation_types.encode(), C.c_double(indel_bias), cap, _p(q, _f64p), _p(w, _i32p), C.byref(sets))
        assert 0 <= m <= cap
        return q[:m].copy(), w[:m].copy(), int(sets.value)

    def score_variant_group(self, mode, reads, records, contig_seq, positions, flank=10, max_haplotypes=1000, methylation_types="",
                            indel_bias=0.8, cap=1 << 20):
        """score_variant_group for one group of substitutions at `positions`: array [combination, input read] of
        get_combination_read_score (mode as score_variants)."""
        n, hs, is_rev, pos, cig, cig_off, bseqs = self._variant_args(reads, records)
        P = np.ascontiguousarray(positions, np.int32)
        sc = np.zeros(cap, np.float64); ni = C.c_int(0)
        self.L.npfull_score_variant_group.restype = C.c_int
        nc = self.L.npfull_score_variant_group(int(mode), n, hs, _p(is_rev, _i32p), _p(pos, _i32p), _p(cig, _u32p),
                                               _p(cig_off, C.POINTER(C.c_int64)), bseqs, contig_seq.encode(), len(P), _p(P, _i32p), int(flank),
                                               int(max_haplotypes), methylation_types.encode(), C.c_double(indel_bias), cap, _p(sc, _f64p),
                                               C.byref(ni))
        assert nc >= 0 and nc * ni.value <= cap
        return sc[:nc * ni.value].reshape(nc, ni.value).copy()

    def many_identity(self, mode, seqs, raws, rcs, n_threads=0, sample_rate=4000.0):
        """OpenMP-over-reads timing driver (see npfull_many_identity): mode 0 eventalign, 1 call-methylation.
        Returns (rows per read, seconds)."""
        import time
        seq_off = np.zeros(len(seqs) + 1, np.int64); seq_off[1:] = np.cumsum([len(q) for q in seqs])
        raw_off = np.zeros(len(raws) + 1, np.int64); raw_off[1:] = np.cumsum([len(r) for r in raws])
        raw = np.ascontiguousarray(np.concatenate(raws), np.float32)
        rc = np.ascontiguousarray(np.array(rcs, np.uint8))
        rows = np.zeros(len(seqs), np.int32)
        t0 = time.perf_counter()
        self.L.npfull_many_identity(int(mode), len(seqs), "".join(seqs).encode(), _p(seq_off, C.POINTER(C.c_int64)), _p(raw, _f32p),
                                    _p(raw_off, C.POINTER(C.c_int64)), _p(rc, _u8p), float(sample_rate), int(n_threads), _p(rows, _i32p))
        return rows, time.perf_counter() - t0

    def many_records(self, recs, contig_seq, n_threads, sample_rate=4000.0):
        """SquiggleRead from raw + align_read_to_ref for every record (dicts: seq, raw, rc, pos, cigar, bam_seq) against one contig, OpenMP
        over records: (row counts, hashes of every record's rows -- see rows_hash, seconds)."""
        n = len(recs)
        seq_off = np.zeros(n + 1, np.int64); seq_off[1:] = np.cumsum([len(r["seq"]) for r in recs])
        bam_off = np.zeros(n + 1, np.int64); bam_off[1:] = np.cumsum([len(r["bam_seq"]) for r in recs])
        raw_off = np.zeros(n + 1, np.int64); raw_off[1:] = np.cumsum([len(r["raw"]) for r in recs])
        cig_off = np.zeros(n + 1, np.int64); cig_off[1:] = np.cumsum([len(r["cigar"]) for r in recs])
        raw = np.concatenate([np.ascontiguousarray(r["raw"], np.float32) for r in recs])
        cig = np.concatenate([np.ascontiguousarray(r["cigar"], np.uint32) for r in recs])
        is_rev = np.array([int(r["rc"]) for r in recs], np.int32); pos = np.array([int(r["pos"]) for r in recs], np.int32)
        rows = np.zeros(n, np.int32); hsh = np.zeros(n, np.uint64)
        i64 = C.POINTER(C.c_int64)
        self.L.npfull_many_records.argtypes = [C.c_int, C.c_char_p, i64, _f32p, i64, _i32p, _i32p, _u32p, i64, C.c_char_p, i64, C.c_char_p, C.c_double,
                                               C.c_int, _i32p, C.POINTER(C.c_uint64)]
        t0 = time.perf_counter()
        self.L.npfull_many_records(n, "".join(r["seq"] for r in recs).encode(), _p(seq_off, i64), _p(raw, _f32p), _p(raw_off, i64), _p(is_rev, _i32p),
                                   _p(pos, _i32p), _p(cig, _u32p), _p(cig_off, i64), "".join(r["bam_seq"] for r in recs).encode(), _p(bam_off, i64),
                                   contig_seq.encode(), float(sample_rate), int(n_threads), _p(rows, _i32p), _p(hsh, C.POINTER(C.c_uint64)))
        return rows, hsh, time.perf_counter() - t0

    def read(self, name, sequence, raw, sample_rate=4000.0, rna=False):
        return FullRead(self.L, name, sequence, raw, sample_rate, rna)

    def aligned_bases(self, is_rev, pos, cigar, bam_seq):
        cig = np.ascontiguousarray(cigar, np.uint32)
        cap = len(bam_seq) + 8
        rp, qp = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = self.L.npfull_aligned_bases(int(is_rev), int(pos), _p(cig, _u32p), len(cig), bam_seq.encode(), cap, _p(rp, _i32p), _p(qp, _i32p))
        return np.stack([rp[:n], qp[:n]], 1)

    def find_by_ref_bounds(self, pairs, ref_start, ref_stop):
        pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
        rp, qp = np.ascontiguousarray(pairs[:, 0]), np.ascontiguousarray(pairs[:, 1])
        r1, r2 = C.c_int(), C.c_int()
        ok = self.L.npfull_find_by_ref_bounds(_p(rp, _i32p), _p(qp, _i32p), len(rp), int(ref_start), int(ref_stop), C.byref(r1), C.byref(r2))
        return (r1.value, r2.value) if ok else None




def have_batch():
    return os.path.exists(_BATCH)


def call_methylation_batch(records, contig_seq, methylation_type="cpg", cap=65536):
    """The reference build with the product's BATCHED binding linked in (`make -C oracle batch`: np_dropin.cpp +
    np_batch_dropin.cpp in place of the hot-path translation units): all records through ONE call of
    np_calculate_methylation_for_batch.  records: dicts(seq [the read's own sequence], raw, rc, pos, cigar, bam_seq).
    Returns (list of per-record dicts of site arrays, status array)."""
    L = C.CDLL(_BATCH)
    n = len(records)
    raw = np.concatenate([np.ascontiguousarray(r["raw"], np.float32) for r in records])
    raw_off = np.zeros(n + 1, np.int64); raw_off[1:] = np.cumsum([len(r["raw"]) for r in records])
    cig = np.concatenate([np.ascontiguousarray(r["cigar"], np.uint32) for r in records])
    cig_off = np.zeros(n + 1, np.int64); cig_off[1:] = np.cumsum([len(r["cigar"]) for r in records])
    is_rev = np.array([int(r["rc"]) for r in records], np.int32); pos = np.array([int(r["pos"]) for r in records], np.int32)
    seqs = (C.c_char_p * n)(*[r["seq"].encode() for r in records])
    bseqs = (C.c_char_p * n)(*[r["bam_seq"].encode() for r in records])
    site_off = np.zeros(n + 1, np.int64); status = np.zeros(n, np.int32)
    st, en, nm = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    lu, lm = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
    sq = C.create_string_buffer(cap * 256)
    tot = L.npfull_call_methylation_batch(n, seqs, _p(raw, C.POINTER(C.c_float)), _p(raw_off, C.POINTER(C.c_int64)), _p(is_rev, _i32p),
                                          _p(pos, _i32p), _p(cig, _u32p), _p(cig_off, C.POINTER(C.c_int64)), bseqs, contig_seq.encode(),
                                          methylation_type.encode(), cap, _p(site_off, C.POINTER(C.c_int64)), _p(st, _i32p), _p(en, _i32p),
                                          _p(nm, _i32p), _p(lu, _f64p), _p(lm, _f64p), sq, _p(status, _i32p))
    assert tot <= cap
    out = []
    for i in range(n):
        a, b = int(site_off[i]), int(site_off[i + 1])
        out.append(dict(start=st[a:b].copy(), end=en[a:b].copy(), n_motif=nm[a:b].copy(), ll_unmeth=lu[a:b].copy(), ll_meth=lm[a:b].copy(),
                        sequence=[sq.raw[j * 256:(j + 1) * 256].split(b"\0", 1)[0].decode() for j in range(a, b)]))
    return out, status


def _batch_args(records):
    n = len(records)
    raw = np.concatenate([np.ascontiguousarray(r["raw"], np.float32) for r in records])
    raw_off = np.zeros(n + 1, np.int64); raw_off[1:] = np.cumsum([len(r["raw"]) for r in records])
    cig = np.concatenate([np.ascontiguousarray(r["cigar"], np.uint32) for r in records])
    cig_off = np.zeros(n + 1, np.int64); cig_off[1:] = np.cumsum([len(r["cigar"]) for r in records])
    is_rev = np.array([int(r["rc"]) for r in records], np.int32); pos = np.array([int(r["pos"]) for r in records], np.int32)
    seqs = (C.c_char_p * n)(*[r["seq"].encode() for r in records])
    bseqs = (C.c_char_p * n)(*[r["bam_seq"].encode() for r in records])
    return n, raw, raw_off, cig, cig_off, is_rev, pos, seqs, bseqs


def call_methylation_pipeline(records, contig_seq, batch_size, methylation_type="cpg", event_cap_divisor=2, rna=None, cap=65536, adc=None, contexts=0):
    """The records through NpBatchPipeline (nanopolish_amd/csrc/np_batch_dropin.cpp) in batches of batch_size, as many in flight as the
    pipeline takes.  contexts = 0: the process-wide context; contexts = N >= 1: N contexts of the pipeline's own on device 0 (the
    multi-GPU form on one device), batches dealt round-robin.  adc = (offset, raw_unit): the records carry int16 ADC counts in r["adc"] instead of pA samples (converted on the device).
    event_cap_divisor > 2 shrinks the device detector's event capacity (overflow route); rna: indices of records flagged
    as RNA reads.  Returns (list of per-record dicts of site arrays, status array)."""
    L = C.CDLL(_BATCH)
    n, raw, raw_off, cig, cig_off, is_rev, pos, seqs, bseqs = _batch_args(records)
    mask = np.zeros(n, np.uint8)
    for i in (rna or []):
        mask[i] = 1
    site_off = np.zeros(n + 1, np.int64); status = np.zeros(n, np.int32)
    st, en, nm = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    lu, lm = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
    counts = np.concatenate([np.ascontiguousarray(r["adc"], np.int16) for r in records]) if adc else None
    tot = L.npfull_call_methylation_pipeline(n, int(batch_size), int(contexts), int(event_cap_divisor), _p(mask, _u8p),
                                             _p(counts, C.POINTER(C.c_int16)) if adc else None, C.c_float(adc[0] if adc else 0.0),
                                             C.c_float(adc[1] if adc else 1.0), seqs, _p(raw, C.POINTER(C.c_float)),
                                             _p(raw_off, C.POINTER(C.c_int64)), _p(is_rev, _i32p), _p(pos, _i32p), _p(cig, _u32p),
                                             _p(cig_off, C.POINTER(C.c_int64)), bseqs, contig_seq.encode(), methylation_type.encode(), cap,
                                             _p(site_off, C.POINTER(C.c_int64)), _p(st, _i32p), _p(en, _i32p), _p(nm, _i32p), _p(lu, _f64p),
                                             _p(lm, _f64p), _p(status, _i32p))
    assert tot <= cap
    out = []
    for i in range(n):
        a, b = int(site_off[i]), int(site_off[i + 1])
        out.append(dict(start=st[a:b].copy(), end=en[a:b].copy(), n_motif=nm[a:b].copy(), ll_unmeth=lu[a:b].copy(), ll_meth=lm[a:b].copy()))
    return out, status


def bench_batch(records, contig_seq, batch_size, n_batches, warmup=2, pipelined=True, adc=None, contexts=0, consumer=1):
    """Seconds for n_batches batches of batch_size records (the given distinct records, cycled) through the batched binding:
    NpBatchPipeline (pipelined; contexts as call_methylation_pipeline) or the synchronous np_calculate_methylation_for_batch.  consumer:
    the harness's stand-in for the batch writer -- 0: count the sites and results.clear() on the calling thread (the reference's
    write_methylation_results_for_batch), 1: count the sites and hand the maps back with NpBatchPipeline::recycle.
    Returns (seconds, sites written, records that did not come back NP_BATCH_OK, host seconds per phase of the timed batches)."""
    L = C.CDLL(_BATCH)
    L.npfull_bench_batch.restype = C.c_double
    n, raw, raw_off, cig, cig_off, is_rev, pos, seqs, bseqs = _batch_args(records)
    n_sites, n_bad = C.c_int64(0), C.c_int64(0)
    hs = np.zeros(10, np.float64)
    counts = np.concatenate([np.ascontiguousarray(r["adc"], np.int16) for r in records]) if adc else None
    sec = L.npfull_bench_batch(n, seqs, _p(raw, C.POINTER(C.c_float)), _p(raw_off, C.POINTER(C.c_int64)), _p(is_rev, _i32p), _p(pos, _i32p),
                               _p(cig, _u32p), _p(cig_off, C.POINTER(C.c_int64)), bseqs, contig_seq.encode(), int(batch_size), int(n_batches),
                               int(warmup), int(bool(pipelined)), int(contexts), int(consumer), _p(counts, C.POINTER(C.c_int16)) if adc else None,
                               C.c_float(adc[0] if adc else 0.0), C.c_float(adc[1] if adc else 1.0), C.byref(n_sites), C.byref(n_bad), _p(hs, _f64p))
    names = ("phase1a_fetch_sizes", "phase1b_pack", "enqueue", "finisher_wait_device", "phase3_maps", "buffer_growth", "collect_wait", "submit",
             "caller_inside_binding", "caller_writer_stand_in")
    return float(sec), int(n_sites.value), int(n_bad.value), {k: round(float(v), 4) for k, v in zip(names, hs)}


def realign_batch(records, contig_seq, sample_rate=4000.0, rna=None):
    """The records through ONE np_realign_reads_batch (nanopolish_amd/csrc/np_eventalign_dropin.cpp; libnp_ref_full_batch.so): list of
    per-record dicts with the rebuilt SquiggleRead's fields (n_events, shift, scale, var, events_per_base, event mean / stdv / duration /
    start_time, event map), the EventAlignment rows and the TSV text the reference's own writer prints from them; and the status array."""
    L = C.CDLL(_BATCH)
    n, raw, raw_off, cig, cig_off, is_rev, pos, seqs, bseqs = _batch_args(records)
    ecap = np.array([len(r["raw"]) // 2 + 2 for r in records], np.int64)
    ev_off = np.zeros(n + 1, np.int64); ev_off[1:] = np.cumsum(ecap)
    mcap = np.array([len(r["seq"]) for r in records], np.int64)
    map_off = np.zeros(n + 1, np.int64); map_off[1:] = np.cumsum(mcap)
    status = np.zeros(n, np.int32); n_events = np.zeros(n, np.int32)
    sh, sc, va, epb = (np.zeros(n, np.float64) for _ in range(4))
    evm, evs, evd = (np.zeros(int(ev_off[-1]), np.float32) for _ in range(3)); evt = np.zeros(int(ev_off[-1]), np.float64)
    ms, me = np.full(int(map_off[-1]), -2, np.int32), np.full(int(map_off[-1]), -2, np.int32)
    row_cap = int(ev_off[-1]) + n
    row_off = np.zeros(n + 1, np.int64); rp = np.zeros(row_cap, np.int32); ei = np.zeros(row_cap, np.int32); st = C.create_string_buffer(row_cap)
    tsv_cap = 160 * row_cap
    tsv = C.create_string_buffer(tsv_cap); tsv_off = np.zeros(n + 1, np.int64)
    i64 = C.POINTER(C.c_int64)
    mask = np.zeros(n, np.uint8)             # rna: indices of the records that are direct-RNA reads (a batch may mix both types)
    for i in (rna or []):
        mask[i] = 1
    L.npfull_realign_batch(n, seqs, _p(raw, C.POINTER(C.c_float)), _p(raw_off, i64), _p(is_rev, _i32p), _p(pos, _i32p), _p(cig, _u32p), _p(cig_off, i64),
                           bseqs, contig_seq.encode(), C.c_double(sample_rate), _p(status, _i32p), _p(n_events, _i32p), _p(sh, _f64p), _p(sc, _f64p),
                           _p(va, _f64p), _p(epb, _f64p), _p(ev_off, i64), _p(evm, _f32p), _p(evs, _f32p), _p(evd, _f32p), _p(evt, _f64p),
                           _p(map_off, i64), _p(ms, _i32p), _p(me, _i32p), C.c_int64(row_cap), _p(row_off, i64), _p(rp, _i32p), _p(ei, _i32p), st,
                           tsv, C.c_int64(tsv_cap), _p(tsv_off, i64), _p(mask, _u8p))
    assert tsv_off[-1] < tsv_cap and row_off[-1] <= row_cap
    out = []
    for i in range(n):
        a, b = int(ev_off[i]), int(ev_off[i]) + int(n_events[i])
        k_n = len(records[i]["seq"]) - (4 if mask[i] else 5)          # k-mers of the read: k = 5 for RNA, 6 for DNA
        out.append(dict(n_events=int(n_events[i]), shift=float(sh[i]), scale=float(sc[i]), var=float(va[i]), events_per_base=float(epb[i]),
                        mean=evm[a:b].copy(), stdv=evs[a:b].copy(), duration=evd[a:b].copy(), start_time=evt[a:b].copy(),
                        map_start=ms[int(map_off[i]):int(map_off[i]) + k_n].copy(), map_stop=me[int(map_off[i]):int(map_off[i]) + k_n].copy(),
                        ref_position=rp[int(row_off[i]):int(row_off[i + 1])].copy(), event_idx=ei[int(row_off[i]):int(row_off[i + 1])].copy(),
                        hmm_state=np.frombuffer(st.raw[int(row_off[i]):int(row_off[i + 1])], np.uint8).copy(),
                        tsv=tsv.raw[int(tsv_off[i]):int(tsv_off[i + 1])].decode()))
    return out, status


def rows_hash(ref_position, event_idx, hmm_state):
    """The hash npfull_many_records takes of a record's rows: h <- h * P + word + 1 (mod 2^64) over (ref_position, event_idx, hmm_state)
    row by row -- here as one dot product with the powers of P (numpy's uint64 arithmetic wraps the same way)."""
    a = np.stack([np.asarray(ref_position).astype(np.uint32), np.asarray(event_idx).astype(np.uint32), np.asarray(hmm_state).astype(np.uint32)], 1)
    a = a.reshape(-1).astype(np.uint64) + np.uint64(1)
    m = len(a)
    if m == 0:
        return 0
    with np.errstate(over="ignore"):
        pw = np.cumprod(np.full(m, 0x9E3779B97F4A7C15, np.uint64))          # P^1 .. P^m
        w = np.concatenate([np.ones(1, np.uint64), pw[:-1]])[::-1]            # P^(m-1) .. P^0
        return int((a * w).sum(dtype=np.uint64))
