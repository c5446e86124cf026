"""ctypes bindings for the two CPU oracles (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess
import time
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libnp_ref.so")
GOLDEN = os.path.join(os.path.dirname(_HERE), "tests", "golden")

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)


def _p(a, t):
    return a.ctypes.data_as(t)


def build_port():
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"])


def build_ref():
    subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "ref"])


def have_ref():
    return os.path.exists(_REF)


def load_models(path=None):
    """Pore-model tables exported from the reference by tests/gen_golden.py (fixture, not source)."""
    z = np.load(path or os.path.join(GOLDEN, "models_r9.4_450bps.npz"))
    out = {}
    for alpha in ("nucleotide", "cpg"):
        out[alpha] = dict(k=6, level_mean=z[alpha + "_level_mean"], level_stdv=z[alpha + "_level_stdv"],
                          level_log_stdv=z[alpha + "_level_log_stdv"])
    return out


class _Model(C.Structure):
    _fields_ = [("k", C.c_int), ("n_states", C.c_int), ("level_mean", c_f64p),
                ("level_stdv", c_f64p), ("level_log_stdv", c_f64p)]


class _Scalings(C.Structure):
    _fields_ = [("shift", C.c_double), ("scale", C.c_double), ("drift", C.c_double),
                ("var", C.c_double), ("log_var", C.c_double)]


ED_DEFAULTS = dict(w1=3, w2=6, t1=1.4, t2=9.0, peak_height=0.2)      # event_detection_defaults, event_detection.h:15-21
ED_RNA = dict(w1=7, w2=14, t1=2.5, t2=9.0, peak_height=1.0)           # event_detection_rna, :23-29


def _detect_events(L, prefix, raw, w1, w2, t1, t2, peak_height):
    """detect_events on a whole raw table -> dict(start u64, length f32, mean f32, stdv f32) (f2)."""
    raw = np.ascontiguousarray(raw, np.float32)
    cap = len(raw) + 1
    st = np.zeros(cap, np.uint64); ln = np.zeros(cap, np.float32); mn = np.zeros(cap, np.float32); sd = np.zeros(cap, np.float32)
    fn = getattr(L, prefix + "_detect_events")
    fn.argtypes = [c_f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_float,
                   C.POINTER(C.c_uint64), c_f32p, c_f32p, c_f32p, C.c_size_t]
    n = fn(_p(raw, c_f32p), len(raw), w1, w2, t1, t2, peak_height, st.ctypes.data_as(C.POINTER(C.c_uint64)),
           _p(ln, c_f32p), _p(mn, c_f32p), _p(sd, c_f32p), cap)
    n = max(n, 0)
    return dict(start=st[:n].copy(), length=ln[:n].copy(), mean=mn[:n].copy(), stdv=sd[:n].copy())


def _detect_events_many(L, prefix, raw, raw_off, n_threads):
    raw = np.ascontiguousarray(raw, np.float32); raw_off = np.ascontiguousarray(raw_off, np.int64)
    n = len(raw_off) - 1
    ev_off = np.zeros(n + 1, np.int64); ev_off[1:] = np.cumsum((raw_off[1:] - raw_off[:-1]) // 2 + 2)
    out = np.zeros(int(ev_off[-1]), np.float32); out_n = np.zeros(n, np.int32)
    fn = getattr(L, prefix + "_detect_events_many")
    fn.argtypes = [C.c_int, c_f32p, c_i64p, c_f32p, c_i64p, c_i32p, C.c_int]
    t0 = time.perf_counter()
    fn(n, _p(raw, c_f32p), _p(raw_off, c_i64p), _p(out, c_f32p), _p(ev_off, c_i64p), _p(out_n, c_i32p), int(n_threads))
    return out, ev_off, out_n, time.perf_counter() - t0


class Oracle:
    """The portable C restatement (oracle/np_oracle.c)."""

    def __init__(self):
        if not os.path.exists(_PORT) or os.path.getmtime(_PORT) < os.path.getmtime(os.path.join(_HERE, "np_oracle.c")):
            build_port()
        L = self.L = C.CDLL(_PORT)
        L.npo_flogsum_table.restype = c_f32p
        L.npo_flogsum.restype = C.c_float
        L.npo_flogsum.argtypes = [C.c_float, C.c_float]
        L.npo_set4.restype = _Scalings
        L.npo_set4.argtypes = [C.c_double] * 4
        L.npo_log_probability_match_r9.restype = C.c_float
        L.npo_log_probability_match_r9.argtypes = [C.POINTER(_Model), C.POINTER(_Scalings), C.c_uint32, C.c_float, C.c_float]
        L.npo_profile_hmm_score.restype = C.c_float
        L.npo_profile_hmm_score.argtypes = [C.POINTER(_Model), C.POINTER(_Scalings), c_f32p, c_u32p, C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_double, C.c_uint32]
        L.npo_profile_hmm_align.restype = C.c_int
        L.npo_profile_hmm_align.argtypes = [C.POINTER(_Model), C.POINTER(_Scalings), c_f32p, c_u32p, C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_double, C.c_uint32,
                                            c_u32p, c_u32p, c_f64p, C.c_char_p, C.c_int]
        L.npo_combine_score_set.restype = C.c_float
        L.npo_combine_score_set.argtypes = [c_f32p, C.c_int]
        L.npo_adaptive_banded_simple_event_align.restype = C.c_int
        L.npo_adaptive_banded_simple_event_align.argtypes = [C.POINTER(_Model), C.POINTER(_Scalings), c_f32p, C.c_uint32,
                                                             c_u32p, C.c_uint32, c_i32p, C.c_int]
        L.npo_kmer_rank.restype = C.c_uint32
        L.npo_kmer_rank.argtypes = [C.c_int, C.c_char_p, C.c_uint32]
        L.npo_alphabet_id.argtypes = [C.c_char_p]
        L.npo_get_closest_event_to.argtypes = [c_i32p, C.c_uint32, C.c_int]
        self._keep = []

    # -- helpers -------------------------------------------------------------------------------
    def model(self, m):
        lm = np.ascontiguousarray(m["level_mean"], np.float64)
        ls = np.ascontiguousarray(m["level_stdv"], np.float64)
        ll = np.ascontiguousarray(m["level_log_stdv"], np.float64)
        self._keep.append((lm, ls, ll))
        return _Model(int(m["k"]), len(lm), _p(lm, c_f64p), _p(ls, c_f64p), _p(ll, c_f64p))

    def scalings(self, shift, scale, var, drift=0.0):
        return self.L.npo_set4(shift, scale, drift, var)

    def alphabet_id(self, name):
        return self.L.npo_alphabet_id(name.encode())

    # -- tables / primitives ---------------------------------------------------------------------
    def flogsum_table(self):
        return np.ctypeslib.as_array(self.L.npo_flogsum_table(), shape=(16000,)).copy()

    def flogsum(self, a, b):
        return self.L.npo_flogsum(a, b)

    def kmer_rank(self, alpha, kmer):
        return self.L.npo_kmer_rank(self.alphabet_id(alpha), kmer.encode(), len(kmer))

    def _strfn(self, fn, alpha, s):
        out = C.create_string_buffer(len(s) + 1)
        getattr(self.L, fn)(self.alphabet_id(alpha), s.encode(), len(s), out)
        return out.value.decode()

    def reverse_complement(self, alpha, s):
        return self._strfn("npo_reverse_complement", alpha, s)

    def methylate(self, alpha, s):
        return self._strfn("npo_methylate", alpha, s)

    def unmethylate(self, alpha, s):
        return self._strfn("npo_unmethylate", alpha, s)

    def is_motif_match(self, alpha, s, i):
        return bool(self.L.npo_is_motif_match(self.alphabet_id(alpha), s.encode(), len(s), i))

    def sequence_kmer_ranks(self, alpha, seq, rc_seq, k, do_rc):
        n = len(seq)
        out = np.zeros(n - k + 1, np.uint32)
        self.L.npo_sequence_kmer_ranks(self.alphabet_id(alpha), seq.encode(), rc_seq.encode() if rc_seq else None,
                                       n, k, int(do_rc), _p(out, c_u32p))
        return out

    def log_probability_match_r9(self, model, sc, rank, level, time=0.0):
        return self.L.npo_log_probability_match_r9(C.byref(model), C.byref(sc), rank, level, time)

    def calculate_transitions(self, events_per_base, indel_bias=1.0):
        out = np.zeros(10, np.float32)
        self.L.npo_calculate_transitions(C.c_double(events_per_base), C.c_double(indel_bias), _p(out, c_f32p))
        return out

    def make_flanks(self, n_events):
        pre = np.zeros(n_events + 1, np.float32)
        post = np.zeros(n_events, np.float32)
        self.L.npo_make_flanks(C.c_uint32(n_events), _p(pre, c_f32p), _p(post, c_f32p))
        return pre, post

    # -- HMM ---------------------------------------------------------------------------------------
    def hmm_score(self, model, sc, events, ranks, e_start, e_stop, stride, events_per_base, indel_bias=1.0, flags=0):
        events = np.ascontiguousarray(events, np.float32)
        ranks = np.ascontiguousarray(ranks, np.uint32)
        return self.L.npo_profile_hmm_score(C.byref(model), C.byref(sc), _p(events, c_f32p), _p(ranks, c_u32p),
                                            len(ranks), e_start, e_stop, stride, events_per_base, indel_bias, flags)

    def hmm_align(self, model, sc, events, ranks, e_start, e_stop, stride, events_per_base, indel_bias=1.0, flags=0):
        events = np.ascontiguousarray(events, np.float32)
        ranks = np.ascontiguousarray(ranks, np.uint32)
        cap = 3 * (abs(int(e_stop) - int(e_start)) + 1 + len(ranks)) + 8
        ev = np.zeros(cap, np.uint32); km = np.zeros(cap, np.uint32); lf = np.zeros(cap, np.float64)
        st = C.create_string_buffer(cap)
        n = self.L.npo_profile_hmm_align(C.byref(model), C.byref(sc), _p(events, c_f32p), _p(ranks, c_u32p), len(ranks),
                                         e_start, e_stop, stride, events_per_base, indel_bias, flags,
                                         _p(ev, c_u32p), _p(km, c_u32p), _p(lf, c_f64p), st, cap)
        if n < 0:
            return None
        return ev[:n].copy(), km[:n].copy(), lf[:n].copy(), np.frombuffer(st.raw[:n], np.uint8).copy()

    def combine_score_set(self, scores):
        s = np.ascontiguousarray(scores, np.float32)
        return self.L.npo_combine_score_set(_p(s, c_f32p), len(s))

    # -- raw loader -----------------------------------------------------------------------------------
    def estimate_scalings_mom(self, model, ranks, events):
        events = np.ascontiguousarray(events, np.float32)
        ranks = np.ascontiguousarray(ranks, np.uint32)
        sh = C.c_double(); sc = C.c_double()
        self.L.npo_estimate_scalings_mom(C.byref(model), _p(ranks, c_u32p), C.c_uint32(len(ranks)),
                                         _p(events, c_f32p), C.c_uint32(len(events)), C.byref(sh), C.byref(sc))
        return sh.value, sc.value

    def event_align(self, model, sc, events, ranks):
        events = np.ascontiguousarray(events, np.float32)
        ranks = np.ascontiguousarray(ranks, np.uint32)
        cap = len(events) + len(ranks) + 2
        out = np.zeros((cap, 2), np.int32)
        n = self.L.npo_adaptive_banded_simple_event_align(C.byref(model), C.byref(sc), _p(events, c_f32p), len(events),
                                                          _p(ranks, c_u32p), len(ranks), _p(out, c_i32p), cap)
        if n < 0:
            return None
        return out[:n].copy()

    # -- glue --------------------------------------------------------------------------------------------
    def build_base_to_event_map(self, pairs, n_kmers):
        pairs = np.ascontiguousarray(pairs, np.int32)
        start = np.zeros(n_kmers, np.int32); stop = np.zeros(n_kmers, np.int32)
        epb = C.c_double()
        self.L.npo_build_base_to_event_map(_p(pairs, c_i32p), len(pairs), C.c_uint32(n_kmers),
                                           _p(start, c_i32p), _p(stop, c_i32p), C.byref(epb))
        return start, stop, epb.value

    def recalibrate(self, model, events, ranks, map_start, map_stop):
        events = np.ascontiguousarray(events, np.float32); ranks = np.ascontiguousarray(ranks, np.uint32)
        ms = np.ascontiguousarray(map_start, np.int32); mp = np.ascontiguousarray(map_stop, np.int32)
        sh = C.c_double(); sc = C.c_double(); va = C.c_double()
        ok = self.L.npo_recalibrate(C.byref(model), _p(events, c_f32p), _p(ranks, c_u32p), C.c_uint32(len(ranks)),
                                    _p(ms, c_i32p), _p(mp, c_i32p), C.byref(sh), C.byref(sc), C.byref(va))
        return (sh.value, sc.value, va.value) if ok else None

    def get_closest_event_to(self, start, k_idx):
        start = np.ascontiguousarray(start, np.int32)
        return self.L.npo_get_closest_event_to(_p(start, c_i32p), len(start), int(k_idx))

    def event_alignment_record(self, aligned_bases, read_length, k, seq_rc, map_start):
        ab = np.ascontiguousarray(aligned_bases, np.int32)
        ms = np.ascontiguousarray(map_start, np.int32)
        out = np.zeros((len(ab), 2), np.int32)
        n = self.L.npo_event_alignment_record(_p(ab, c_i32p), len(ab), read_length, k, int(seq_rc),
                                              _p(ms, c_i32p), C.c_uint32(len(ms)), _p(out, c_i32p))
        return out[:n].copy()

    def cigar_aligned_bases(self, cigar, pos=0):
        cg = np.ascontiguousarray(cigar, np.uint32)
        n = self.L.npo_cigar_aligned_bases(_p(cg, c_u32p), len(cg), int(pos), None, 0)
        if n < 0:
            return None
        out = np.zeros((n, 2), np.int32)
        self.L.npo_cigar_aligned_bases(_p(cg, c_u32p), len(cg), int(pos), _p(out, c_i32p), n)
        return out

    def find_by_ref_bounds(self, pairs, ref_start, ref_stop):
        pairs = np.ascontiguousarray(pairs, np.int32)
        a = C.c_int(); b = C.c_int()
        ok = self.L.npo_find_by_ref_bounds(_p(pairs, c_i32p), len(pairs), ref_start, ref_stop, C.byref(a), C.byref(b))
        return (a.value, b.value) if ok else None

    def scan_motif_groups(self, alpha, ref_seq, min_separation=10):
        cap = len(ref_seq) + 1
        f = np.zeros(cap, np.int32); l = np.zeros(cap, np.int32); c = np.zeros(cap, np.int32)
        n = self.L.npo_scan_motif_groups(self.alphabet_id(alpha), ref_seq.encode(), len(ref_seq), min_separation,
                                         _p(f, c_i32p), _p(l, c_i32p), _p(c, c_i32p), cap)
        return f[:n].copy(), l[:n].copy(), c[:n].copy()

    def detect_events(self, raw, w1=3, w2=6, t1=1.4, t2=9.0, peak_height=0.2):
        return _detect_events(self.L, "npo", raw, w1, w2, t1, t2, peak_height)

    def aligner_constants(self, n_events, n_kmers):
        out = np.zeros(4)
        self.L.npo_aligner_constants(C.c_uint32(n_events), C.c_uint32(n_kmers), _p(out, c_f64p))
        return tuple(float(v) for v in out)

    def detect_events_many(self, raw, raw_off, n_threads=1):
        out, ev_off, out_n, self.last_call_s = _detect_events_many(self.L, "npo", raw, raw_off, n_threads)
        return out, ev_off, out_n

    # -- bounded drivers for the CPU baseline ---------------------------------------------------------------
    def align_many(self, model, events, event_off, ranks, rank_off, shift, scale, n_threads=1):
        n_reads = len(event_off) - 1
        pair_off = np.zeros(n_reads + 1, np.int64)
        pair_off[1:] = np.cumsum((event_off[1:] - event_off[:-1]) + (rank_off[1:] - rank_off[:-1]) + 2)
        out = np.zeros((int(pair_off[-1]), 2), np.int32)
        out_n = np.zeros(n_reads, np.int32)
        events = np.ascontiguousarray(events, np.float32); ranks = np.ascontiguousarray(ranks, np.uint32)
        event_off = np.ascontiguousarray(event_off, np.int64); rank_off = np.ascontiguousarray(rank_off, np.int64)
        shift = np.ascontiguousarray(shift, np.float64); scale = np.ascontiguousarray(scale, np.float64)
        _t0 = time.perf_counter()
        self.L.npo_align_many(C.byref(model), n_reads, _p(events, c_f32p), _p(event_off, c_i64p), _p(ranks, c_u32p),
                              _p(rank_off, c_i64p), _p(shift, c_f64p), _p(scale, c_f64p), _p(out, c_i32p),
                              _p(pair_off, c_i64p), _p(out_n, c_i32p), int(n_threads))
        self.last_call_s = time.perf_counter() - _t0
        return out, pair_off, out_n

    def score_many(self, model, job_read, events, event_off, shift, scale, var, epb, ranks, job_rank_off,
                   e_start, e_stop, stride, indel_bias=1.0, flags=3, n_threads=1):
        n_jobs = len(job_read)
        out = np.zeros(n_jobs, np.float32)
        a = lambda x, t: np.ascontiguousarray(x, t)
        job_read = a(job_read, np.int32); events = a(events, np.float32); event_off = a(event_off, np.int64)
        shift = a(shift, np.float64); scale = a(scale, np.float64); var = a(var, np.float64); epb = a(epb, np.float64)
        ranks = a(ranks, np.uint32); job_rank_off = a(job_rank_off, np.int64)
        e_start = a(e_start, np.uint32); e_stop = a(e_stop, np.uint32); stride = a(stride, np.int8)
        _t0 = time.perf_counter()
        self.L.npo_score_many(C.byref(model), C.c_int64(n_jobs), _p(job_read, c_i32p), _p(events, c_f32p),
                              _p(event_off, c_i64p), _p(shift, c_f64p), _p(scale, c_f64p), _p(var, c_f64p),
                              _p(epb, c_f64p), _p(ranks, c_u32p), _p(job_rank_off, c_i64p), _p(e_start, c_u32p),
                              _p(e_stop, c_u32p), stride.ctypes.data_as(C.POINTER(C.c_int8)),
                              C.c_double(indel_bias), C.c_uint32(flags), _p(out, c_f32p), int(n_threads))
        self.last_call_s = time.perf_counter() - _t0
        return out


class RefOracle:
    """The reference's own code (oracle/_ref/libnp_ref.so). Only where it has been built."""
    KIT = b"r9.4_450bps"

    def __init__(self, path=None):
        """path: another build of the same harness, e.g. oracle/_ref/libnp_ref_dropin.so (the reference with its two
        hot-path translation units replaced by the product's shim) -- used by tests/test_gpu_dropin.py."""
        if path is None and not have_ref():
            if os.path.isdir("/root/reference"):
                build_ref()
            else:
                raise RuntimeError("oracle/_ref/libnp_ref.so not built and /root/reference absent")
        L = self.L = C.CDLL(path or _REF)
        L.npref_add_logs.restype = C.c_float
        L.npref_add_logs.argtypes = [C.c_float, C.c_float]
        L.npref_log_probability_match_r9.restype = C.c_float
        L.npref_log_probability_match_r9.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_float] + [C.c_double] * 4
        L.npref_hmm_score.restype = C.c_float
        L.npref_hmm_score.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, c_f32p, C.c_int, C.c_uint32,
                                      C.c_uint32, C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_uint32]
        L.npref_hmm_score_set.restype = C.c_float
        L.npref_hmm_align.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, c_f32p, C.c_int, C.c_uint32,
                                      C.c_uint32, C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_uint32,
                                      c_u32p, c_u32p, c_f64p, C.c_char_p, C.c_int]
        L.npref_event_align.argtypes = [C.c_char_p, c_f32p, C.c_int, C.c_char_p] + [C.c_double] * 4 + [c_i32p, C.c_int]
        L.npref_kmer_rank.argtypes = [C.c_char_p, C.c_char_p, C.c_int]

    def flogsum_table(self):
        t = np.zeros(16000, np.float32)
        self.L.npref_flogsum_table(_p(t, c_f32p))
        return t

    def add_logs(self, a, b):
        return self.L.npref_add_logs(a, b)

    def model(self, alphabet, k=6, kit=None):
        kit = kit or self.KIT
        n = self.L.npref_model_size(kit, alphabet.encode(), k)
        lm = np.zeros(n); ls = np.zeros(n); ll = np.zeros(n)
        self.L.npref_model_get(kit, alphabet.encode(), k, _p(lm, c_f64p), _p(ls, c_f64p), _p(ll, c_f64p))
        return dict(k=k, level_mean=lm, level_stdv=ls, level_log_stdv=ll)

    def set_indel_bias(self, v):
        """hmm_indel_bias_factor (src/hmm/nanopolish_profile_hmm_r9.cpp:19) for the *_many drivers (the single-call entry points
        take it per call and reset it to 1.0)"""
        self.L.npref_set_indel_bias.argtypes = [C.c_double]
        self.L.npref_set_indel_bias(float(v))

    def shift_model(self, alphabet, delta, k=6):
        """test hook: add delta to every level_mean of the registered model, in place (same PoreModel address)"""
        self.L.npref_shift_model.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_double]
        self.L.npref_shift_model(self.KIT, alphabet.encode(), k, delta)

    def kmer_rank(self, alpha, kmer):
        return self.L.npref_kmer_rank(alpha.encode(), kmer.encode(), len(kmer))

    def _strfn(self, fn, alpha, s):
        out = C.create_string_buffer(len(s) + 8)
        getattr(self.L, fn)(alpha.encode(), s.encode(), out)
        return out.value.decode()

    def reverse_complement(self, alpha, s):
        return self._strfn("npref_reverse_complement", alpha, s)

    def methylate(self, alpha, s):
        return self._strfn("npref_methylate", alpha, s)

    def unmethylate(self, alpha, s):
        return self._strfn("npref_unmethylate", alpha, s)

    def is_motif_match(self, alpha, s, i):
        return bool(self.L.npref_is_motif_match(alpha.encode(), s.encode(), i))

    def log_probability_match_r9(self, alphabet, rank, level, shift, scale, drift, var):
        return self.L.npref_log_probability_match_r9(self.KIT, alphabet.encode(), rank, level, shift, scale, drift, var)

    def estimate_scalings_mom(self, seq, events):
        events = np.ascontiguousarray(events, np.float32)
        sh = C.c_double(); sc = C.c_double()
        self.L.npref_estimate_scalings_mom(self.KIT, seq.encode(), _p(events, c_f32p), len(events), C.byref(sh), C.byref(sc))
        return sh.value, sc.value

    def event_align(self, events, seq, shift, scale, var=1.0, drift=0.0):
        events = np.ascontiguousarray(events, np.float32)
        cap = len(events) + len(seq) + 2
        out = np.zeros((cap, 2), np.int32)
        n = self.L.npref_event_align(self.KIT, _p(events, c_f32p), len(events), seq.encode(), shift, scale, drift, var,
                                     _p(out, c_i32p), cap)
        return out[:n].copy()

    def hmm_score(self, alphabet, seq, rc_seq, events, e_start, e_stop, stride, rc, shift, scale, var,
                  events_per_base, indel_bias=1.0, flags=0):
        events = np.ascontiguousarray(events, np.float32)
        return self.L.npref_hmm_score(self.KIT, alphabet.encode(), seq.encode(), rc_seq.encode() if rc_seq else None,
                                      _p(events, c_f32p), len(events), e_start, e_stop, stride, int(rc),
                                      shift, scale, var, events_per_base, indel_bias, flags)

    def hmm_score_set(self, seqs, alphabets, events, e_start, e_stop, stride, rc, shift, scale, var,
                      events_per_base, indel_bias=1.0, flags=0):
        events = np.ascontiguousarray(events, np.float32)
        n = len(seqs)
        sa = (C.c_char_p * n)(*[s.encode() for s in seqs])
        aa = (C.c_char_p * n)(*[a.encode() for a in alphabets])
        self.L.npref_hmm_score_set.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), c_f32p,
                                               C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_uint32]
        return self.L.npref_hmm_score_set(self.KIT, n, sa, aa, _p(events, c_f32p), len(events), e_start, e_stop,
                                          stride, int(rc), shift, scale, var, events_per_base, indel_bias, flags)

    def hmm_score_vec(self, alphabet, seq, datas, indel_bias=1.0, flags=0):
        """profile_hmm_score(sequence, std::vector<HMMInputData>, flags) (src/hmm/nanopolish_profile_hmm.cpp:14-21): one sequence
        against several reads' windows.  datas: dicts(events, e_start, e_stop, stride, rc, shift, scale, var, events_per_base)."""
        n = len(datas)
        ev = np.concatenate([np.ascontiguousarray(d["events"], np.float32) for d in datas])
        eo = np.zeros(n + 1, np.int64); eo[1:] = np.cumsum([len(d["events"]) for d in datas])
        a = lambda k, t: np.ascontiguousarray([d[k] for d in datas], t)
        e1, e2, st, rc = a("e_start", np.uint32), a("e_stop", np.uint32), a("stride", np.int32), a("rc", np.int32)
        sh, sc, vr, epb = a("shift", np.float64), a("scale", np.float64), a("var", np.float64), a("events_per_base", np.float64)
        f = self.L.npref_hmm_score_vec
        f.restype = C.c_float
        f.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, c_f32p, c_i64p, c_u32p, c_u32p, c_i32p, c_i32p, c_f64p, c_f64p,
                      c_f64p, c_f64p, C.c_double, C.c_uint32]
        return f(self.KIT, alphabet.encode(), seq.encode(), n, _p(ev, c_f32p), _p(eo, c_i64p), _p(e1, c_u32p), _p(e2, c_u32p),
                 _p(st, c_i32p), _p(rc, c_i32p), _p(sh, c_f64p), _p(sc, c_f64p), _p(vr, c_f64p), _p(epb, c_f64p), indel_bias, flags)

    def hmm_align(self, alphabet, seq, rc_seq, events, e_start, e_stop, stride, rc, shift, scale, var,
                  events_per_base, indel_bias=1.0, flags=0):
        events = np.ascontiguousarray(events, np.float32)
        cap = 3 * (abs(int(e_stop) - int(e_start)) + 1 + len(seq)) + 8
        ev = np.zeros(cap, np.uint32); km = np.zeros(cap, np.uint32); lf = np.zeros(cap, np.float64)
        st = C.create_string_buffer(cap)
        n = self.L.npref_hmm_align(self.KIT, alphabet.encode(), seq.encode(), rc_seq.encode() if rc_seq else None,
                                   _p(events, c_f32p), len(events), e_start, e_stop, stride, int(rc),
                                   shift, scale, var, events_per_base, indel_bias, flags,
                                   _p(ev, c_u32p), _p(km, c_u32p), _p(lf, c_f64p), st, cap)
        return ev[:n].copy(), km[:n].copy(), lf[:n].copy(), np.frombuffer(st.raw[:n], np.uint8).copy()

    def detect_events(self, raw, w1=3, w2=6, t1=1.4, t2=9.0, peak_height=0.2):
        return _detect_events(self.L, "npref", raw, w1, w2, t1, t2, peak_height)

    def detect_events_many(self, raw, raw_off, n_threads=1):
        out, ev_off, out_n, self.last_call_s = _detect_events_many(self.L, "npref", raw, raw_off, n_threads)
        return out, ev_off, out_n

    # -- bounded drivers for the CPU baseline (kind="reference") --------------------------------------------
    def align_many(self, seqs, events, event_off, shift, scale, n_threads=1):
        n = len(seqs)
        event_off = np.ascontiguousarray(event_off, np.int64)
        pair_off = np.zeros(n + 1, np.int64)
        pair_off[1:] = np.cumsum((event_off[1:] - event_off[:-1]) + np.array([len(s) for s in seqs]) + 2)
        out = np.zeros((int(pair_off[-1]), 2), np.int32); out_n = np.zeros(n, np.int32)
        events = np.ascontiguousarray(events, np.float32)
        shift = np.ascontiguousarray(shift, np.float64); scale = np.ascontiguousarray(scale, np.float64)
        sa = (C.c_char_p * n)(*[s.encode() for s in seqs])
        self.L.npref_align_many.argtypes = [C.c_char_p, C.c_int, c_f32p, c_i64p, C.POINTER(C.c_char_p), c_f64p, c_f64p,
                                            c_i32p, c_i64p, c_i32p, C.c_int]
        _t0 = time.perf_counter()
        self.L.npref_align_many(self.KIT, n, _p(events, c_f32p), _p(event_off, c_i64p), sa, _p(shift, c_f64p),
                                _p(scale, c_f64p), _p(out, c_i32p), _p(pair_off, c_i64p), _p(out_n, c_i32p), int(n_threads))
        self.last_call_s = time.perf_counter() - _t0
        return out, pair_off, out_n

    def score_many_reads(self, alphabet, events, event_off, shift, scale, var, epb, job_off, seqs, rc_seqs,
                         e_start, e_stop, stride, rc, flags=3, n_threads=1):
        n = len(event_off) - 1
        nj = len(seqs)
        a = lambda x, t: np.ascontiguousarray(x, t)
        events = a(events, np.float32); event_off = a(event_off, np.int64); job_off = a(job_off, np.int64)
        shift = a(shift, np.float64); scale = a(scale, np.float64); var = a(var, np.float64); epb = a(epb, np.float64)
        e_start = a(e_start, np.uint32); e_stop = a(e_stop, np.uint32); stride = a(stride, np.int32); rc = a(rc, np.int32)
        sa = (C.c_char_p * nj)(*[s.encode() for s in seqs]); ra = (C.c_char_p * nj)(*[s.encode() for s in rc_seqs])
        out = np.zeros(nj, np.float32)
        self.L.npref_score_many_reads.argtypes = [C.c_char_p, C.c_char_p, C.c_int, c_f32p, c_i64p, c_f64p, c_f64p, c_f64p,
                                                  c_f64p, c_i64p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), c_u32p,
                                                  c_u32p, c_i32p, c_i32p, C.c_uint32, c_f32p, C.c_int]
        _t0 = time.perf_counter()
        self.L.npref_score_many_reads(self.KIT, alphabet.encode(), n, _p(events, c_f32p), _p(event_off, c_i64p),
                                      _p(shift, c_f64p), _p(scale, c_f64p), _p(var, c_f64p), _p(epb, c_f64p),
                                      _p(job_off, c_i64p), sa, ra, _p(e_start, c_u32p), _p(e_stop, c_u32p),
                                      _p(stride, c_i32p), _p(rc, c_i32p), flags, _p(out, c_f32p), int(n_threads))
        self.last_call_s = time.perf_counter() - _t0
        return out
