"""Host-side mirror of the reference's entry points for this path, on top of the C ABI (include/np_hmm.h).

Names and argument meaning follow the reference:
    profile_hmm_score / profile_hmm_score_set / profile_hmm_align   src/hmm/nanopolish_profile_hmm.h:24-31
    adaptive_banded_simple_event_align, estimate_scalings_using_mom src/nanopolish_raw_loader.h:16-24
so the parity tests read like calls into nanopolish.  Everything computes on the MI355X through libnp_hip.so;
nothing here falls back to a CPU implementation.
"""
import ctypes as C
import numpy as np

from . import lib as _l

HAF_ALLOW_PRE_CLIP = 1
HAF_ALLOW_POST_CLIP = 2


def _p(a, t):
    return a.ctypes.data_as(t)


# ---- pure-host helpers (Alphabet / HMMInputSequence) -------------------------------------------------------
def alphabet_id(name):
    r = _l.load_library().np_alphabet_id(name.encode())
    if r < 0:
        raise ValueError("unknown alphabet %r" % name)
    return r


def kmer_rank(alphabet, kmer):
    return _l.load_library().np_kmer_rank(alphabet_id(alphabet), kmer.encode(), len(kmer))


def _strfn(fn, alphabet, s):
    out = C.create_string_buffer(len(s) + 1)
    getattr(_l.load_library(), fn)(alphabet_id(alphabet), s.encode(), len(s), out)
    return out.value.decode()


def reverse_complement(alphabet, s):
    return _strfn("np_reverse_complement", alphabet, s)


def methylate(alphabet, s):
    return _strfn("np_methylate", alphabet, s)


def unmethylate(alphabet, s):
    return _strfn("np_unmethylate", alphabet, s)


def is_motif_match(alphabet, s, i):
    return bool(_l.load_library().np_is_motif_match(alphabet_id(alphabet), s.encode(), len(s), i))


def sequence_kmer_ranks(alphabet, seq, rc_seq=None, k=6, do_rc=False):
    """HMMInputSequence(seq[, rc_seq], alphabet).get_kmer_rank(i, k, do_rc) for all i."""
    out = np.zeros(len(seq) - k + 1, np.uint16)
    rc = _l.load_library().np_sequence_kmer_ranks(alphabet_id(alphabet), seq.encode(), rc_seq.encode() if rc_seq else None,
                                                  len(seq), k, int(do_rc), _p(out, _l.c_u16p))
    if rc != 0:
        raise ValueError("np_sequence_kmer_ranks failed")
    return out


def calculate_transitions(events_per_base, indel_bias=1.0):
    out = np.zeros(10, np.float32)
    _l.load_library().np_calculate_transitions(events_per_base, indel_bias, _p(out, _l.c_f32p))
    return out


def estimate_scalings_using_mom(model, ranks, events):
    lm = np.ascontiguousarray(model["level_mean"], np.float64)
    ranks = np.ascontiguousarray(ranks, np.uint16)
    events = np.ascontiguousarray(events, np.float32)
    sh = C.c_double(); sc = C.c_double()
    _l.load_library().np_estimate_scalings_mom(_p(lm, _l.c_f64p), _p(ranks, _l.c_u16p), len(ranks), _p(events, _l.c_f32p),
                                               len(events), C.byref(sh), C.byref(sc))
    return sh.value, sc.value


def scan_motif_groups(alphabet, ref_seq, min_separation=10):
    cap = len(ref_seq) + 1
    f = np.zeros(cap, np.int32); l = np.zeros(cap, np.int32); c = np.zeros(cap, np.int32)
    n = _l.load_library().np_scan_motif_groups(alphabet_id(alphabet), ref_seq.encode(), len(ref_seq), min_separation,
                                               _p(f, _l.c_i32p), _p(l, _l.c_i32p), _p(c, _l.c_i32p), cap)
    return f[:n].copy(), l[:n].copy(), c[:n].copy()


def cm_build_jobs_identity(ref_seq, read_rc, k=6, alphabet="cpg", min_separation=10, min_flank=10):
    """Work items of calculate_methylation_for_read for an identity-aligned read (see np_cm_build_jobs_identity)."""
    n = len(ref_seq)
    cap_jobs = n // 2 + 1
    cap_ranks = 4 * n + 1024
    f = np.zeros(cap_jobs, np.int32); l = np.zeros(cap_jobs, np.int32); c = np.zeros(cap_jobs, np.int32)
    kpos = np.zeros(2 * cap_jobs, np.int32); nk = np.zeros(cap_jobs, np.int32)
    ru = np.zeros(cap_ranks, np.uint16); rm = np.zeros(cap_ranks, np.uint16); ro = np.zeros(cap_jobs + 1, np.int64)
    nj = _l.load_library().np_cm_build_jobs_identity(alphabet_id(alphabet), ref_seq.encode(), n, int(read_rc), k,
                                                     min_separation, min_flank, cap_jobs, cap_ranks,
                                                     _p(f, _l.c_i32p), _p(l, _l.c_i32p), _p(c, _l.c_i32p), _p(kpos, _l.c_i32p),
                                                     _p(nk, _l.c_i32p), _p(ru, _l.c_u16p), _p(rm, _l.c_u16p), _p(ro, _l.c_i64p))
    if nj < 0:
        raise RuntimeError("np_cm_build_jobs_identity: %d" % nj)
    w = int(ro[nj])
    return dict(first=f[:nj].copy(), last=l[:nj].copy(), n_motif=c[:nj].copy(), kpos=kpos[:2 * nj].reshape(-1, 2).copy(),
                n_kmers=nk[:nj].copy(), ranks_unmeth=ru[:w].copy(), ranks_meth=rm[:w].copy(), rank_off=ro[:nj + 1].copy())


CIGAR_OPS = "MIDNSHP=X"


def cigar_words(ops):
    """[(op_char, length), ...] -> uint32 BAM CIGAR words (length << 4 | op)"""
    return np.array([(int(n) << 4) | CIGAR_OPS.index(o) for o, n in ops], np.uint32)


def cigar_aligned_bases(cigar, ref_pos0=0):
    """get_aligned_segments of a non-spliced record (see np_cigar_aligned_bases): int32[n, 2] of (ref_pos, read_pos)"""
    cg = np.ascontiguousarray(cigar, np.uint32)
    L = _l.load_library()
    n = L.np_cigar_aligned_bases(_p(cg, _l.c_u32p), len(cg), int(ref_pos0), None, None, 0)
    if n < 0:
        raise ValueError("np_cigar_aligned_bases: %d (spliced or malformed CIGAR)" % n)
    rp, qp = np.zeros(n, np.int32), np.zeros(n, np.int32)
    L.np_cigar_aligned_bases(_p(cg, _l.c_u32p), len(cg), int(ref_pos0), _p(rp, _l.c_i32p), _p(qp, _l.c_i32p), n)
    return np.stack([rp, qp], 1)


def cm_build_jobs_cigar(ref_seq, cigar, read_len, read_rc, k=6, alphabet="cpg", min_separation=10, min_flank=10):
    """Work items of calculate_methylation_for_read for a CIGAR-aligned read (see np_cm_build_jobs_cigar).
    ref_seq: the fetched reference segment contig[pos .. bam_endpos] (clipped); first/last are relative to pos."""
    n = len(ref_seq)
    cg = np.ascontiguousarray(cigar, np.uint32)
    cap_jobs = n // 2 + 1
    cap_ranks = 4 * n + 1024
    f = np.zeros(cap_jobs, np.int32); l = np.zeros(cap_jobs, np.int32); c = np.zeros(cap_jobs, np.int32)
    kpos = np.zeros(2 * cap_jobs, np.int32); nk = np.zeros(cap_jobs, np.int32)
    ru = np.zeros(cap_ranks, np.uint16); rm = np.zeros(cap_ranks, np.uint16); ro = np.zeros(cap_jobs + 1, np.int64)
    deg = np.zeros(2, np.int32)
    nj = _l.load_library().np_cm_build_jobs_cigar(alphabet_id(alphabet), ref_seq.encode(), n, _p(cg, _l.c_u32p), len(cg), int(read_len),
                                                  int(read_rc), k, min_separation, min_flank, cap_jobs, cap_ranks,
                                                  _p(f, _l.c_i32p), _p(l, _l.c_i32p), _p(c, _l.c_i32p), _p(kpos, _l.c_i32p),
                                                  _p(nk, _l.c_i32p), _p(ru, _l.c_u16p), _p(rm, _l.c_u16p), _p(ro, _l.c_i64p),
                                                  _p(deg, _l.c_i32p))
    if nj == -1:                      # NP_ERR_INVALID: a spliced / padded CIGAR, which the reference rejects (exit / assert)
        raise ValueError("np_cm_build_jobs_cigar: spliced or malformed CIGAR")
    if nj < 0:
        raise RuntimeError("np_cm_build_jobs_cigar: %d" % nj)
    w = int(ro[nj])
    return dict(first=f[:nj].copy(), last=l[:nj].copy(), n_motif=c[:nj].copy(), kpos=kpos[:2 * nj].reshape(-1, 2).copy(),
                n_kmers=nk[:nj].copy(), ranks_unmeth=ru[:w].copy(), ranks_meth=rm[:w].copy(), rank_off=ro[:nj + 1].copy(),
                deg_kpos=deg.copy())


# ---- device context ---------------------------------------------------------------------------------------------
class Context:
    """np_ctx wrapper.  Raises RuntimeError if no gfx950 device is usable (no CPU fallback)."""

    def __init__(self, device=0, indel_bias=1.0):
        self.L = _l.load_library()
        p = _l.Params()
        self.L.np_default_params(C.byref(p))
        p.hmm_indel_bias_factor = indel_bias
        self.h = self.L.np_create(device, C.byref(p))
        if not self.h:
            raise RuntimeError("np_create failed: %s" % self.L.np_last_error(None).decode())
        self.models = {}

    def close(self):
        if self.h:
            self.L.np_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.np_last_error(self.h).decode()))

    def set_option(self, name, value):
        self._chk(self.L.np_set_option(self.h, name.encode(), int(value)), "np_set_option")

    def get_stat(self, name):
        return int(self.L.np_get_stat(self.h, name.encode()))

    def info(self):
        """what np_create's hardware probe found (np_ctx_info)"""
        return self.L.np_ctx_info(self.h).decode()

    def register_model(self, model, name=None):
        lm = np.ascontiguousarray(model["level_mean"], np.float64)
        ls = np.ascontiguousarray(model["level_stdv"], np.float64)
        ll = np.ascontiguousarray(model["level_log_stdv"], np.float64)
        mid = self.L.np_register_model(self.h, int(model["k"]), len(lm), _p(lm, _l.c_f64p), _p(ls, _l.c_f64p), _p(ll, _l.c_f64p))
        if mid < 0:
            self._chk(mid, "np_register_model")
        if name:
            self.models[name] = mid
        return mid

    # -- drop-in entry points (host buffers) --------------------------------------------------------------------
    def _hmm_jobs(self, jobs):
        arr = (_l.HmmJob * len(jobs))()
        keep = []
        for i, j in enumerate(jobs):
            ev = np.ascontiguousarray(j["events"], np.float32)
            rk = np.ascontiguousarray(j["ranks"], np.uint16)
            keep.append((ev, rk))
            a = arr[i]
            a.event_mean = _p(ev, _l.c_f32p); a.n_events_total = len(ev)
            a.e_start = int(j["e_start"]); a.e_stop = int(j["e_stop"]); a.stride = int(j["stride"])
            a.kmer_rank = _p(rk, _l.c_u16p); a.n_kmers = len(rk); a.model = int(j["model"])
            a.scale = j["scale"]; a.shift = j["shift"]; a.var = j["var"]; a.events_per_base = j["events_per_base"]
            a.flags = int(j.get("flags", 0)); a.indel_bias = float(j.get("indel_bias", 0.0))
        return arr, keep

    def profile_hmm_score(self, jobs):
        """jobs: list of dicts (events, ranks, e_start, e_stop, stride, model, scale, shift, var, events_per_base, flags).
        Returns float32 scores, one per job, as profile_hmm_score would."""
        arr, keep = self._hmm_jobs(jobs)
        out = np.zeros(len(jobs), np.float32)
        self._chk(self.L.np_hmm_score_host(self.h, len(jobs), arr, _p(out, _l.c_f32p)), "np_hmm_score_host")
        return out

    def profile_hmm_score_set(self, job_sets):
        """job_sets: list of lists of jobs (sequence 0 under the nucleotide model, the others under their alphabets'
        models).  Combination as profile_hmm_score_set (src/hmm/nanopolish_profile_hmm.cpp:32-56)."""
        flat = [j for s in job_sets for j in s]
        arr, keep = self._hmm_jobs(flat)
        off = np.zeros(len(job_sets) + 1, np.int32)
        off[1:] = np.cumsum([len(s) for s in job_sets])
        out = np.zeros(len(job_sets), np.float32)
        self._chk(self.L.np_hmm_score_set_host(self.h, len(job_sets), _p(off, _l.c_i32p), arr, _p(out, _l.c_f32p)),
                  "np_hmm_score_set_host")
        return out

    def profile_hmm_align(self, jobs):
        """Returns a list of (event_idx, kmer_idx, l_fm, state) arrays, one tuple per job (HMMAlignmentState fields)."""
        arr, keep = self._hmm_jobs(jobs)
        cap = sum(abs(int(j["e_stop"]) - int(j["e_start"])) + 1 + len(j["ranks"]) + 1 for j in jobs)
        st = (_l.HmmState * cap)()
        off = np.zeros(len(jobs) + 1, np.int64)
        self._chk(self.L.np_hmm_align_host(self.h, len(jobs), arr, st, cap, _p(off, _l.c_i64p)), "np_hmm_align_host")
        raw = np.frombuffer(st, dtype=np.dtype([("event_idx", "<u4"), ("kmer_idx", "<u4"), ("l_fm", "<f8"), ("state", "u1"),
                                                ("pad", "u1", 7)]))
        res = []
        for i in range(len(jobs)):
            r = raw[off[i]:off[i + 1]]
            res.append((r["event_idx"].copy(), r["kmer_idx"].copy(), r["l_fm"].copy(), r["state"].copy()))
        return res

    def adaptive_banded_simple_event_align(self, reads):
        """reads: list of dicts (events, ranks [nucleotide], model, scale, shift, var=1).  Returns a list of (n,2) int32
        arrays of AlignedPair{ref_pos, read_pos}; an empty array is the reference's empty vector (QC failure)."""
        n = len(reads)
        arr = (_l.AlignJob * n)()
        keep = []
        cap = 0
        for i, r in enumerate(reads):
            ev = np.ascontiguousarray(r["events"], np.float32)
            rk = np.ascontiguousarray(r["ranks"], np.uint16)
            keep.append((ev, rk))
            a = arr[i]
            a.event_mean = _p(ev, _l.c_f32p); a.n_events = len(ev); a.kmer_rank = _p(rk, _l.c_u16p); a.n_kmers = len(rk)
            a.model = int(r["model"]); a.scale = r["scale"]; a.shift = r["shift"]; a.var = r.get("var", 1.0)
            cap += len(ev) + len(rk) + 2
        pairs = np.zeros((cap, 2), np.int32)
        off = np.zeros(n + 1, np.int64)
        self._chk(self.L.np_event_align_host(self.h, n, arr, pairs.ctypes.data_as(C.POINTER(_l.Pair)), cap, _p(off, _l.c_i64p)),
                  "np_event_align_host")
        return [pairs[off[i]:off[i + 1]].copy() for i in range(n)]

    def detect_events(self, raws, rna=False, params=None):
        """detect_events (src/thirdparty/scrappie/event_detection.c:268-319) on whole raw tables, as load_from_raw calls it.
        raws: list of float32 arrays (pA).  Returns a list of dicts(start u32, length f32, mean f32, stdv f32)."""
        n = len(raws)
        keep = [np.ascontiguousarray(r, np.float32) for r in raws]
        ptrs = (_l.c_f32p * n)(*[_p(r, _l.c_f32p) for r in keep])
        ns = np.array([len(r) for r in keep], np.uint32)
        prm = _l.DetectorParam()
        self.L.np_event_detection_params(C.byref(prm), int(bool(rna)))
        if params:
            prm.window_length1, prm.window_length2 = params["w1"], params["w2"]
            prm.threshold1, prm.threshold2, prm.peak_height = params["t1"], params["t2"], params["peak_height"]
        cap = int((ns.astype(np.int64) // 2 + 2).sum())
        st = np.zeros(cap, np.uint32); ln = np.zeros(cap, np.float32); mn = np.zeros(cap, np.float32); sd = np.zeros(cap, np.float32)
        off = np.zeros(n + 1, np.int64)
        self._chk(self.L.np_detect_events_host(self.h, n, ptrs, ns.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(prm),
                                               st.ctypes.data_as(C.POINTER(C.c_uint32)), _p(ln, _l.c_f32p), _p(mn, _l.c_f32p),
                                               _p(sd, _l.c_f32p), cap, _p(off, _l.c_i64p)), "np_detect_events_host")
        return [dict(start=st[off[i]:off[i + 1]].copy(), length=ln[off[i]:off[i + 1]].copy(), mean=mn[off[i]:off[i + 1]].copy(),
                     stdv=sd[off[i]:off[i + 1]].copy()) for i in range(n)]

    # -- timing -----------------------------------------------------------------------------------------------------
    def sync(self, stream=None):
        self._chk(self.L.np_sync(self.h, stream), "np_sync")

    def kernel_time(self, which, reset=False):
        ms = C.c_double(); n = C.c_int64()
        self._chk(self.L.np_kernel_time(self.h, which, C.byref(ms), C.byref(n), int(reset)), "np_kernel_time")
        return ms.value, n.value
