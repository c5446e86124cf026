"""ctypes view of include/np_hmm.h.  Fails loudly if libnp_hip.so is missing."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("NP_HIP_LIB") or os.path.join(_HERE, "libnp_hip.so")   # NP_HIP_LIB: A/B builds while tuning
_lib = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u16p = C.POINTER(C.c_uint16)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)

# exported symbols of include/np_hmm.h (tests check that every one resolves)
SYMBOLS = [
    "np_default_params", "np_create", "np_destroy", "np_last_error", "np_version", "np_ctx_info", "np_set_option", "np_get_stat", "np_register_model", "np_update_model", "np_site_table_dev", "np_hmm_score_set_combine_dev", "np_dev_alloc", "np_dev_free", "np_host_alloc", "np_host_free", "np_stream_create", "np_stream_destroy", "np_event_create", "np_event_destroy", "np_event_record", "np_stream_wait_event", "np_event_sync", "np_event_query", "np_reverse_events_dev", "np_copy_to_device", "np_copy_to_host", "np_memset_dev",
    "np_alphabet_id", "np_alphabet_size", "np_kmer_rank", "np_reverse_complement", "np_methylate", "np_unmethylate",
    "np_is_motif_match", "np_sequence_kmer_ranks", "np_calculate_transitions", "np_estimate_scalings_mom",
    "np_scan_motif_groups", "np_cm_build_jobs_identity", "np_fill_read_host", "np_hmm_score_host", "np_hmm_score_set_host", "np_hmm_align_host", "np_event_align_host",
    "np_event_align_dev", "np_hmm_score_dev", "np_resolve_jobs_dev", "np_calibrate_resolve_dev", "np_event_detection_params", "np_detect_events_dev", "np_detect_events_host", "np_mom_fill_dev", "np_cm_build_jobs_identity_dev", "np_cm_build_jobs_cigar_dev", "np_cm_discard_degenerate_dev", "np_set_job_layout", "np_cigar_aligned_bases", "np_cm_build_jobs_cigar", "np_eventalign_dev", "np_aligner_constants", "np_restated_log_exp", "np_selftest_libm", "np_site_table_genome_dev", "np_site_table_genome_indexed_dev", "np_genome_site_index_dev", "np_adc_to_pa_dev", "np_adc_to_pa_checked_dev", "np_detect_events_checked_dev", "np_detect_events_adc_dev", "np_sync", "np_last_kernel_ms", "np_kernel_time", "np_selftest_division", "np_selftest_division_small", "np_selftest_tstat_ratio",
]


class Params(C.Structure):
    _fields_ = [("hmm_indel_bias_factor", C.c_double), ("min_average_log_emission", C.c_double),
                ("max_gap_threshold", C.c_int32), ("reserved", C.c_int32)]


class HmmJob(C.Structure):
    _fields_ = [("event_mean", c_f32p), ("n_events_total", C.c_uint32), ("e_start", C.c_uint32), ("e_stop", C.c_uint32),
                ("stride", C.c_int32), ("kmer_rank", c_u16p), ("n_kmers", C.c_uint32), ("model", C.c_int32),
                ("scale", C.c_double), ("shift", C.c_double), ("var", C.c_double), ("events_per_base", C.c_double),
                ("flags", C.c_uint32), ("reserved", C.c_uint32), ("indel_bias", C.c_double)]


class HmmState(C.Structure):
    _fields_ = [("event_idx", C.c_uint32), ("kmer_idx", C.c_uint32), ("l_fm", C.c_double), ("state", C.c_char),
                ("pad", C.c_char * 7)]


class Pair(C.Structure):
    _fields_ = [("ref_pos", C.c_int32), ("read_pos", C.c_int32)]


class AlignJob(C.Structure):
    _fields_ = [("event_mean", c_f32p), ("n_events", C.c_uint32), ("kmer_rank", c_u16p), ("n_kmers", C.c_uint32),
                ("model", C.c_int32), ("scale", C.c_double), ("shift", C.c_double), ("var", C.c_double)]


class ReadDev(C.Structure):
    _fields_ = [("scale", C.c_double), ("shift", C.c_double), ("var", C.c_double), ("log_var", C.c_double),
                ("lp_skip", C.c_double), ("lp_stay", C.c_double), ("lp_step", C.c_double), ("lp_trim", C.c_double),
                ("event_off", C.c_int64), ("rank_off", C.c_int64), ("n_events", C.c_uint32), ("n_kmers", C.c_uint32),
                ("trans", C.c_float * 10), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class DetectorParam(C.Structure):
    _fields_ = [("window_length1", C.c_uint32), ("window_length2", C.c_uint32), ("threshold1", C.c_float),
                ("threshold2", C.c_float), ("peak_height", C.c_float)]


class HmmJobDev(C.Structure):
    _fields_ = [("rank_off", C.c_int64), ("n_kmers", C.c_uint32), ("read", C.c_uint32), ("e_start", C.c_uint32),
                ("e_stop", C.c_uint32), ("stride", C.c_int32), ("flags", C.c_uint32)]


def library_path():
    return _LIB


def build_library(verbose=False):
    """hipcc --offload-arch=gfx950 build of every kernel (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return _LIB


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise RuntimeError(
            "nanopolish_amd: %s is missing -- the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C nanopolish_amd/csrc`).  There is no CPU fallback." % _LIB)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.  Import torch first (when it is
    # installed) so that libnp_hip.so binds to the runtime torch uses, whatever order the caller imports things in;
    # without torch the library's rpath (/opt/rocm/lib) provides the runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB)
    vp = C.c_void_p
    L.np_create.restype = vp
    L.np_create.argtypes = [C.c_int, C.POINTER(Params)]
    L.np_destroy.argtypes = [vp]
    L.np_last_error.restype = C.c_char_p
    L.np_last_error.argtypes = [vp]
    L.np_version.restype = C.c_char_p
    L.np_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.np_get_stat.argtypes = [vp, C.c_char_p]; L.np_get_stat.restype = C.c_int64
    L.np_ctx_info.argtypes = [vp]; L.np_ctx_info.restype = C.c_char_p
    L.np_register_model.argtypes = [vp, C.c_int, C.c_int, c_f64p, c_f64p, c_f64p]
    L.np_update_model.argtypes = [vp, C.c_int, C.c_int, c_f64p, c_f64p, c_f64p]
    L.np_alphabet_id.argtypes = [C.c_char_p]
    L.np_alphabet_size.restype = C.c_uint32
    L.np_kmer_rank.restype = C.c_uint32
    L.np_kmer_rank.argtypes = [C.c_int, C.c_char_p, C.c_uint32]
    for f in ("np_reverse_complement", "np_methylate", "np_unmethylate"):
        getattr(L, f).argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p]
    L.np_is_motif_match.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_size_t]
    L.np_sequence_kmer_ranks.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, c_u16p]
    L.np_calculate_transitions.argtypes = [C.c_double, C.c_double, c_f32p]
    L.np_calculate_transitions.restype = None
    L.np_estimate_scalings_mom.argtypes = [c_f64p, c_u16p, C.c_uint32, c_f32p, C.c_uint32, c_f64p, c_f64p]
    L.np_estimate_scalings_mom.restype = None
    L.np_scan_motif_groups.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_int, c_i32p, c_i32p, c_i32p, C.c_int]
    L.np_cm_build_jobs_identity.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                            C.c_int64, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_u16p, c_u16p, c_i64p]
    L.np_fill_read_host.argtypes = [C.POINTER(ReadDev), C.c_double, C.c_double, C.c_double, C.c_int64, C.c_uint32,
                                    C.c_int64, C.c_uint32]
    L.np_fill_read_host.restype = None
    L.np_hmm_score_host.argtypes = [vp, C.c_int, C.POINTER(HmmJob), c_f32p]
    L.np_hmm_score_set_host.argtypes = [vp, C.c_int, c_i32p, C.POINTER(HmmJob), c_f32p]
    L.np_hmm_align_host.argtypes = [vp, C.c_int, C.POINTER(HmmJob), C.POINTER(HmmState), C.c_int64, c_i64p]
    L.np_event_align_host.argtypes = [vp, C.c_int, C.POINTER(AlignJob), C.POINTER(Pair), C.c_int64, c_i64p]
    L.np_event_align_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int64, vp, vp, vp, vp]
    L.np_hmm_score_dev.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, C.c_int, vp]
    L.np_resolve_jobs_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int64, vp, vp]
    L.np_selftest_division.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.np_selftest_tstat_ratio.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.np_selftest_division_small.argtypes = [vp, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.np_set_job_layout.argtypes = [vp, C.c_int, vp, vp, C.c_int64]
    L.np_calibrate_resolve_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int64, vp, vp]
    L.np_event_detection_params.argtypes = [C.POINTER(DetectorParam), C.c_int]
    L.np_event_detection_params.restype = None
    L.np_detect_events_dev.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int64, C.POINTER(DetectorParam), vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    L.np_detect_events_checked_dev.argtypes = L.np_detect_events_dev.argtypes + [vp]
    L.np_detect_events_adc_dev.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int64, vp, vp, vp, C.POINTER(DetectorParam), vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    L.np_detect_events_host.argtypes = [vp, C.c_int, C.POINTER(c_f32p), C.POINTER(C.c_uint32), C.POINTER(DetectorParam),
                                        C.POINTER(C.c_uint32), c_f32p, c_f32p, c_f32p, C.c_int64, c_i64p]
    L.np_mom_fill_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    L.np_reverse_events_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.np_event_query.argtypes = [vp, vp]
    L.np_cm_build_jobs_identity_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_uint32, C.c_int, C.c_int, vp, C.c_int64, vp, vp, vp,
                                                vp, vp, vp, vp, vp]
    L.np_cm_build_jobs_cigar_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int64, vp, vp, C.c_int, C.c_uint32, C.c_int, C.c_int, vp,
                                             C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.np_eventalign_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int64, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
    L.np_cm_discard_degenerate_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int64, vp]
    L.np_cigar_aligned_bases.argtypes = [c_u32p, C.c_int, C.c_int, c_i32p, c_i32p, C.c_int]
    L.np_cm_build_jobs_cigar.argtypes = [C.c_int, C.c_char_p, C.c_size_t, c_u32p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                         C.c_int, C.c_int64, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_u16p, c_u16p, c_i64p, c_i32p]
    L.np_aligner_constants.argtypes = [C.c_uint32, C.c_uint32, c_f64p]
    L.np_aligner_constants.restype = None
    L.np_restated_log_exp.argtypes = [c_f64p, C.c_size_t, c_f64p, c_f64p]
    L.np_restated_log_exp.restype = None
    L.np_selftest_libm.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.np_sync.argtypes = [vp, vp]
    L.np_adc_to_pa_dev.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int64, vp, vp, vp]
    L.np_adc_to_pa_checked_dev.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int64, vp, vp, vp, vp]
    L.np_site_table_dev.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, C.c_double, C.c_int64, vp]
    L.np_site_table_genome_dev.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int64, vp, vp]
    L.np_site_table_genome_indexed_dev.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int64, vp, vp, vp, vp]
    L.np_genome_site_index_dev.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp, vp, vp]
    L.np_hmm_score_set_combine_dev.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp]
    L.np_last_kernel_ms.argtypes = [vp, C.c_int, c_f32p]
    L.np_kernel_time.argtypes = [vp, C.c_int, c_f64p, c_i64p, C.c_int]
    _lib = L
    return L
