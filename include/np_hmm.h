/* include/np_hmm.h -- C ABI of the MI355X-native nanopolish signal-HMM hot path.
 *
 * nanopolish (v0.14.0) has no plugin/FFI layer: its "boundary" for this path is the C++ free-function
 * API
 *     float profile_hmm_score(const HMMInputSequence&, const HMMInputData&, uint32_t flags)      src/hmm/nanopolish_profile_hmm.h:24
 *     float profile_hmm_score(const HMMInputSequence&, const std::vector<HMMInputData>&, uint32) src/hmm/nanopolish_profile_hmm.h:25
 *     float profile_hmm_score_set(const std::vector<HMMInputSequence>&, const HMMInputData&, u32) src/hmm/nanopolish_profile_hmm.h:28
 *     std::vector<HMMAlignmentState> profile_hmm_align(const HMMInputSequence&, const HMMInputData&, u32)  :31
 *     std::vector<AlignedPair> adaptive_banded_simple_event_align(SquiggleRead&, const PoreModel&, const std::string&)
 *                                                                                                src/nanopolish_raw_loader.h:22-24
 * This header is what a thin shim behind those signatures binds (nanopolish_amd/csrc/np_dropin.cpp is that
 * shim; INTEGRATION.md shows where it plugs into the reference).  Plain pointers and sizes only; no C++ or
 * torch types.  Every function returns NP_OK (0) or a negative error code and never throws.
 *
 * Two flavours of every entry point:
 *   *_host  : caller passes host buffers (what the per-call C++ API hands over); the library packs,
 *             uploads, launches, downloads.  Synchronous, thread-safe (internal lock).
 *   *_dev   : caller passes device-resident SoA buffers (HBM) and a stream; nothing is copied, the call
 *             only enqueues kernels.  This is the throughput path fed at the BamProcessor batch boundary
 *             (src/common/nanopolish_bam_processor.cpp:90-119).
 */
#ifndef NP_HMM_H
#define NP_HMM_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NP_OK 0
#define NP_ERR_INVALID (-1)     /* bad argument */
#define NP_ERR_DEVICE (-2)      /* HIP error (see np_last_error) */
#define NP_ERR_NOMEM (-3)
#define NP_ERR_UNSUPPORTED (-4) /* problem outside what the kernels cover (e.g. > NP_MAX_KMERS k-mers) */

/* HMMAlignmentFlags, src/hmm/nanopolish_profile_hmm.h:34-38 */
#define NP_HAF_ALLOW_PRE_CLIP 1u
#define NP_HAF_ALLOW_POST_CLIP 2u
/* device work items only: set by np_resolve_jobs_dev on items the reference would skip (score = NaN) */
#define NP_JOB_SKIP 0x80000000u

#define NP_MAX_KMERS 1024      /* k-mers per profile_hmm_* call (reference callers stay <= 260) */
#define NP_MAX_WINDOW_EVENTS (1u << 20)  /* events per profile_hmm_* call: the length of the universal clip-flank table (np_create) */
#define NP_ALN_BANDWIDTH 100   /* ALN_BANDWIDTH, src/nanopolish_raw_loader.cpp:72 */

typedef struct np_ctx np_ctx;

/* Behaviour knobs that are globals / compile-time constants in the reference. */
typedef struct np_params {
    double hmm_indel_bias_factor;   /* src/hmm/nanopolish_profile_hmm_r9.cpp:19 (1.0; variants sets .8/.9) */
    double min_average_log_emission;/* src/nanopolish_raw_loader.cpp:92  (-5.0) */
    int32_t max_gap_threshold;      /* src/nanopolish_raw_loader.cpp:93  (50)   */
    int32_t reserved;
} np_params;

void np_default_params(np_params* p);

/* ---- context ------------------------------------------------------------------------------------- */
/* device: HIP device ordinal.  Fails (returns NULL, see np_last_error(NULL)) if no gfx950 device/HIP
 * runtime is usable: there is NO CPU fallback in this library. */
np_ctx* np_create(int device, const np_params* params);
void    np_destroy(np_ctx* ctx);
const char* np_last_error(const np_ctx* ctx);
const char* np_version(void);
/* What np_create's hardware probe found, one line: the clamp-free fast paths rest on two hardware rules -- an LDS read past the
 * workgroup's allocation returns 0 (the forward kernel's log-sum lookup), a range-checked buffer access outside its descriptor
 * reads 0 / is dropped (the event aligner's and the chain kernel's prefetches) -- and on every forward kernel owning exactly
 * one LDS object.  If the LDS rule or the size check fails the context scores with the clamped lookup (same results, ~5 %
 * slower); if the buffer rule fails np_create fails.  NP_VERBOSE=1 prints the line at np_create; NP_LSE_CLAMP=1 forces the
 * clamped lookup. */
const char* np_ctx_info(const np_ctx* ctx);

/* Tuning / test knobs (defaults are what bench.py measures): "align_blocks_per_cu", "hmm_blocks_per_cu" (persistent grid
 * sizes), "align_lpt" (1: the event aligner takes the batch's reads longest first; 0: in index order),
 * "stream_switch_wait" (1: a call on another stream than the context's previous call waits for that stream's tail; 0: the caller orders
 * the streams it uses with one context by its own events), "ed_warmup" (samples each segment of the parallel peak walk starts early; 0 forces every segment through the
 * repair path -- results never depend on it), "ed_ratio_exact" (1: the fused walk evaluates the t-statistic's (float)(|dm| / sqrt(cvw)) by the exact
 * square root and division for every value; 0, the default when np_create's probe passes: by a once-refined v_rsq_f64 wherever that provably
 * rounds to the same float -- events never depend on it), "ea_rows_cap" (events per eventalign segment the chain kernel's scratch holds;
 * a longer segment ends its read with NP_EA_OVERFLOW), "ea_waves_per_cu" (persistent grid of the chain kernel), "lse_oor" (0: the forward kernel clamps its log-sum table index;
 * 1: it relies on the LDS out-of-range rule, the default when np_create's probe passes and refused (NP_ERR_UNSUPPORTED) when it did not --
 * scores never depend on it), "hmm_prio" (wave priority of the forward kernels, 0 ... 2),
 * "recal_shape" (np_calibrate_resolve_dev's workgroup shape, 3: the default -- sixteen waves x four reads, 32-k-mer chunks --, 0 ... 2: the shapes it was measured against; results never depend on it), "small_batch_path"
 * (1: np_hmm_score_host sends a small batch as one pinned blob; 0: the general path -- the tests compare the two).
 * Environment, read at np_create (each the option of the same name): NP_ALIGN_BLOCKS_PER_CU, NP_HMM_BLOCKS_PER_CU, NP_ALIGN_LPT, NP_ED_WARMUP, NP_ED_RATIO_EXACT,
 * NP_EA_WAVES_PER_CU, NP_RECAL_SHAPE; NP_EA_WALK_PRIO=1 runs the chain kernel's back-track at wave priority 3 (measured slower,
 * profiles/r05_soak.md); NP_LSE_CLAMP, NP_HOST_CONSTANTS, NP_VERBOSE as described at np_ctx_info / in INTEGRATION.md. */
int np_set_option(np_ctx* ctx, const char* name, int64_t value);
/* Read-only facts about the context (-1: unknown name): "align_blocks" / "align_scratch_bytes" (persistent grid and per-wave scratch
 * of the most recent event-align launch: the grid shrinks under a 48 GB scratch budget when a batch holds ultra-long reads),
 * "align_blocks_max", "lse_oor" (1: the clamp-free log-sum lookup is in use, see np_ctx_info), "lse_probe_ok" (1: np_create's probe found the LDS rule to hold), "n_cu", "ed_serial_reads" / "ed_refused_reads" (most recent event-detection
 * call: reads that took the serial prefix-sum path, reads refused with NP_ED_INEXACT; waits for the call), "ea_lattice_cells" /
 * "ea_lattice_rows" / "ea_lattice_kmers" (sum over the segments of the most recent np_eventalign_dev call of the reference's
 * lattice size (e + 1) x 3 (n + 2), of e and of n; waits for the call), "ea_cycles_geometry" / "ea_cycles_fill" /
 * "ea_cycles_backtrack" (two-read chain kernel: shader cycles, summed over the waves, spent finding segments, sweeping them, and
 * walking back + emitting). */
int64_t np_get_stat(np_ctx* ctx, const char* name);

/* Upload a pore model (PoreModel::states, src/pore_model/nanopolish_poremodel.h:20-67,107): the three
 * per-state doubles the path reads.  Returns a model id >= 0, or a negative error. */
int np_register_model(np_ctx* ctx, int k, int n_states,
                      const double* level_mean, const double* level_stdv, const double* level_log_stdv);
/* Replace the parameters of a registered model in place (same n_states).  The reference overwrites a registered PoreModel
 * at the same address (PoreModelSet::register_model, src/pore_model/nanopolish_pore_model_set.cpp:70; methyltrain does so
 * every training round), so a binding that caches ids by PoreModel* must refresh the device copy when the content changes
 * (np_dropin.cpp does, by a content hash).  Waits for the work enqueued on the context so far. */
int np_update_model(np_ctx* ctx, int model, int n_states,
                    const double* level_mean, const double* level_stdv, const double* level_log_stdv);

/* ---- host-side helpers (pure CPU, no device): alphabets & transitions ------------------------------ */
/* Alphabets of src/common/nanopolish_alphabet.{h,cpp}: "nucleotide","cpg","gpc","dam","dcm","u_to_t_rna" */
int      np_alphabet_id(const char* name);
uint32_t np_alphabet_size(int alphabet);
uint32_t np_kmer_rank(int alphabet, const char* kmer, uint32_t k);                 /* Alphabet::kmer_rank          */
int      np_reverse_complement(int alphabet, const char* in, size_t n, char* out); /* Alphabet::reverse_complement */
int      np_methylate(int alphabet, const char* in, size_t n, char* out);          /* Alphabet::methylate          */
int      np_unmethylate(int alphabet, const char* in, size_t n, char* out);        /* Alphabet::unmethylate        */
int      np_is_motif_match(int alphabet, const char* str, size_t n, size_t i);     /* Alphabet::is_motif_match     */
/* HMMInputSequence::get_kmer_rank for every k-mer (src/hmm/nanopolish_hmm_input_sequence.h:76-91).
 * rc_seq may be NULL (then reverse_complement(seq) is used, as the 1- and 2-argument constructors do). */
int      np_sequence_kmer_ranks(int alphabet, const char* seq, const char* rc_seq, size_t n, uint32_t k,
                                int do_rc, uint16_t* out_ranks);
/* calculate_transitions (src/hmm/nanopolish_profile_hmm_r9.inl:17-76) with host libm, exactly as the reference */
void     np_calculate_transitions(double events_per_base, double indel_bias, float out[10]);
/* estimate_scalings_using_mom (src/nanopolish_raw_loader.cpp:17-60) */
void     np_estimate_scalings_mom(const double* model_level_mean, const uint16_t* kmer_ranks, uint32_t n_kmers,
                                  const float* event_mean, uint32_t n_events, double* shift, double* scale);
/* motif scan + grouping of calculate_methylation_for_read (src/basemods/nanopolish_basemods.cpp:298-320) */
int      np_scan_motif_groups(int alphabet, const char* ref_seq, size_t n, int min_separation,
                              int32_t* first_site, int32_t* last_site, int32_t* n_motif, int cap);

/* Work items of calculate_methylation_for_read (src/basemods/nanopolish_basemods.cpp:298-378) for an identity-aligned
 * read (bench/test layout): groups, windows, boundary rule, methylated/unmethylated k-mer ranks, and the read-strand
 * k-mer positions (kpos) whose closest events bound each window.  Returns #jobs or a negative error. */
int      np_cm_build_jobs_identity(int alphabet, const char* ref_seq, size_t n, int read_rc, uint32_t k,
                                   int min_separation, int min_flank, int cap_jobs, int64_t cap_ranks,
                                   int32_t* first_site, int32_t* last_site, int32_t* n_motif,
                                   int32_t* kpos, int32_t* job_n_kmers,
                                   uint16_t* ranks_unmeth, uint16_t* ranks_meth, int64_t* rank_off);

/* ---- per-call ("drop-in") entry points, host buffers ------------------------------------------------- */
/* One profile_hmm_score / profile_hmm_align problem, flattened from (HMMInputSequence, HMMInputData):
 *   event_mean   read->events[strand][*].mean (drift == 0 on the R9 path)     src/nanopolish_squiggle_read.h:149-154
 *   kmer_rank    HMMInputSequence::get_kmer_rank(i, k, data.rc), i = 0..n_kmers-1
 *   e_start/e_stop/stride   HMMInputData::event_start_idx/event_stop_idx/event_stride
 *   scale/shift/var         read->scalings[strand]                           src/nanopolish_squiggle_read.h:63-93
 *   events_per_base         read->events_per_base[strand]
 */
typedef struct np_hmm_job {
    const float*    event_mean;      /* whole-read event means; indexed by event idx */
    uint32_t        n_events_total;
    uint32_t        e_start, e_stop;
    int32_t         stride;          /* +1 / -1 */
    const uint16_t* kmer_rank;
    uint32_t        n_kmers;
    int32_t         model;           /* id from np_register_model */
    double          scale, shift, var;
    double          events_per_base;
    uint32_t        flags;           /* NP_HAF_* */
    uint32_t        reserved;
    double          indel_bias;      /* hmm_indel_bias_factor for this call (src/hmm/nanopolish_profile_hmm_r9.cpp:19);
                                        0 = the context's np_params value */
} np_hmm_job;

/* HMMAlignmentState, src/common/nanopolish_common.h:65-73 (l_posterior / log_transition_probability are
 * always -INFINITY in the reference and are not transported) */
typedef struct np_hmm_state {
    uint32_t event_idx;
    uint32_t kmer_idx;
    double   l_fm;
    char     state;     /* 'K','B','M' */
    char     pad[7];
} np_hmm_state;

/* AlignedPair, src/alignment/nanopolish_anchor.h:18-22 */
typedef struct np_pair { int32_t ref_pos; int32_t read_pos; } np_pair;

typedef struct np_align_job {
    const float*    event_mean;   /* read.events[0][*].mean */
    uint32_t        n_events;
    const uint16_t* kmer_rank;    /* nucleotide k-mer ranks of `sequence` */
    uint32_t        n_kmers;
    int32_t         model;
    double          scale, shift; /* read.scalings[0] (var = 1, drift = 0 at the call site, raw_loader.cpp:52) */
    double          var;
} np_align_job;

int np_hmm_score_host(np_ctx* ctx, int n_jobs, const np_hmm_job* jobs, float* out_scores);
/* profile_hmm_score_set: set q is jobs[set_off[q] .. set_off[q+1]) (sequence 0 first); one combined score per set. */
int np_hmm_score_set_host(np_ctx* ctx, int n_sets, const int32_t* set_off, const np_hmm_job* jobs, float* out_scores);
/* out_states: concatenated; out_off[n_jobs+1] offsets; cap = capacity of out_states.
 * A job for which the reference would hit an assert reports count 0. */
int np_hmm_align_host(np_ctx* ctx, int n_jobs, const np_hmm_job* jobs,
                      np_hmm_state* out_states, int64_t cap, int64_t* out_off);
/* out_pairs: concatenated; out_off[n_jobs+1]; an empty range == the reference's empty vector (QC failure). */
int np_event_align_host(np_ctx* ctx, int n_jobs, const np_align_job* jobs,
                        np_pair* out_pairs, int64_t cap, int64_t* out_off);

/* ---- batched device-resident entry points ------------------------------------------------------------- */
/* All pointers below are DEVICE pointers unless stated otherwise; `stream` is a hipStream_t (0 = the
 * context's own stream).  The calls only enqueue work.
 * ONE stream at a time per context: the work queues, class counters and scratch buffers of a context are shared by all of
 * its calls, so a context serialises them.  When a call arrives on a different stream than the previous one (including a
 * *_host call, which uses the context's own stream), the library makes the new stream wait for everything enqueued on the
 * old one (an event wait on the device, nothing blocks the host).  For concurrent pipelines create one context per stream.
 * Scratch grows on demand: the first call at a larger batch size reallocates (a device-wide synchronisation); steady-state
 * calls do not. */

/* Per-read record (device).  Filled by np_fill_read_host() on the host, then uploaded by the caller. */
typedef struct np_read_dev {
    double  scale, shift, var, log_var;  /* SquiggleScalings used for emissions                      */
    double  lp_skip, lp_stay, lp_step, lp_trim; /* aligner constants, src/nanopolish_raw_loader.cpp:99-108 */
    int64_t event_off;                   /* offset of this read's events in the batch event array    */
    int64_t rank_off;                    /* offset of this read's nucleotide k-mer ranks             */
    uint32_t n_events;
    uint32_t n_kmers;
    float   trans[10];                   /* calculate_transitions order (r9.h:75-95), HMM scoring    */
    uint32_t flags;
    uint32_t reserved;
} np_read_dev;

/* Host helper: fills the aligner constants with host libm exactly as the reference does. */
void np_fill_read_host(np_read_dev* r, double shift, double scale, double var,
                       int64_t event_off, uint32_t n_events, int64_t rank_off, uint32_t n_kmers);

/* One HMM work item (device). */
typedef struct np_hmm_job_dev {
    int64_t  rank_off;     /* into the job k-mer rank array (uint16) */
    uint32_t n_kmers;
    uint32_t read;         /* index into np_read_dev[] */
    uint32_t e_start, e_stop;
    int32_t  stride;
    uint32_t flags;
} np_hmm_job_dev;

/* Event alignment of a batch of reads (kernel A).
 *   pairs_out : np_pair[pair_off[n_reads]] ; pair_off (device int64[n_reads+1]) must give each read room for
 *               n_events + n_kmers + 2 pairs.  The alignment of read r is written ascending into
 *               pairs_out[pair_off[r] + pair_begin[r] .. pair_off[r+1])  -- right-aligned --
 *   pair_begin (int32[n_reads]) and n_pairs (int32[n_reads], 0 == QC failure) are outputs.
 *   max_bands : max over the batch of n_events + n_kmers + 2 (sizes the per-wave trace scratch). */
int np_event_align_dev(np_ctx* ctx, void* stream, int n_reads, const np_read_dev* reads,
                       const float* event_mean, const uint16_t* kmer_rank, int model, int64_t max_bands,
                       const int64_t* pair_off, np_pair* pairs_out, int32_t* pair_begin, int32_t* n_pairs);

/* Forward scores of a batch of HMM work items (kernel B).
 *   order (host pointer, may be NULL): nothing to provide; the library bins jobs by size on the device. */
int np_hmm_score_dev(np_ctx* ctx, void* stream, int64_t n_jobs, const np_hmm_job_dev* jobs,
                     const np_read_dev* reads, const float* event_mean, const uint16_t* job_kmer_rank,
                     int model, float* out_scores);

/* profile_hmm_score_set on the device (src/hmm/nanopolish_profile_hmm.cpp:32-56).  The member sequences of a set are scored
 * under different pore models (sequence 0 under the base model, the others under their alphabets' models), and one
 * np_hmm_score_dev call binds one model: score the members model by model with np_hmm_score_dev (any layout), then combine
 *     out[q] = (+)_{t in set q} (score_t - log n_q)      (add_logs = p7_FLogsum on float casts, accumulated in double)
 * here.  Set q's members are member_scores[member_idx[set_off[q] .. set_off[q+1])] in sequence order (member_idx NULL: the
 * scores are already laid out set by set).  set_off: int64[n_sets+1].  An empty set yields -inf. */
int np_hmm_score_set_combine_dev(np_ctx* ctx, void* stream, int64_t n_sets, const int64_t* set_off, const int64_t* member_idx,
                                 const float* member_scores, float* out_scores);

/* f4 on the device: the per-site table of a batch of call-methylation results, i.e. what the reference's TSV writer
 * (src/nanopolish_call_methylation.cpp:531-550, "%.2lf" of sum_ll_m - sum_ll_u) followed by
 * scripts/calculate_methylation_frequency.py:16-23,41-49 (groups not split) accumulates:
 *     table[row] += (1, n_motif, n_motif if llr > 0)    for every scored group with |llr| >= call_threshold * n_motif,
 * llr = the group's log-likelihood ratio after the text round trip (correctly rounded to two decimals, ties to even, as
 * printf does).  row = first_site[g] (+ read_base[jobs[2g].read] when read_base is given: the record's reference offset).
 *   scores      : 2 floats per group (unmethylated, methylated), NaN = group skipped (as np_hmm_score_dev leaves them)
 *   table       : int32[n_pos][3] = (num_reads, called_sites, called_sites_methylated), ACCUMULATED into (zero it first);
 *                 this is the payload of the job's only inter-GPU exchange, one all-reduce(sum). */
int np_site_table_dev(np_ctx* ctx, void* stream, int64_t n_groups, const float* scores, const int32_t* first_site,
                      const int32_t* n_motif, const np_hmm_job_dev* jobs, const int64_t* read_base, double call_threshold,
                      int64_t n_pos, int32_t* table);

/* The same aggregation for reads that overlap on a genome, keyed as the reference keys a site: (contig, start, end) of the group
 * (src/nanopolish_call_methylation.cpp:532-550; scripts/calculate_methylation_frequency.py:16-23: `key = (c, start, end)`).  A read's groups are the
 * genome's clusters of motif sites (gaps <= min_separation, src/basemods/nanopolish_basemods.cpp:306-320) cut by the read's segment, so a read that ends
 * or starts inside a cluster reports a group with another end or start -- a key of its own in the script.
 *   first_site / last_site : per group, relative to the read's reference segment (what np_cm_build_jobs_cigar_dev writes)
 *   read_base   : int64[n_reads], genome offset of every read's segment (np_cm_build_jobs_cigar_dev's ref_begin); jobs: the read of group g
 *   genome / contig_off / n_contigs : the resident contigs (bytes, concatenated) and int64[n_contigs + 1] offsets: clusters do not cross contigs
 *   alphabet    : 1 cpg, 2 gpc, 3 dam, 4 dcm (np_cm_build_jobs_*_dev's numbering);  min_separation: the builder's (10)
 *   table       : int32[n_pos][6], ACCUMULATED into (zero it first).  Columns 0-2 = (num_reads, called_sites, called_sites_methylated) of the key
 *                 (start = row, end = the end of start's cluster); columns 3-5 = the same for the key (start = the start of end's cluster, end = row)
 *                 with an end BEFORE the cluster's.  Every key a read longer than one cluster can produce is one of the two.
 *   n_overflow  : uint64, device, ACCUMULATED: groups cut on both sides (left out of the table).
 * The table is the payload of the job's only inter-GPU exchange, one all-reduce(sum). */
int np_site_table_genome_dev(np_ctx* ctx, void* stream, int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* last_site,
                             const int32_t* n_motif, const np_hmm_job_dev* jobs, const int64_t* read_base, const char* genome,
                             const int64_t* contig_off, int n_contigs, int alphabet, int min_separation, double call_threshold, int64_t n_pos,
                             int32_t* table, uint64_t* n_overflow);

/* The motif sites of the resident genome as a rank structure, for a genome-keyed table with one row per SITE instead of one per base: both keys of
 * np_site_table_genome_dev are positions of motif sites (a group's first and last), so the table and the all-reduce payload shrink from
 * 24 bytes x bases to 24 bytes x sites (5 Mb at the bench's site density: 120 MB -> 7.5 MB; a 3.1 Gb genome: 74 GB -> under 1 GB).
 *   site_mask : uint64[ceil(n_pos / 64)], bit (p & 63) of word p >> 6 = a recognition site of the alphabet starts at p, inside its contig
 *               (Alphabet::is_motif_match on the contig, src/common/nanopolish_alphabet.h -- the test np_site_table_genome_dev applies)
 *   word_rank : uint32[ceil(n_pos / 64) + 1], the number of sites before every word; the last entry is the total
 *   n_sites   : int64, device: the total.
 * The ordinal of the site at p is word_rank[p >> 6] + popcount(site_mask[p >> 6] & ((1 << (p & 63)) - 1)).  Built once per genome. */
int np_genome_site_index_dev(np_ctx* ctx, void* stream, const char* genome, const int64_t* contig_off, int n_contigs, int alphabet, int64_t n_pos,
                             uint64_t* site_mask, uint32_t* word_rank, int64_t* n_sites);

/* np_site_table_genome_dev into a table with one row per motif site: table int32[n_sites][6], row = the ordinal of the key's site
 * (np_genome_site_index_dev's site_mask / word_rank, of the same genome and alphabet).  Everything else as np_site_table_genome_dev. */
int np_site_table_genome_indexed_dev(np_ctx* ctx, void* stream, int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* last_site,
                                     const int32_t* n_motif, const np_hmm_job_dev* jobs, const int64_t* read_base, const char* genome,
                                     const int64_t* contig_off, int n_contigs, int alphabet, int min_separation, double call_threshold, int64_t n_pos,
                                     const uint64_t* site_mask, const uint32_t* word_rank, int32_t* table, uint64_t* n_overflow);

/* Read-level glue between the two kernels (src/nanopolish_squiggle_read.cpp:161-186,273-301):
 * builds base_to_event_map[].start for every read from kernel A's pairs, events_per_base and the
 * HMM transitions, then resolves each work item's event bounds
 *     e_start = get_closest_event_to(kpos_start), e_stop = get_closest_event_to(kpos_stop)
 * and applies the skip rule |e2-e1| <= 10 (src/basemods/nanopolish_basemods.cpp:356): skipped items get
 * the NP_JOB_SKIP flag and score NaN.
 *   pairs     : what np_event_align_dev wrote (a monotone path that advances one k-mer and/or one event per
 *               entry -- the event map relies on that to run without atomics)
 *   map_start : int32[sum n_kmers] scratch/output (per read at rank_off)
 *   kpos      : int32[2*n_jobs] read-strand k-mer positions bounding each item
 *   events_per_base : double[n_reads] output */
int np_resolve_jobs_dev(np_ctx* ctx, void* stream, int n_reads, np_read_dev* reads,
                        const int64_t* pair_off, const np_pair* pairs, const int32_t* pair_begin, const int32_t* n_pairs,
                        int32_t* map_start, double* events_per_base,
                        int64_t n_jobs, np_hmm_job_dev* jobs, const int32_t* kpos);

/* The same glue with load_from_raw's calibration step in between:
 *   event map (start, and stop when map_stop is not NULL: recalibration and the window bounds read .start only; callers that rebuild
 *   the reference's base_to_event_map -- the eventalign binding -- want both) -> recalibrate_model(scale_var, no drift;
 *   src/nanopolish_methyltrain.cpp:204-306) on the
 *   'M' entries of get_eventalignment_for_1d_basecalls (src/nanopolish_squiggle_read.cpp:339-389) -> work-item bounds.
 * reads[r] leaves with the calibrated shift/scale/var/log_var (and the HMM transitions); calibrated[r] = 0 marks reads
 * with < 200 'M' events or var > 2.5 (events cleared in the reference, squiggle_read.cpp:320-323): their items are
 * skipped.  event_mean / kmer_rank / model: the read's events, nucleotide k-mer ranks and base model, as for kernel A. */
int np_calibrate_resolve_dev(np_ctx* ctx, void* stream, int n_reads, np_read_dev* reads,
                             const float* event_mean, const uint16_t* kmer_rank, int model,
                             const int64_t* pair_off, const np_pair* pairs, const int32_t* pair_begin, const int32_t* n_pairs,
                             int32_t* map_start, int32_t* map_stop, double* events_per_base, int32_t* calibrated,
                             int64_t n_jobs, np_hmm_job_dev* jobs, const int32_t* kpos);

/* The event aligner's per-read constants {lp_skip, lp_stay, lp_step, lp_trim} (src/nanopolish_raw_loader.cpp:99-108) computed
 * with the library's restatement of glibc's log/exp (csrc/np_log.h) -- what np_mom_fill_dev computes on the device -- and
 * the restated functions themselves (out_log[i] = log(x[i]), out_exp[i] = exp(-x[i])); host-only, for verification. */
void np_aligner_constants(uint32_t n_events, uint32_t n_kmers, double out[4]);
void np_restated_log_exp(const double* x, size_t n, double* out_log, double* out_exp);
/* Host-only: the restated log / exp / logf against THIS host's libm on n pseudo-random arguments of the ranges the path uses;
 * *n_mismatch == 0 means the constants the device computes are bit-identical to what the reference computes on this host
 * (glibc 2.35 x86-64 with FMA does; another libm may not -- the restatement then still matches the build this repository
 * was verified against, tests/test_host_logic.py reports the difference). */
int  np_selftest_libm(uint64_t n, uint64_t seed, uint64_t* n_mismatch);

/* ---- f3: work-item generation on the device ------------------------------------------------ */
/* Device twin of np_cm_build_jobs_identity for a batch of identity-aligned reads: motif scan and grouping
 * (src/basemods/nanopolish_basemods.cpp:298-320), window and boundary rules (:328-345; src/alignment/nanopolish_alignment_db.cpp:
 * 65-71,697-708), methylated / unmethylated k-mer ranks of every window (Alphabet::methylate / reverse_complement,
 * HMMInputSequence::get_kmer_rank).  alphabet: cpg, gpc, dam or dcm.  All pointers are device pointers.
 *   ref_seq / seq_off : the reference strand of every read (A/C/G/T bytes, concatenated), int64[n_reads+1]
 *   group_off         : int64[n_reads+1], per-read CAPACITY in groups (slots); total_group_slots = group_off[n_reads] (host value)
 *   rank_off          : int64[n_reads+1], per-read capacity in job k-mer ranks (both versions of all windows)
 *   jobs              : 2 work items per group slot (unmethylated, methylated), ready for np_resolve_jobs_dev /
 *                       np_calibrate_resolve_dev; unused slots carry NP_JOB_SKIP
 *   kpos              : 2 x int32 per work item; first_site / last_site / n_motif: per group slot
 *   n_groups          : per read: groups written, or -1 if a capacity was too small */
int np_cm_build_jobs_identity_dev(np_ctx* ctx, void* stream, int n_reads, const char* ref_seq, const int64_t* seq_off,
                                  const uint8_t* read_rc, int alphabet, uint32_t k, int min_separation, int min_flank,
                                  const int64_t* group_off, int64_t total_group_slots, const int64_t* rank_off,
                                  np_hmm_job_dev* jobs, int32_t* kpos, uint16_t* job_ranks,
                                  int32_t* first_site, int32_t* last_site, int32_t* n_motif, int32_t* n_groups);

/* CIGAR-driven twin: the reads' base-to-reference alignments come from BAM CIGARs instead of
 * the identity.  Replaces, per read, SequenceAlignmentRecord / get_aligned_segments (src/alignment/nanopolish_alignment_db.cpp:
 * 30-50, src/alignment/nanopolish_anchor.cpp:20-95), the pair filter of EventAlignmentRecord (:63-72) and
 * AlignmentDB::_find_by_ref_bounds (:688-731) for both ends of every window; the aligned pairs are never materialised (a
 * per-read scan of the CIGAR operations + binary searches).  All pointers are device pointers.
 *   genome               : the contig(s), resident on the device (A/C/G/T, already disambiguated)
 *   ref_begin / ref_len  : int64[n_reads] / int32[n_reads]: the segment the reference fetches for read r,
 *                          contig[pos .. bam_endpos] inclusive, clipped to the contig (basemods.cpp:259-270), as an offset into genome
 *   cigar / cigar_off    : uint32 BAM words (length << 4 | op) of all reads, int64[n_reads+1]; a spliced record (N) yields no items
 *   read_len / read_rc   : SquiggleRead::read_sequence.length(), bam_is_rev
 *   deg_kpos             : int32[2*n_reads] output for np_cm_discard_degenerate_dev
 * first_site / last_site are relative to the record's pos.  Other arguments as np_cm_build_jobs_identity_dev. */
int np_cm_build_jobs_cigar_dev(np_ctx* ctx, void* stream, int n_reads, const char* genome, const int64_t* ref_begin,
                               const int32_t* ref_len, const uint32_t* cigar, const int64_t* cigar_off, int64_t total_cigar_ops,
                               const int32_t* read_len, const uint8_t* read_rc, int alphabet, uint32_t k, int min_separation,
                               int min_flank, const int64_t* group_off, int64_t total_group_slots, const int64_t* rank_off,
                               np_hmm_job_dev* jobs, int32_t* kpos, uint16_t* job_ranks,
                               int32_t* first_site, int32_t* last_site, int32_t* n_motif, int32_t* n_groups, int32_t* deg_kpos);
/* Declares the SLOT LAYOUT of the work-item array the calls that follow operate on (round 5): read r owns the items
 * [2 group_off[r], 2 group_off[r + 1]) -- the group slots np_cm_build_jobs_*_dev was given -- of which only the first 2 n_groups[r] are
 * live (n_groups is what the builder writes; a negative count reads as 0).  The slot ranges are sized for the worst case (one group per
 * min_separation + 1 bases, 2.7 x the groups a read has on average), and a kernel that visits "all n_jobs items" spends most of its
 * memory traffic on empty slots.  With the layout declared
 *     np_cm_build_jobs_*_dev (called with these group_off / n_groups), np_resolve_jobs_dev, np_calibrate_resolve_dev,
 *     np_cm_discard_degenerate_dev and np_hmm_score_dev
 * touch live items only: unused slots are neither written nor read, and their scores are NaN (one fill).  group_off and n_groups
 * (device arrays) must stay valid until the layout is cleared or replaced; n_jobs of those calls must be 2 * total_slots (anything else
 * is refused).  n_reads = 0 clears the layout: every kernel visits all n_jobs items again, as a caller with work items of its own needs
 * (the default).  Results never depend on it.  Reference: the per-read group lists of calculate_methylation_for_read,
 * src/basemods/nanopolish_basemods.cpp:289-336 -- the reference has no slots to skip. */
int np_set_job_layout(np_ctx* ctx, int n_reads, const int64_t* group_off, const int32_t* n_groups, int64_t total_slots);

/* After np_resolve_jobs_dev / np_calibrate_resolve_dev: drops every work item of a read whose first and last aligned event
 * coincide (the "degenerate alignment" rule of EventAlignmentRecord, alignment_db.cpp:83-86). */
int np_cm_discard_degenerate_dev(np_ctx* ctx, void* stream, const np_read_dev* reads, const int32_t* map_start,
                                 const int32_t* deg_kpos, int64_t n_jobs, np_hmm_job_dev* jobs);

/* Host mirrors (pure CPU): get_aligned_segments for a non-spliced record (returns the number of aligned pairs; NP_ERR_INVALID
 * for a spliced or malformed CIGAR), and the work items of a CIGAR-aligned read (see np_cm_build_jobs_identity;
 * ref_seq[0..n) is the fetched reference segment, deg_kpos[2] as above). */
int np_cigar_aligned_bases(const uint32_t* cigar, int n_cigar, int ref_pos0, int32_t* ref_pos, int32_t* read_pos, int cap);
int np_cm_build_jobs_cigar(int alphabet, const char* ref_seq, size_t n, const uint32_t* cigar, int n_cigar,
                           int read_len, int read_rc, uint32_t k, int min_separation, int min_flank,
                           int cap_jobs, int64_t cap_ranks, int32_t* first_site, int32_t* last_site, int32_t* n_motif,
                           int32_t* kpos, int32_t* job_n_kmers, uint16_t* ranks_unmeth, uint16_t* ranks_meth,
                           int64_t* rank_off, int32_t* deg_kpos);

/* ---- eventalign: the segment chain of align_read_to_ref on the device ---------------------------------------------------------- */
/* Per-read status of np_eventalign_dev */
#define NP_EA_OK          0
#define NP_EA_OVERFLOW    1   /* a segment needs more lattice rows than the scratch holds, or the output capacity is too small */
#define NP_EA_BAD_RECORD  2   /* the record points outside the read / the fetched reference (the reference asserts there) */
/* align_read_to_ref (src/alignment/nanopolish_eventalign.cpp:612-826) for a batch of reads whose event alignment, event map
 * and (calibrated) scalings are on the device (np_event_align_dev -> np_resolve_jobs_dev / np_calibrate_resolve_dev with
 * n_jobs = 0 is enough): per read the chain of ~100-base segments, each one profile_hmm_align (flags 0, base model) from the
 * previous segment's last emitted event, emitting ~50 aligned events per segment.  One wavefront per read.
 *   reads / event_mean / map_start / n_pairs : as left by the calls above
 *   genome, ref_begin, ref_len, cigar, cigar_off, read_len, read_rc : as np_cm_build_jobs_cigar_dev
 *   out_off  : int64[n_reads+1], per-read output capacity (n_events + 1 rows always suffice)
 *   out_ref / out_event / out_state : EventAlignment::ref_position (relative to the record's pos), ::event_idx, ::hmm_state
 *              ('M' or 'B'), in the reference's output order
 *   n_out / status / n_calls : rows written, NP_EA_*, number of profile_hmm_align calls (segments) per read
 * A read that failed the aligner / calibration / events-per-base QC (no events in the reference) yields n_out = 0. */
int np_eventalign_dev(np_ctx* ctx, void* stream, int n_reads, const np_read_dev* reads, const float* event_mean,
                      const int32_t* map_start, const int32_t* n_pairs, const double* events_per_base, const int32_t* calibrated,
                      int model, const char* genome, const int64_t* ref_begin, const int32_t* ref_len,
                      const uint32_t* cigar, const int64_t* cigar_off, int64_t total_cigar_ops,
                      const int32_t* read_len, const uint8_t* read_rc, uint32_t k,
                      const int64_t* out_off, int32_t* out_ref, int32_t* out_event, uint8_t* out_state,
                      int32_t* n_out, int32_t* status, int32_t* n_calls);

/* ---- f2: the stage in front of the event aligner ------------------------------------------ */
/* detector_param, src/thirdparty/scrappie/event_detection.h:6-12 */
typedef struct np_detector_param {
    uint32_t window_length1, window_length2;
    float threshold1, threshold2, peak_height;
} np_detector_param;
/* event_detection_defaults (rna == 0, :15-21) / event_detection_rna (rna != 0, :23-29) */
void np_event_detection_params(np_detector_param* p, int rna);

#define NP_ED_OVERFLOW (-1)   /* n_events[r]: more events than the caller's capacity for the read                      */
#define NP_ED_INEXACT  (-2)   /* n_events[r]: the read holds a non-finite sample (inf / nan): the reference's own result is     */
                              /* undefined.                 (A read whose double-precision prefix sums are merely not       */
                              /* PROVABLY exact -- a near-zero sample does that -- is not refused any more: its sums are      */
                              /* accumulated serially, as the reference does; np_get_stat "ed_serial_reads" counts them.)     */

/* The signal loaders' conversion of ADC counts to pA, in front of detect_events:
 *     rawptr[i] = ((float)raw_signal[i] + offset) * raw_unit,   raw_unit = range / digitisation  (all fp32)
 * (src/io/nanopolish_fast5_loader.cpp:96-103, src/io/nanopolish_fast5_io.cpp:163-165).  A host-fed batch then uploads int16
 * samples, half the bytes.  adc: int16[total samples]; offset / raw_unit: float[n_reads]; raw_pa: float[total samples] out.
 * (The detector's exactness verdicts are NOT kept as context state any more -- round 5 matched them to the next np_detect_events_dev by pointer
 * identity, and a caller that edited raw_pa in place between the two calls silently inherited stale verdicts.  They are an explicit
 * by-product now: np_adc_to_pa_checked_dev / np_detect_events_checked_dev below.) */
int np_adc_to_pa_dev(np_ctx* ctx, void* stream, int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples,
                     const float* offset, const float* raw_unit, float* raw_pa);

/* The same conversion, which also takes the detector's per-read exactness verdict from the values on their way out (one pass over the samples
 * instead of two: the bound of src/thirdparty/scrappie/event_detection.c:35-58's prefix sums).
 *   verdict : int32[n_reads], device, out -- opaque; valid for exactly the samples this call wrote.
 * Hand it to np_detect_events_checked_dev together with the same (raw_pa, raw_off, n_reads).  CONTRACT: the caller must not change raw_pa
 * between the two calls; one that does passes verdict = NULL there (or calls np_detect_events_dev) and the detector makes its own pass.
 * Nothing is remembered in the context: a stale verdict can only come from the caller's own hands. */
int np_adc_to_pa_checked_dev(np_ctx* ctx, void* stream, int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples,
                             const float* offset, const float* raw_unit, float* raw_pa, int32_t* verdict);

/* detect_events (src/thirdparty/scrappie/event_detection.c:268-319) on the WHOLE raw table of every read, as
 * SquiggleRead::load_from_raw runs it (src/nanopolish_squiggle_read.cpp:229-236; the trim it computes is discarded there).
 *   raw / raw_off      : float[total samples] (pA), int64[n_reads+1]
 *   max_samples        : largest read (host value, sizes the launch)
 *   tstat              : scratch, 2 floats per sample + 64 bytes, 64-byte aligned (read back in whole cache lines)
 *   event_off          : int64[n_reads+1], per-read CAPACITY offsets into the event arrays (n_samples/2+2 is always enough:
 *                        boundaries of one detector are at least two samples apart)
 *   event_start/length/mean/stdv : event_t fields (scrappie_structures.h:8-15) of the detected events
 *   n_events           : per read: the event count (1 + #boundaries), 0 if no boundary was found (undefined in the
 *                        reference), or NP_ED_OVERFLOW / NP_ED_INEXACT */
int np_detect_events_dev(np_ctx* ctx, void* stream, int n_reads, const float* raw, const int64_t* raw_off, int64_t max_samples,
                         const np_detector_param* params, float* tstat, const int64_t* event_off, int64_t max_events,
                         uint32_t* event_start, float* event_length, float* event_mean, float* event_stdv, int32_t* n_events);
/* The same call with the exactness verdicts np_adc_to_pa_checked_dev produced for exactly these samples (verdict: device int32[n_reads], or
 * NULL = np_detect_events_dev).  Events are identical either way; with verdicts the detector skips its own bound pass over the samples. */
int np_detect_events_checked_dev(np_ctx* ctx, void* stream, int n_reads, const float* raw, const int64_t* raw_off, int64_t max_samples,
                                 const np_detector_param* params, float* tstat, const int64_t* event_off, int64_t max_events,
                                 uint32_t* event_start, float* event_length, float* event_mean, float* event_stdv, int32_t* n_events,
                                 const int32_t* verdict);

/* The two calls above as ONE, for a caller that keeps the counts: int16 ADC counts in, events out (the loaders' conversion,
 * src/io/nanopolish_fast5_loader.cpp:96-103, and scrappie's detect_events, src/thirdparty/scrappie/event_detection.c:268-319, as
 * SquiggleRead::load_from_raw chains them).  Events are bit-identical to np_adc_to_pa_checked_dev + np_detect_events_checked_dev on the same counts.
 * With the DNA windows (3 and 6 samples) a read of 2 048 samples or more never exists as pA values: the conversion's pass only takes the
 * exactness verdict, the peak walk and the event sums convert the counts they load -- 2 bytes per sample read where the two calls write 4 and
 * read them back twice.
 *   adc      : 4-byte aligned; raw_off / max_samples / offset / raw_unit as np_adc_to_pa_dev
 *   raw_pa   : float scratch with the counts' layout (raw_off): written for reads shorter than 2 048 samples and for reads whose prefix sums are
 *              not provably exact (the serial path), unspecified elsewhere; every read's with other window lengths (RNA: the two-call form inside)
 *   the rest as np_detect_events_dev. */
int np_detect_events_adc_dev(np_ctx* ctx, void* stream, int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples,
                             const float* offset, const float* raw_unit, float* raw_pa, const np_detector_param* params, float* tstat,
                             const int64_t* event_off, int64_t max_events, uint32_t* event_start, float* event_length, float* event_mean,
                             float* event_stdv, int32_t* n_events);
/* Host-pointer convenience form for one batch of reads (copies in and out). out_* are concatenated, out_off[n_reads+1]. */
int np_detect_events_host(np_ctx* ctx, int n_reads, const float* const* raw, const uint32_t* n_samples,
                          const np_detector_param* params, uint32_t* out_start, float* out_length, float* out_mean,
                          float* out_stdv, int64_t cap, int64_t* out_off);

/* estimate_scalings_using_mom (src/nanopolish_raw_loader.cpp:30-75) + the event aligner's per-read constants (:99-108) on the
 * device: completes reads[r] (event_off, rank_off, n_kmers already set by the caller) with n_events, shift/scale (var = 1)
 * and lp_skip/lp_stay/lp_step/lp_trim, computed with glibc's log/exp restated (csrc/np_log.h).  reads_b (nullable): a second
 * record array (the one kernel B reads) that receives n_events. */
int np_mom_fill_dev(np_ctx* ctx, void* stream, int n_reads, np_read_dev* reads, np_read_dev* reads_b, const float* event_mean,
                    const int32_t* n_events, const uint16_t* kmer_rank, int model);

/* Direct-RNA reads (SquiggleRead::nucleotide_type == SRNT_RNA): load_from_raw uses the kit r9.4_70bps / alphabet u_to_t_rna / k = 5 model, the
 * RNA detector parameters (np_event_detection_params(p, 1)), and REVERSES the detected events -- the strand is sequenced 3' -> 5' -- after the
 * MoM scalings and before the event aligner (src/nanopolish_squiggle_read.cpp:206-213,260-263).  This entry point is that reversal for a
 * batch, in place, on whichever of the four per-event arrays are given (event_off / n_events as np_detect_events_dev left them). */
int np_reverse_events_dev(np_ctx* ctx, void* stream, int n_reads, const int64_t* event_off, const int32_t* n_events, uint32_t* event_start,
                          float* event_length, float* event_mean, float* event_stdv);

/* Device self-test: the emission's exact fast division (reciprocal + two fused corrections) against the IEEE fp32
 * divide on n_samples pseudo-random operand pairs; *n_mismatch must come back 0. */
int np_selftest_division(np_ctx* ctx, uint64_t n_samples, uint64_t seed, uint64_t* n_mismatch);

/* Device self-test: the detector's two-operation division by a window length w (q = RN(x ch + RN(x cl)), ch + cl = 1 / w to twice the
 * precision; csrc/np_events_kernels.hip:div_small_f32 / div_small_f64 -- event_detection.c:97-101,111 divide by w_lengthf) against the IEEE
 * divide: fp32 on EVERY float of magnitude 0 or >= 2^-100 (*n_f32_compared of them), fp64 on n_f64 pseudo-random doubles.  w = 3 and 6 run
 * with the constants the fused walk itself uses.  Both mismatch counts must come back 0. */
int np_selftest_division_small(np_ctx* ctx, int w, uint64_t n_f64, uint64_t* n_mismatch_f32, uint64_t* n_mismatch_f64, uint64_t* n_f32_compared);

/* Device self-test: the fused detector walk's last step, (float)(|delta_mean| / sqrt(combined_var / w)) in double (event_detection.c:111), as
 * it is FILTERED (csrc/np_events_kernels.hip:ed_ratio_filtered: a reciprocal square root refined once, trusted only where the product stays clear
 * of every float rounding boundary by 2^-39, the exact sequence elsewhere) against the exact sequence on n_samples pseudo-random operand pairs,
 * a third of them steered to within 2^-24 of a boundary.  *n_mismatch (trusted values that differ) must come back 0; *n_sent_to_exact counts
 * the values the filter handed to the exact sequence; *farthest_disagreement is the largest distance from a boundary, in units of 2^-53, at
 * which an UNFILTERED value differed -- against the filter's band of 16384. */
int np_selftest_tstat_ratio(np_ctx* ctx, uint64_t n_samples, uint64_t seed, uint64_t* n_mismatch, uint64_t* n_sent_to_exact, uint64_t* farthest_disagreement);

/* Device memory, pinned host memory, streams and events for bindings that are not HIP programs themselves
 * (csrc/np_batch_dropin.cpp is plain C++ inside a nanopolish build).  Copies and fills are plain stream operations enqueued on
 * `stream` (0 = the context's own): they take no part in the one-stream-at-a-time rule of the compute entry points, so an upload
 * on one stream can run beside the context's kernels on another -- ordered by the caller's events (np_event_record on the
 * producing stream, np_stream_wait_event on the consuming one).  Host buffers of asynchronous copies must stay valid until the
 * copy has completed; pageable host memory makes a copy synchronous, as in HIP: use np_host_alloc (pinned) for staging. */
void* np_dev_alloc(np_ctx* ctx, size_t bytes);
void  np_dev_free(np_ctx* ctx, void* p);
void* np_host_alloc(np_ctx* ctx, size_t bytes);
void  np_host_free(np_ctx* ctx, void* p);
int   np_copy_to_device(np_ctx* ctx, void* stream, void* dst_dev, const void* src_host, size_t bytes);
int   np_copy_to_host(np_ctx* ctx, void* stream, void* dst_host, const void* src_dev, size_t bytes);
int   np_memset_dev(np_ctx* ctx, void* stream, void* dst_dev, int value, size_t bytes);
void* np_stream_create(np_ctx* ctx);                 /* a hipStream_t (non-blocking), NULL on failure */
void  np_stream_destroy(np_ctx* ctx, void* stream);  /* waits for the stream's work first */
void* np_event_create(np_ctx* ctx);                  /* a hipEvent_t without timing; a host thread that waits on it (np_event_sync) SLEEPS
                                                        (hipEventBlockingSync): the bindings' waiting threads must not spin on a core the
                                                        packing and result-building workers need */
void  np_event_destroy(np_ctx* ctx, void* event);
int   np_event_record(np_ctx* ctx, void* event, void* stream);
int   np_stream_wait_event(np_ctx* ctx, void* stream, void* event);
int   np_event_sync(np_ctx* ctx, void* event);       /* blocks the calling host thread */
int   np_event_query(np_ctx* ctx, void* event);      /* NP_OK: the work before the event's last record has finished; 1: not yet; < 0: error */

/* Synchronise the context's stream (or the given one). */
int np_sync(np_ctx* ctx, void* stream);

/* Time (ms) spent in the most recent launch of each kernel family on the device, measured with HIP events
 * on the launching stream.  which: 0 = event align, 1 = hmm score, 2 = resolve / calibrate / work items, 3 = hmm viterbi,
 * 4 = event detection, 5 = MoM scalings, 6 = eventalign chain (7 and 8 are unused: they belonged to round 3's split-aligner and
 * side-stream experiments, removed in round 5; their write-ups are profiles/r03_kernel_a_split.md and profiles/r03_experiments_tail.md). */
int np_last_kernel_ms(np_ctx* ctx, int which, float* ms);
/* Accumulated device time (ms) and launch count of a kernel family since the last reset (call after np_sync). */
int np_kernel_time(np_ctx* ctx, int which, double* total_ms, int64_t* launches, int reset);

#ifdef __cplusplus
}
#endif
#endif
