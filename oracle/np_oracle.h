/* oracle/np_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Portable plain-C restatement of nanopolish's signal-HMM hot path (v0.14.0), used as the CPU
 * parity checker for the HIP kernels.  Every function cites the reference file:line it follows
 * (paths relative to the nanopolish source tree).  Pinned against the reference's own code
 * (oracle/_ref/libnp_ref.so, built in place by oracle/Makefile) by tests/test_oracle_vs_ref.py
 * in the build container, and against the committed golden vectors (tests/golden/) everywhere.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 */
#ifndef NP_ORACLE_H
#define NP_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- alphabets: src/common/nanopolish_alphabet.{h,cpp} -------------------------------- */
enum { NPO_ALPHA_NUCLEOTIDE = 0, NPO_ALPHA_CPG, NPO_ALPHA_GPC, NPO_ALPHA_DAM, NPO_ALPHA_DCM, NPO_ALPHA_U_TO_T_RNA, NPO_NUM_ALPHABETS };
int      npo_alphabet_id(const char* name);
uint32_t npo_alphabet_size(int a);
uint32_t npo_kmer_rank(int a, const char* str, uint32_t k);
void     npo_reverse_complement(int a, const char* in, int n, char* out);   /* out[n] NUL-terminated */
void     npo_methylate(int a, const char* in, int n, char* out);
void     npo_unmethylate(int a, const char* in, int n, char* out);
int      npo_is_motif_match(int a, const char* str, int n, int i);
/* ranks of the n-k+1 k-mers as HMMInputSequence::get_kmer_rank(i,k,do_rc) reports them */
void     npo_sequence_kmer_ranks(int a, const char* seq, const char* rc_seq, int n, int k, int do_rc, uint32_t* out);

/* ---- log-sum table: src/common/logsum.{h,cpp} ------------------------------------------ */
#define NPO_LOGSUM_TBL 16000
const float* npo_flogsum_table(void);
float        npo_flogsum(float a, float b);

/* ---- model + scalings ---------------------------------------------------------------------- */
typedef struct {
    int k;
    int n_states;
    const double* level_mean;     /* PoreModelStateParams::level_mean     */
    const double* level_stdv;     /*                        level_stdv     */
    const double* level_log_stdv; /*                        level_log_stdv */
} npo_model;

typedef struct { double shift, scale, drift, var, log_var; } npo_scalings;
npo_scalings npo_set4(double shift, double scale, double drift, double var);

float npo_log_probability_match_r9(const npo_model* m, const npo_scalings* s, uint32_t rank, float level, float time);

/* ---- profile HMM: src/hmm/nanopolish_profile_hmm_r9.{h,cpp,inl} ------------------------------ */
#define NPO_HAF_ALLOW_PRE_CLIP 1u
#define NPO_HAF_ALLOW_POST_CLIP 2u
/* order: lp_mm_self, lp_mb, lp_mk, lp_mm_next, lp_bb, lp_bk, lp_bm_next, lp_bm_self, lp_kk, lp_km */
void  npo_calculate_transitions(double events_per_base, double indel_bias, float out[10]);
void  npo_make_flanks(uint32_t num_events, float* pre_flank /* n+1 */, float* post_flank /* n */);

float npo_profile_hmm_score(const npo_model* m, const npo_scalings* s, const float* event_mean,
                            const uint32_t* kmer_ranks, uint32_t n_kmers,
                            uint32_t e_start, uint32_t e_stop, int stride,
                            double events_per_base, double indel_bias, uint32_t flags);

/* returns the number of alignment states (ascending order), -1 if the reference would assert */
int   npo_profile_hmm_align(const npo_model* m, const npo_scalings* s, const float* event_mean,
                            const uint32_t* kmer_ranks, uint32_t n_kmers,
                            uint32_t e_start, uint32_t e_stop, int stride,
                            double events_per_base, double indel_bias, uint32_t flags,
                            uint32_t* out_event_idx, uint32_t* out_kmer_idx, double* out_l_fm, char* out_state, int cap);

/* profile_hmm_score_set's combination of per-sequence scores (src/hmm/nanopolish_profile_hmm.cpp:32-56) */
float npo_combine_score_set(const float* scores, int n);

/* ---- raw loader: src/nanopolish_raw_loader.cpp ------------------------------------------------- */
void npo_estimate_scalings_mom(const npo_model* m, const uint32_t* kmer_ranks, uint32_t n_kmers,
                               const float* event_mean, uint32_t n_events, double* shift, double* scale);

/* returns #pairs (0 == QC failure, as the reference's empty vector); -2 if the reference would read
 * its trace array out of bounds (undefined behaviour there). out_pairs: interleaved (ref_pos, read_pos). */
int  npo_adaptive_banded_simple_event_align(const npo_model* m, const npo_scalings* s,
                                            const float* event_mean, uint32_t n_events,
                                            const uint32_t* kmer_ranks, uint32_t n_kmers,
                                            int32_t* out_pairs, int cap);

/* ---- read-level glue between the two kernels: src/nanopolish_squiggle_read.cpp ------------------ */
/* base_to_event_map (:273-301): start/stop per k-mer (-1 if none), events_per_base */
void npo_build_base_to_event_map(const int32_t* pairs, int n_pairs, uint32_t n_kmers,
                                 int32_t* start, int32_t* stop, double* events_per_base);
/* get_closest_event_to (:161-186) */
int  npo_get_closest_event_to(const int32_t* start, uint32_t n_kmers, int k_idx);

/* f1: get_eventalignment_for_1d_basecalls (squiggle_read.cpp:339-389) + recalibrate_model (methyltrain.cpp:204-306,
 * scale_var, no drift).  Returns 1 if recalibrated.  The Eigen fullPivLu step is restated without Eigen (the only unpinned step). */
void npo_set_eigen32_scalar_div(int on);   /* test switch, see np_oracle.c */
int  npo_recalibrate(const npo_model* m, const float* event_mean, const uint32_t* kmer_ranks, uint32_t n_kmers,
                     const int32_t* map_start, const int32_t* map_stop, double* shift, double* scale, double* var);

/* ---- f2: scrappie event detection (src/thirdparty/scrappie/event_detection.c:268-319) on a whole raw table ------- */
int  npo_detect_events(const float* raw, size_t n, size_t w1, size_t w2, float t1, float t2, float peak_height,
                       uint64_t* out_start, float* out_length, float* out_mean, float* out_stdv, size_t cap);
void npo_detect_events_many(int n_reads, const float* raw, const int64_t* raw_off, float* out_mean, const int64_t* ev_off,
                            int32_t* out_n, int n_threads);

void npo_aligner_constants(uint32_t n_events, uint32_t n_kmers, double out[4]);   /* raw_loader.cpp:99-108 */

/* ---- call-methylation work-item rules: src/basemods/nanopolish_basemods.cpp:298-358 ---------------- */
/* EventAlignmentRecord (src/alignment/nanopolish_alignment_db.cpp:55-91): maps aligned bases to events.
 * aligned_bases: interleaved (ref_pos, read_pos); out_aligned_events: interleaved (ref_pos, event_idx).
 * returns count (0 if degenerate). */
int  npo_event_alignment_record(const int32_t* aligned_bases, int n_bases, int read_length, int k, int seq_rc,
                                const int32_t* map_start, uint32_t n_kmers, int32_t* out_aligned_events);
/* get_aligned_segments (src/alignment/nanopolish_anchor.cpp:20-95) for the single segment SequenceAlignmentRecord accepts
 * (src/alignment/nanopolish_alignment_db.cpp:41-49): (ref_pos, read_pos) pairs of the M/=/X operations of a BAM CIGAR
 * (uint32 words, length << 4 | op).  Returns the number of pairs (counted even beyond cap), or -1 for a spliced record. */
int  npo_cigar_aligned_bases(const uint32_t* cigar, int n_cigar, int pos, int32_t* out_pairs, int cap);
/* AlignmentDB::_find_by_ref_bounds (src/alignment/nanopolish_alignment_db.cpp:688-731) */
int  npo_find_by_ref_bounds(const int32_t* pairs, int n, int ref_start, int ref_stop, int* read_start, int* read_stop);
/* Motif scan + grouping (:298-320) and window rule (:328-338). Outputs per group: first/last motif site
 * position and #motifs.  Returns #groups (all groups, before the skip rules). */
int  npo_scan_motif_groups(int a, const char* ref_seq, int n, int min_separation,
                           int32_t* first_site, int32_t* last_site, int32_t* n_motif, int cap);

/* ---- bounded CPU driver used by bench.py's cpu_baseline (kind="port") ----------------------------- */
void npo_align_many(const npo_model* m, int n_reads, const float* event_mean, const int64_t* event_off,
                    const uint32_t* ranks, const int64_t* rank_off, const double* shift, const double* scale,
                    int32_t* out_pairs, const int64_t* pair_off, int32_t* out_n, int n_threads);
void npo_score_many(const npo_model* m, int64_t n_jobs, const int32_t* job_read,
                    const float* event_mean, const int64_t* event_off,
                    const double* shift, const double* scale, const double* var, const double* events_per_base,
                    const uint32_t* ranks, const int64_t* job_rank_off,
                    const uint32_t* e_start, const uint32_t* e_stop, const int8_t* stride,
                    double indel_bias, uint32_t flags, float* out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
