// np_kernels.h -- host-visible launch interface of the HIP kernels (internal to the library).
#pragma once
#include "np_device.h"

struct np_hmm_args {
    const np_hmm_job_dev* jobs;
    const uint32_t* order;         // job indices of this size class
    const uint32_t* n_class_jobs;  // device counter: number of entries in `order`
    const np_read_dev* reads;
    const float* event_mean;
    const uint16_t* ranks;
    const np_state_dev* model;
    const float* logsum;           // flogsum_lookup[16000]
    const float* flank;            // universal clip-flank table (see np_capi: flank[i] == pre_flank[i] == post_flank[e-1-i])
    uint32_t* counter;             // work-queue head, zeroed before launch
    float* out;                    // forward scores, indexed by job
    // Viterbi only
    uint8_t* bp;                   // back-pointer scratch
    float* vm;                     // lattice scratch
    const int64_t* cell_off;       // per job offset into bp / vm (in cells)
    np_hmm_state* states;          // output
    const int64_t* state_off;
    int32_t* n_states;
    int prio;                      // forward kernel: wave priority (s_setprio 0..3) -- above 0 only when the caller co-schedules it with
                                   // the event aligner's back-track launch, whose scalar chain would otherwise starve it
};

struct np_ea_args {
    int n_reads;
    const np_read_dev* reads;      // calibrated scalings + HMM transitions (np_resolve_jobs_dev / np_calibrate_resolve_dev)
    const float* event_mean;
    const int32_t* map_start;      // base_to_event_map[].start per read at rank_off
    const int32_t* n_pairs;        // aligner result per read (0: failed, events cleared)
    const double* events_per_base;
    const int32_t* calibrated;     // may be null
    const np_state_dev* model;
    const float* flank;
    const char* genome;
    const int64_t* ref_begin;
    const int32_t* ref_len;
    const uint32_t* cigar;
    const int64_t* cigar_off;
    const int32_t* op_ref;         // np_cigar_index_kernel output
    const int32_t* op_read;
    const int32_t* cig_reads;      // 4 x int32 per read (first_q, last_q, ok, pad)
    const int32_t* read_len;
    const uint8_t* read_rc;
    int k;
    uint8_t* bp;                   // per-wave back-pointer scratch: rows_cap x 128 B
    size_t bp_stride;
    int rows_cap;
    int max_kmers;                 // k-mers per segment the launched instantiation holds (32 lanes x blocks per lane)
    uint32_t* path;                // per-wave path list: rows_cap + 128 entries
    size_t path_stride;
    const int64_t* out_off;
    int32_t* out_ref;
    int32_t* out_event;
    uint8_t* out_state;
    int32_t* n_out;
    int32_t* status;
    int32_t* n_calls;
    uint32_t* counter;
    int walk_prio;                 // two-read kernel: raise the wave priority during back-track + emission (experiment knob)
    unsigned long long* stats;     // [0] += lattice cells (e + 1) x 3 (n + 2) of every segment, [1] += lattice rows e, [2] += k-mers n (nullable)
};

struct np_align_args {
    const np_read_dev* reads;
    const float* event_mean;
    const uint16_t* ranks;
    const np_state_dev* model;
    const int64_t* pair_off;
    np_pair* pairs;
    int32_t* pair_begin;
    int32_t* n_pairs;
    uint64_t* trace;               // scratch: n_wave_slots * trace_stride u64
    uint64_t trace_stride;         // u64 per resident wave (>= 32 * ceil(max_bands / 8): 256 B per 8 bands)
    float4* kparams;               // scratch: n_wave_slots * kp_stride records (scaled Gaussian per k-mer of the read in flight)
    uint64_t kp_stride;            // records per resident wave (>= max k-mers per read)
    uint32_t* counter;
    const uint32_t* order;         // issue order of the read queue (longest first), or null: index order
    int32_t n_reads;
    int32_t max_gap_threshold;
    double min_average_log_emission;
};

#define NP_NUM_CLASSES 8
// size classes of the HMM kernels: (lanes per job, k-mer blocks per lane), by the item's k-mers: <= 16, 24, 32, 64, 128, 256, 512, 1024.
// The three-lane class (round 3) is for 17..24 k-mers -- the variants shape (17) and the two-site methylation windows: 21 items per
// wave on 63 lanes, 6..8 blocks per lane and a skew of two steps, where four lanes carry 5..6 blocks (20..24 slots) and three steps.
static const int NP_CLASS_SEG[NP_NUM_CLASSES] = {2, 3, 4, 8, 16, 32, 64, 64};
static const int NP_CLASS_C[NP_NUM_CLASSES] = {8, 8, 8, 8, 8, 8, 8, 16};

__host__ __device__ inline int np_size_class(uint32_t n)
{
    if (n == 0) return -1;
    if (n <= 16) return 0;
    if (n <= 24) return 1;
    if (n <= 32) return 2;
    if (n <= 64) return 3;
    if (n <= 128) return 4;
    if (n <= 256) return 5;
    if (n <= 512) return 6;
    if (n <= 1024) return 7;
    return -1;
}

// Work-item binning.  A bin is (size class, k-mer blocks per lane, event-count bucket): the items that share a wave
// then need the same number of blocks per lane (the forward kernel runs every lane for the wave's maximum) and have
// nearly the same number of rows (no padding steps); inside a class the widest and longest packs are issued first.
// Counting sort in three small kernels: histogram -> exclusive scan -> scatter.
#define NP_EBUCKETS 64
#define NP_CPL 8                 // blocks-per-lane groups inside a class (ceil(n / SEG) scaled to 1..8)
#define NP_NBINS (NP_NUM_CLASSES * NP_CPL * NP_EBUCKETS)

__host__ __device__ inline int np_job_bin(const np_hmm_job_dev& jb, uint32_t flank_len)
{
    const uint32_t e = (jb.e_stop > jb.e_start ? jb.e_stop - jb.e_start : jb.e_start - jb.e_stop) + 1u;
    const int cls = (e <= flank_len && !(jb.flags & NP_JOB_SKIP)) ? np_size_class(jb.n_kmers) : -1;
    if (cls < 0) return -1;
    // bucket width 1, 1, 2, 4, 8, 16, 16, 16 events (round 3: the two smallest classes -- three quarters of the methylation items and all of
    // the variants shape -- are sorted by their EXACT event count: a pack runs for its longest item, and buckets of four events cost
    // those classes 5 % of their steps)
    const uint32_t shift = cls < 2 ? 0u : (cls < 6 ? (uint32_t)(cls - 1) : 4u);
    const uint32_t bucket = (e >> shift) < (NP_EBUCKETS - 1) ? (e >> shift) : (NP_EBUCKETS - 1);
    // blocks per lane the item needs: ceil(n / SEG), in units of the class' C / 8 (C = 8: 1..8; C = 16: pairs)
    const int seg = cls == 0 ? 2 : (cls == 1 ? 3 : ((1 << cls) < 64 ? (1 << cls) : 64)), cu = (cls == NP_NUM_CLASSES - 1 ? 16 : 8) / NP_CPL;   // NP_CLASS_SEG / NP_CLASS_C
    const int cpl = ((int)jb.n_kmers + seg - 1) / seg;
    const int cg = (cpl + cu - 1) / cu;                                     // 1..8
    // descending blocks per lane, then descending event count, inside a class
    return (cls * NP_CPL + (NP_CPL - cg)) * NP_EBUCKETS + (NP_EBUCKETS - 1 - (int)bucket);
}


hipError_t np_launch_hmm_forward(int cls, const np_hmm_args& a, int n_blocks, bool lse_oor, hipStream_t s);
hipError_t np_hmm_forward_lds_bytes(int cls, size_t* bytes);
// hardware probes (np_hmm_kernels.hip): out = 8 x uint32 (zeroed), buf = 16 floats of 1.0f, sbuf = 32 x uint16
hipError_t np_launch_probe(const float* logsum, const float* buf, uint16_t* sbuf, uint32_t* out, int n_blocks, hipStream_t s);
hipError_t np_launch_hmm_viterbi(int cls, const np_hmm_args& a, int n_blocks, hipStream_t s);
hipError_t np_launch_hmm_backtrack(const np_hmm_args& a, int64_t n_jobs, hipStream_t s);
hipError_t np_launch_event_align(const np_align_args& a, int n_blocks, hipStream_t s);
hipError_t np_launch_align_order(int n_reads, const np_read_dev* reads, uint32_t* scratch /* 2048 + n_reads */, hipStream_t s);
int np_align_block_threads(void);
int np_hmm_block_threads(int cls);
int np_vit_block_threads(void);

// glue kernels
// The slot layout of a work-item array (np_set_job_layout): read r owns items [2 group_off[r], 2 group_off[r + 1]), of which the first
// 2 max(n_groups[r], 0) are live; n_reads == 0: no layout, every kernel visits all n_jobs items.
struct np_slots { const int64_t* group_off; const int32_t* n_groups; int n_reads; };
hipError_t np_launch_classify(const np_hmm_job_dev* jobs, int64_t n_jobs, uint32_t* class_count /*[7]*/,
                              uint32_t* order /*[NP_NUM_CLASSES][n_jobs]*/, float* out_scores, uint32_t flank_len, uint32_t* bins /*[2 * 8 * 8 * 64]*/, np_slots lay, hipStream_t s);
hipError_t np_launch_build_map(int n_reads, np_read_dev* reads, const int64_t* pair_off, const np_pair* pairs,
                               const int32_t* pair_begin, const int32_t* n_pairs, int32_t* map_start, int32_t* map_stop,
                               double* events_per_base, double indel_bias, hipStream_t s);
hipError_t np_launch_recalibrate(int n_reads, np_read_dev* reads, const float* event_mean, const uint16_t* ranks,
                                 const np_state_dev* model, int n_states, const int32_t* n_pairs, const int32_t* map_start,
                                 int32_t* calibrated, const uint32_t* order /* read order of the groups (np_launch_align_order) or null */,
                                 int shape /* 0: the default workgroup shape; 1, 2: A/B alternatives */, hipStream_t s);
hipError_t np_launch_discard_degenerate(int64_t n_jobs, np_hmm_job_dev* jobs, const np_read_dev* reads, const int32_t* map_start,
                                        const int32_t* deg_kpos, np_slots lay, hipStream_t s);
hipError_t np_launch_resolve(int64_t n_jobs, np_hmm_job_dev* jobs, const np_read_dev* reads, const int32_t* n_pairs,
                             const double* events_per_base, const int32_t* calibrated, const int32_t* map_start,
                             const int32_t* kpos, np_slots lay, hipStream_t s);
hipError_t np_launch_site_table(int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* n_motif,
                                const np_hmm_job_dev* jobs, const int64_t* read_base, double call_threshold, int64_t n_pos,
                                int32_t* table, hipStream_t s);
hipError_t np_launch_site_table_genome(int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* last_site, const int32_t* n_motif,
                                       const np_hmm_job_dev* jobs, const int64_t* read_base, const char* genome, const int64_t* contig_off, int n_contigs,
                                       int alphabet, int min_separation, double call_threshold, int64_t n_pos, int32_t* table,
                                       unsigned long long* n_overflow, const uint64_t* site_mask, const uint32_t* word_rank, hipStream_t s);
hipError_t np_launch_genome_site_index(const char* genome, const int64_t* contig_off, int n_contigs, int alphabet, int64_t n_pos, uint64_t* site_mask,
                                       uint32_t* word_rank, int64_t* n_sites, uint32_t* chunk_scratch, hipStream_t s);
int64_t np_site_rank_chunks(int64_t n_pos);
hipError_t np_launch_score_set_combine(int64_t n_sets, const int64_t* set_off, const int64_t* member_idx, const float* member_scores,
                                       const float* logsum, const double* log_n /* host-constants mode: log(0 .. 64) by the process's libm; else null */, float* out, hipStream_t s);
hipError_t np_launch_selftest_div(uint64_t n_samples, uint64_t seed, unsigned long long* d_mismatches, hipStream_t s);
hipError_t np_launch_selftest_ratio(uint64_t n_samples, uint64_t seed, unsigned long long* d_out, hipStream_t s);
hipError_t np_launch_selftest_div_small(int w, double chd, double cld, float chf, float clf, uint64_t n_f64, unsigned long long* d_out, hipStream_t s);

// ---- f2: event detection + method-of-moments scalings (np_events_kernels.hip) ------------------------------------
hipError_t np_launch_detect_events(int n_reads, const float* raw, const int64_t* raw_off, int64_t max_samples, const np_detector_param& p,
                                   float2* tstat, int32_t* status, const int64_t* event_off, int64_t max_events, uint32_t* event_start,
                                   float* event_length, float* event_mean, float* event_stdv, int32_t* n_events, int warmup /* < 0: default */,
                                   bool checked /* status already holds these samples' exactness verdicts (np_launch_adc_to_pa) */, hipStream_t s, int ratio_exact = 0);
hipError_t np_launch_detect_events_adc(int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples, const float* offset, const float* raw_unit,
                                       float* raw_pa, const np_detector_param& p, float2* tstat, int32_t* status, const int64_t* event_off, int64_t max_events,
                                       uint32_t* event_start, float* event_length, float* event_mean, float* event_stdv, int32_t* n_events, int warmup, hipStream_t s, int ratio_exact = 0);
hipError_t np_launch_adc_to_pa(int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples, const float* offset,
                               const float* raw_unit, float* raw_pa, int32_t* status /* optional */, hipStream_t s);
hipError_t np_launch_reverse_events(int n_reads, const int64_t* event_off, const int32_t* n_events, uint32_t* start, float* length, float* mean,
                                    float* stdv, hipStream_t s);
hipError_t np_launch_mom_fill(int n_reads, np_read_dev* reads, np_read_dev* reads_b, const float* event_mean, const int32_t* n_events,
                              const uint16_t* ranks, const np_state_dev* model, int n_states, hipStream_t s);

// ---- f3: call-methylation work items of identity-aligned reads (np_jobs_kernels.hip) -----------------------------------
hipError_t np_launch_cm_build_jobs(int n_reads, const char* seq, const int64_t* seq_off, const uint8_t* read_rc, int alphabet, int k,
                                   int min_separation, int min_flank, const int64_t* group_off, const int64_t* rank_off_cap,
                                   np_hmm_job_dev* jobs, int32_t* kpos, uint16_t* job_ranks, int32_t* first_site, int32_t* last_site,
                                   int32_t* n_motif, int64_t* group_rank_off, int32_t* n_groups, int write_unused, hipStream_t s);
hipError_t np_launch_eventalign_chain(const np_ea_args& a, const np_ea_args* a_dev, int n_blocks, int variant, hipStream_t s);
int np_eventalign_line_bytes(int variant);
int np_eventalign_max_kmers(int variant);
hipError_t np_launch_cigar_index(int n_reads, const uint32_t* cigar, const int64_t* cigar_off, const int32_t* read_len, int k, int32_t* op_ref,
                                 int32_t* op_read, void* cig_reads, hipStream_t s);
hipError_t np_launch_cm_build_jobs_cigar(int n_reads, const char* genome, const int64_t* ref_begin, const int32_t* ref_len,
                                         const uint32_t* cigar, const int64_t* cigar_off, const int32_t* read_len, const uint8_t* read_rc,
                                         int alphabet, int k, int min_separation, int min_flank, const int64_t* group_off,
                                         const int64_t* rank_off_cap, np_hmm_job_dev* jobs, int32_t* kpos, uint16_t* job_ranks,
                                         int32_t* first_site, int32_t* last_site, int32_t* n_motif, int64_t* group_rank_off,
                                         int32_t* n_groups, int32_t* deg_kpos, int32_t* op_ref, int32_t* op_read, void* cig_reads,
                                         int32_t* group_kpos, int write_unused, hipStream_t s);
