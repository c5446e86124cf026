/* oracle/np_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See np_oracle.h.
 *
 * Plain-C restatement of the nanopolish (v0.14.0) signal-HMM hot path.  It keeps the reference's
 * data layout (full lattices, flat band/trace arrays) and evaluation order on purpose: clarity and
 * bit-equality matter here, speed does not.  Compile like the reference: -O3, baseline x86-64,
 * -ffp-contract=off (the reference Makefile:12-13 never enables FMA).
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py compares the HMM / aligner / detector functions below with the
 * reference's own code (oracle/_ref/libnp_ref.so) bit-for-bit on seeded inputs; tests/test_oracle_vs_ref_full.py does the
 * same for the read-level helpers (npo_build_base_to_event_map, npo_get_closest_event_to, npo_cigar_aligned_bases,
 * npo_event_alignment_record, npo_find_by_ref_bounds, npo_scan_motif_groups, npo_recalibrate) against the reference's
 * SquiggleRead::load_from_raw, EventAlignmentRecord, calculate_methylation_for_read and align_read_to_ref compiled in
 * place (oracle/_ref/libnp_ref_full.so: htslib's record layout and accessors stood in for from the SAM specification,
 * Eigen's 2x2 full-pivot LU restated in oracle/stubs_full/Eigen/Dense -- the one step that is a restatement on BOTH sides).
 * tests/test_oracle_golden.py and tests/test_reflevel_golden.py re-check the committed vectors that code produced
 * (tests/golden/) where the reference is not available (GPU box).
 */
#include "np_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* =====================================================================================
 * Alphabets -- src/common/nanopolish_alphabet.h:59-253, tables src/common/nanopolish_alphabet.cpp
 * ===================================================================================== */
typedef struct {
    const char* name;
    const char* base;
    const char* complement;
    uint32_t size;
    uint32_t n_sites;
    uint32_t site_len;
    const char* sites[2];
    const char* sites_meth[2];
    const char* sites_meth_comp[2];
} alpha_t;

static const alpha_t ALPHA[NPO_NUM_ALPHABETS] = {
    /* nanopolish_alphabet.cpp:17-39  */ { "nucleotide", "ACGT",  "TGCA",  4, 0, 0, {0,0}, {0,0}, {0,0} },
    /* :67-95   */ { "cpg", "ACGMT", "TGCGA", 5, 1, 2, {"CG",0}, {"MG",0}, {"GM",0} },
    /* :97-125  */ { "gpc", "ACGMT", "TGCGA", 5, 1, 2, {"GC",0}, {"GM",0}, {"MG",0} },
    /* :127-155 */ { "dam", "ACGMT", "TGCTA", 5, 1, 4, {"GATC",0}, {"GMTC",0}, {"CTMG",0} },
    /* :157-186 */ { "dcm", "ACGMT", "TGCGA", 5, 2, 5, {"CCAGG","CCTGG"}, {"CMAGG","CMTGG"}, {"GGTMC","GGAMC"} },
    /* :41-65   */ { "u_to_t_rna", "ACGT", "TGCA", 4, 0, 0, {0,0}, {0,0}, {0,0} },
};

int npo_alphabet_id(const char* name)
{
    for(int i = 0; i < NPO_NUM_ALPHABETS; ++i) if(strcmp(name, ALPHA[i].name) == 0) return i;
    return -1;
}
uint32_t npo_alphabet_size(int a) { return ALPHA[a].size; }

/* _rank[256] tables: every byte ranks 0 except the alphabet's own symbols */
static inline uint8_t a_rank(int a, char b)
{
    if(ALPHA[a].size == 4) { switch(b) { case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 0; } }
    switch(b) { case 'C': return 1; case 'G': return 2; case 'M': return 3; case 'T': return 4; default: return 0; }
}
static inline char a_complement(int a, char b) { return ALPHA[a].complement[a_rank(a, b)]; }

/* Alphabet::kmer_rank, nanopolish_alphabet.h:78-89 */
uint32_t npo_kmer_rank(int a, const char* str, uint32_t k)
{
    uint32_t p = 1, r = 0;
    for(uint32_t i = 0; i < k; ++i) {
        r += a_rank(a, str[k - i - 1]) * p;
        p *= ALPHA[a].size;
    }
    return r;
}

typedef struct { unsigned offset, length; int covers; } rmatch_t;

/* match_to_site, nanopolish_alphabet.h:27-56.  str has n characters. */
static rmatch_t match_to_site(const char* str, int n, int i, const char* recognition, int rl)
{
    rmatch_t m; m.offset = 0; m.length = 0; m.covers = 0;
    /* Case 1: the whole of str is a substring of recognition (strstr(recognition, str)) */
    int p = -1;
    if(i == 0) {
        if(n == 0) p = 0;
        for(int o = 0; p < 0 && o + n <= rl; ++o) if(strncmp(recognition + o, str, n) == 0) p = o;
    }
    if(i == 0 && p >= 0) {
        m.offset = p; m.length = n;
    } else {
        /* Case 2: the suffix str[i..n) is a prefix of recognition */
        int cl = rl < n - i ? rl : n - i;
        if(strncmp(str + i, recognition, cl) == 0) { m.offset = 0; m.length = cl; }
    }
    if(m.length > 0) {
        for(unsigned j = 0; j < m.length; ++j) if(str[i + j] == 'M') m.covers = 1;
    }
    return m;
}

/* Alphabet::reverse_complement, nanopolish_alphabet.h:118-150 */
void npo_reverse_complement(int a, const char* in, int n, char* out)
{
    const alpha_t* A = &ALPHA[a];
    int i = 0, j = n - 1;
    while(i < n) {
        int recognition_index = -1;
        rmatch_t match; match.offset = match.length = 0; match.covers = 0;
        for(uint32_t k = 0; k < A->n_sites; ++k) {
            match = match_to_site(in, n, i, A->sites_meth[k], A->site_len);
            if(match.length > 0 && match.covers) { recognition_index = k; break; }
        }
        if(recognition_index != -1) {
            for(unsigned k = match.offset; k < match.offset + match.length; ++k) {
                out[j--] = A->sites_meth_comp[recognition_index][k];
                i += 1;
            }
        } else {
            out[j--] = a_complement(a, in[i++]);
        }
    }
    out[n] = 0;
}

/* Alphabet::methylate, nanopolish_alphabet.h:189-212 */
void npo_methylate(int a, const char* in, int n, char* out)
{
    const alpha_t* A = &ALPHA[a];
    memcpy(out, in, n); out[n] = 0;
    int i = 0;
    while(i < n) {
        int stride = 1;
        for(uint32_t j = 0; j < A->n_sites; ++j) {
            rmatch_t match = match_to_site(in, n, i, A->sites[j], A->site_len);
            if(match.length == A->site_len) {
                memcpy(out + i, A->sites_meth[j], A->site_len);
                stride = match.length;
                break;
            }
        }
        i += stride;
    }
}

/* Alphabet::unmethylate, nanopolish_alphabet.h:215-238 */
void npo_unmethylate(int a, const char* in, int n, char* out)
{
    const alpha_t* A = &ALPHA[a];
    memcpy(out, in, n); out[n] = 0;
    int i = 0;
    while(i < n) {
        int stride = 1;
        for(uint32_t j = 0; j < A->n_sites; ++j) {
            rmatch_t match = match_to_site(in, n, i, A->sites_meth[j], A->site_len);
            if(match.length > 0) {
                memcpy(out + i, A->sites[j] + match.offset, match.length);
                stride = match.length;
                break;
            }
        }
        i += stride;
    }
}

/* Alphabet::is_motif_match, nanopolish_alphabet.h:244-253 */
int npo_is_motif_match(int a, const char* str, int n, int i)
{
    const alpha_t* A = &ALPHA[a];
    for(uint32_t j = 0; j < A->n_sites; ++j) {
        rmatch_t match = match_to_site(str, n, i, A->sites[j], A->site_len);
        if(match.length == A->site_len) return 1;
    }
    return 0;
}

/* HMMInputSequence::get_kmer_rank, src/hmm/nanopolish_hmm_input_sequence.h:76-91 */
void npo_sequence_kmer_ranks(int a, const char* seq, const char* rc_seq, int n, int k, int do_rc, uint32_t* out)
{
    char* tmp = NULL;
    if(do_rc && rc_seq == NULL) { tmp = (char*)malloc(n + 1); npo_reverse_complement(a, seq, n, tmp); rc_seq = tmp; }
    int n_kmers = n - k + 1;
    for(int i = 0; i < n_kmers; ++i)
        out[i] = !do_rc ? npo_kmer_rank(a, seq + i, k) : npo_kmer_rank(a, rc_seq + (n - i - k), k);
    free(tmp);
}

/* =====================================================================================
 * p7_FLogsum -- src/common/logsum.h:55-66, table src/common/logsum.cpp:57-69
 * ===================================================================================== */
static float g_flogsum[NPO_LOGSUM_TBL];

/* static initialisation, as logsum.cpp:96-97 (Init_Caller) */
__attribute__((constructor)) static void npo_flogsum_init(void)
{
    for(int i = 0; i < NPO_LOGSUM_TBL; i++)
        g_flogsum[i] = log(1. + exp((double) -i / 1000.f));   /* logsum.cpp:65 */
    if((float)log(0.3989422804014327) != -0x1.d67f1cp-1f) { fprintf(stderr, "npo: log_inv_sqrt_2pi literal mismatch\n"); abort(); }
}

const float* npo_flogsum_table(void) { return g_flogsum; }

static inline float flogsum_inl(float a, float b)
{
    const float max = a > b ? a : b;  /* ESL_MAX */
    const float min = a < b ? a : b;  /* ESL_MIN */
    return (min == -INFINITY || (max - min) >= 15.7f) ? max : max + g_flogsum[(int)((max - min) * 1000.f)];
}

float npo_flogsum(float a, float b) { return flogsum_inl(a, b); }

/* add_logs, src/common/nanopolish_common.h:97-104: double in, p7_FLogsum(float,float), double out */
static inline double add_logs(double a, double b) { return flogsum_inl((float)a, (float)b); }

/* =====================================================================================
 * Scalings + emission
 * ===================================================================================== */
/* SquiggleScalings::set4/set6, src/nanopolish_squiggle_read.cpp:38-65 */
npo_scalings npo_set4(double shift, double scale, double drift, double var)
{
    npo_scalings s; s.shift = shift; s.scale = scale; s.drift = drift; s.var = var; s.log_var = log(var);
    return s;
}

/* log_probability_match_r9, src/hmm/nanopolish_emissions.h:57-68
 *  + get_drift_scaled_level            src/nanopolish_squiggle_read.h:149-154
 *  + get_scaled_gaussian_from_pore_model_state  src/nanopolish_squiggle_read.h:217-226
 *  + log_normal_pdf                    src/hmm/nanopolish_emissions.h:51-55 */
float npo_log_probability_match_r9(const npo_model* m, const npo_scalings* s, uint32_t rank, float level_in, float time)
{
    /* static const float log_inv_sqrt_2pi = log(0.3989422804014327);  emissions.h:43 (double log -> float).
     * Kept as a literal so the oracle does not pay a libm call per emission; checked against libm in
     * npo_flogsum_table(). */
    const float log_inv_sqrt_2pi = -0x1.d67f1cp-1f;
    float level = level_in - time * s->drift;                        /* float - (float*double) -> double -> float */
    float gp_mean = s->scale * m->level_mean[rank] + s->shift;       /* double math, float store */
    float gp_stdv = m->level_stdv[rank] * s->var;
    float gp_log_stdv = m->level_log_stdv[rank] + s->log_var;
    float a = (level - gp_mean) / gp_stdv;
    return log_inv_sqrt_2pi - gp_log_stdv + (-0.5f * a * a);
}

/* =====================================================================================
 * Profile HMM -- src/hmm/nanopolish_profile_hmm_r9.{h,cpp,inl}
 * ===================================================================================== */
enum { PSR9_KMER_SKIP = 0, PSR9_BAD_EVENT, PSR9_MATCH, PSR9_NUM_STATES = 3 };        /* r9.h:52-59 */
enum { HMT_FROM_SAME_M = 0, HMT_FROM_PREV_M, HMT_FROM_SAME_B, HMT_FROM_PREV_B, HMT_FROM_PREV_K, HMT_FROM_SOFT, HMT_NUM = 6 }; /* r9.h:61-70 */

/* calculate_transitions, r9.inl:17-76 (identical for every k-mer) */
void npo_calculate_transitions(double events_per_base, double indel_bias, float out[10])
{
    double read_events_per_base = events_per_base;
    read_events_per_base *= indel_bias;
    read_events_per_base = read_events_per_base > 1.25 ? read_events_per_base : 1.25;  /* std::max(1.25, x) */

    float p_stay = 1 - (1 / read_events_per_base);
    float p_skip = 0.0025;
    float p_bad = 0.001;
    float p_bad_self = p_bad;
    float p_skip_self = 0.3;

    float p_mk = p_skip;
    float p_mb = p_bad;
    float p_mm_self = p_stay;
    float p_mm_next = 1.0f - p_mm_self - p_mk - p_mb;

    float p_bb = p_bad_self;
    float p_bk, p_bm_next, p_bm_self;
    p_bk = p_bm_next = p_bm_self = (1.0f - p_bb) / 3;

    float p_kk = p_skip_self;
    float p_km = 1.0f - p_kk;

    /* log(float) resolves to the float overload in the reference TU (<cmath>) */
    out[0] = logf(p_mm_self);  /* lp_mm_self */
    out[1] = logf(p_mb);       /* lp_mb      */
    out[2] = logf(p_mk);       /* lp_mk      */
    out[3] = logf(p_mm_next);  /* lp_mm_next */
    out[4] = logf(p_bb);       /* lp_bb      */
    out[5] = logf(p_bk);       /* lp_bk      */
    out[6] = logf(p_bm_next);  /* lp_bm_next */
    out[7] = logf(p_bm_self);  /* lp_bm_self */
    out[8] = logf(p_kk);       /* lp_kk      */
    out[9] = logf(p_km);       /* lp_km      */
}

/* make_pre_flanking / make_post_flanking, r9.inl:200-260; background emission -3.0f (emissions.h:98-103) */
#define TRANS_CLIP_SELF 0.9
#define TRANS_START_TO_CLIP 0.5
void npo_make_flanks(uint32_t num_events, float* pre_flank, float* post_flank)
{
    const float bg = -3.0f;
    /* pre: r9.inl:204-226 */
    pre_flank[0] = log(1 - TRANS_START_TO_CLIP);
    pre_flank[1] = log(TRANS_START_TO_CLIP) + bg + log(1 - TRANS_CLIP_SELF);
    for(uint32_t i = 2; i < num_events + 1; ++i)
        pre_flank[i] = log(TRANS_CLIP_SELF) + bg + pre_flank[i - 1];
    /* post: r9.inl:236-259 */
    post_flank[num_events - 1] = log(1 - TRANS_START_TO_CLIP);
    if(num_events > 1) {
        post_flank[num_events - 2] = log(TRANS_START_TO_CLIP) + bg + log(1 - TRANS_CLIP_SELF);
        for(int i = (int)num_events - 3; i >= 0; --i)
            post_flank[i] = log(TRANS_CLIP_SELF) + bg + post_flank[i + 1];
    }
}

typedef struct {
    int viterbi;
    uint32_t n_rows, n_cols;
    float* fm;        /* FloatMatrix row-major (nanopolish_matrix.h:72-85) */
    uint8_t* bm;      /* UInt8Matrix, Viterbi only */
    float lp_end;
} hmm_out_t;

#define FM(o, r, c) ((o)->fm[(size_t)(r) * (o)->n_cols + (c)])
#define BM(o, r, c) ((o)->bm[(size_t)(r) * (o)->n_cols + (c)])

/* ProfileHMMForwardOutputR9::update_cell r9.inl:85-93 / ProfileHMMViterbiOutputR9::update_cell r9.inl:135-147 */
static inline void update_cell(hmm_out_t* o, uint32_t row, uint32_t col, const float x[HMT_NUM], float lp_emission)
{
    if(!o->viterbi) {
        float sum = x[0];
        for(int i = 1; i < HMT_NUM; ++i) sum = add_logs(sum, x[i]);
        sum += lp_emission;
        FM(o, row, col) = sum;
    } else {
        float max = x[0];
        uint8_t from = 0;
        for(int i = 1; i < HMT_NUM; ++i) {
            max = x[i] > max ? x[i] : max;
            from = max == x[i] ? i : from;
        }
        FM(o, row, col) = max + lp_emission;
        BM(o, row, col) = from;
    }
}

/* update_end: forward r9.inl:96-99, viterbi r9.inl:150-157 */
static inline void update_end(hmm_out_t* o, float v)
{
    if(!o->viterbi) o->lp_end = add_logs(o->lp_end, v);
    else if(v > o->lp_end) o->lp_end = v;
}

/* profile_hmm_fill_generic_r9, r9.inl:265-433 */
static float hmm_fill(hmm_out_t* o, const npo_model* m, const npo_scalings* s, const float* event_mean,
                      const uint32_t* kmer_ranks, uint32_t e_start, int stride,
                      double events_per_base, double indel_bias, uint32_t flags)
{
    uint32_t num_blocks = o->n_cols / PSR9_NUM_STATES;
    uint32_t last_event_row_idx = o->n_rows - 1;
    uint32_t num_kmers = num_blocks - 2;
    uint32_t last_kmer_idx = num_kmers - 1;

    float bt[10];
    npo_calculate_transitions(events_per_base, indel_bias, bt);
    const float lp_mm_self = bt[0], lp_mb = bt[1], lp_mk = bt[2], lp_mm_next = bt[3], lp_bb = bt[4],
                lp_bk = bt[5], lp_bm_next = bt[6], lp_bm_self = bt[7], lp_kk = bt[8], lp_km = bt[9];

    uint32_t num_events = o->n_rows - 1;
    float* pre_flank = (float*)malloc(sizeof(float) * (num_events + 1));
    float* post_flank = (float*)malloc(sizeof(float) * num_events);
    npo_make_flanks(num_events, pre_flank, post_flank);

    float lp_sm, lp_ms;
    lp_sm = lp_ms = 0.0f;
    float BAD_EVENT_PENALTY = 0.0f;

    for(uint32_t row = 1; row < o->n_rows; row++) {
        for(uint32_t block = 1; block < num_blocks - 1; block++) {
            uint32_t kmer_idx = block - 1;
            uint32_t prev_block = block - 1;
            uint32_t prev_block_offset = PSR9_NUM_STATES * prev_block;
            uint32_t curr_block_offset = PSR9_NUM_STATES * block;

            uint32_t event_idx = e_start + (row - 1) * stride;
            uint32_t rank = kmer_ranks[kmer_idx];
            float lp_emission_m = npo_log_probability_match_r9(m, s, rank, event_mean[event_idx], 0.0f);
            float lp_emission_b = BAD_EVENT_PENALTY;

            float x[HMT_NUM];
            /* PSR9_MATCH, r9.inl:350-365 */
            x[HMT_FROM_SAME_M] = lp_mm_self + FM(o, row - 1, curr_block_offset + PSR9_MATCH);
            x[HMT_FROM_PREV_M] = lp_mm_next + FM(o, row - 1, prev_block_offset + PSR9_MATCH);
            x[HMT_FROM_SAME_B] = lp_bm_self + FM(o, row - 1, curr_block_offset + PSR9_BAD_EVENT);
            x[HMT_FROM_PREV_B] = lp_bm_next + FM(o, row - 1, prev_block_offset + PSR9_BAD_EVENT);
            x[HMT_FROM_PREV_K] = lp_km + FM(o, row - 1, prev_block_offset + PSR9_KMER_SKIP);
            x[HMT_FROM_SOFT] = (kmer_idx == 0 && (event_idx == e_start || (flags & NPO_HAF_ALLOW_PRE_CLIP)))
                                   ? lp_sm + pre_flank[row - 1] : -INFINITY;
            update_cell(o, row, curr_block_offset + PSR9_MATCH, x, lp_emission_m);

            /* PSR9_BAD_EVENT, r9.inl:368-374 */
            x[HMT_FROM_SAME_M] = lp_mb + FM(o, row - 1, curr_block_offset + PSR9_MATCH);
            x[HMT_FROM_PREV_M] = -INFINITY;
            x[HMT_FROM_SAME_B] = lp_bb + FM(o, row - 1, curr_block_offset + PSR9_BAD_EVENT);
            x[HMT_FROM_PREV_B] = -INFINITY;
            x[HMT_FROM_PREV_K] = -INFINITY;
            x[HMT_FROM_SOFT] = -INFINITY;
            update_cell(o, row, curr_block_offset + PSR9_BAD_EVENT, x, lp_emission_b);

            /* PSR9_KMER_SKIP, r9.inl:377-383 */
            x[HMT_FROM_SAME_M] = -INFINITY;
            x[HMT_FROM_PREV_M] = lp_mk + FM(o, row, prev_block_offset + PSR9_MATCH);
            x[HMT_FROM_SAME_B] = -INFINITY;
            x[HMT_FROM_PREV_B] = lp_bk + FM(o, row, prev_block_offset + PSR9_BAD_EVENT);
            x[HMT_FROM_PREV_K] = lp_kk + FM(o, row, prev_block_offset + PSR9_KMER_SKIP);
            x[HMT_FROM_SOFT] = -INFINITY;
            update_cell(o, row, curr_block_offset + PSR9_KMER_SKIP, x, 0.0f);

            /* end state, r9.inl:388-396 */
            if(kmer_idx == last_kmer_idx && ((flags & NPO_HAF_ALLOW_POST_CLIP) || row == last_event_row_idx)) {
                float lp1 = lp_ms + FM(o, row, curr_block_offset + PSR9_MATCH) + post_flank[row - 1];
                float lp2 = lp_ms + FM(o, row, curr_block_offset + PSR9_BAD_EVENT) + post_flank[row - 1];
                float lp3 = lp_ms + FM(o, row, curr_block_offset + PSR9_KMER_SKIP) + post_flank[row - 1];
                update_end(o, lp1);
                update_end(o, lp2);
                update_end(o, lp3);
            }
        }
    }
    free(pre_flank);
    free(post_flank);
    return o->lp_end;
}

/* allocate_matrix (nanopolish_matrix.h:36-44: malloc + memset 0) + profile_hmm_forward_initialize_r9 (r9.cpp:21-33) */
static void hmm_alloc_init(hmm_out_t* o, uint32_t n_rows, uint32_t n_cols, int viterbi)
{
    o->viterbi = viterbi; o->n_rows = n_rows; o->n_cols = n_cols; o->lp_end = -INFINITY;
    o->fm = (float*)calloc((size_t)n_rows * n_cols, sizeof(float));
    o->bm = viterbi ? (uint8_t*)calloc((size_t)n_rows * n_cols, 1) : NULL;
    for(uint32_t si = 0; si < n_cols; si++) FM(o, 0, si) = -INFINITY;
    for(uint32_t ri = 0; ri < n_rows; ri++) {
        FM(o, ri, PSR9_KMER_SKIP) = -INFINITY;
        FM(o, ri, PSR9_BAD_EVENT) = -INFINITY;
        FM(o, ri, PSR9_MATCH) = -INFINITY;
    }
}

/* profile_hmm_score_r9, r9.cpp:35-65 */
float npo_profile_hmm_score(const npo_model* m, const npo_scalings* s, const float* event_mean,
                            const uint32_t* kmer_ranks, uint32_t n_kmers,
                            uint32_t e_start, uint32_t e_stop, int stride,
                            double events_per_base, double indel_bias, uint32_t flags)
{
    uint32_t n_states = PSR9_NUM_STATES * (n_kmers + 2);
    uint32_t n_events = e_stop > e_start ? e_stop - e_start + 1 : e_start - e_stop + 1;
    uint32_t n_rows = n_events + 1;
    hmm_out_t o;
    hmm_alloc_init(&o, n_rows, n_states, 0);
    float score = hmm_fill(&o, m, s, event_mean, kmer_ranks, e_start, stride, events_per_base, indel_bias, flags);
    free(o.fm);
    return score;
}

/* profile_hmm_align_r9, r9.cpp:73-204 */
int npo_profile_hmm_align(const npo_model* m, const npo_scalings* s, const float* event_mean,
                          const uint32_t* kmer_ranks, uint32_t n_kmers,
                          uint32_t e_start, uint32_t e_stop, int stride,
                          double events_per_base, double indel_bias, uint32_t flags,
                          uint32_t* out_event_idx, uint32_t* out_kmer_idx, double* out_l_fm, char* out_state, int cap)
{
    uint32_t n_states = PSR9_NUM_STATES * (n_kmers + 2);
    uint32_t n_events = e_stop > e_start ? e_stop - e_start + 1 : e_start - e_stop + 1;
    if(n_events < 2) return -1;                      /* assert(n_events >= 2), r9.cpp:88 */
    uint32_t n_rows = n_events + 1;
    hmm_out_t o;
    hmm_alloc_init(&o, n_rows, n_states, 1);
    hmm_fill(&o, m, s, event_mean, kmer_ranks, e_start, stride, events_per_base, indel_bias, flags);

    int n = 0, err = 0;
    uint32_t row = n_rows - 1;
    uint32_t col = PSR9_NUM_STATES * n_kmers + PSR9_MATCH;      /* r9.cpp:117-118 */
    while(row > 0) {
        uint32_t event_idx = e_start + (row - 1) * stride;
        uint32_t block = col / PSR9_NUM_STATES;
        uint32_t kmer_idx = block - 1;
        int curr_ps = col % PSR9_NUM_STATES;
        if(block == 0 || FM(&o, row, col) == -INFINITY) { err = 1; break; }   /* asserts r9.cpp:131-132 */
        if(n < cap) {
            out_event_idx[n] = event_idx; out_kmer_idx[n] = kmer_idx;
            out_l_fm[n] = FM(&o, row, col); out_state[n] = "KBMNS"[curr_ps];
        }
        n++;
        int movement = BM(&o, row, col);
        if(movement == HMT_FROM_SOFT) break;
        int next_ps = PSR9_MATCH;
        switch(movement) {
            case HMT_FROM_SAME_M: next_ps = PSR9_MATCH; break;
            case HMT_FROM_PREV_M: kmer_idx -= 1; next_ps = PSR9_MATCH; break;
            case HMT_FROM_SAME_B: next_ps = PSR9_BAD_EVENT; break;
            case HMT_FROM_PREV_B: kmer_idx -= 1; next_ps = PSR9_BAD_EVENT; break;
            case HMT_FROM_PREV_K: kmer_idx -= 1; next_ps = PSR9_KMER_SKIP; break;
        }
        if(curr_ps != PSR9_KMER_SKIP) row -= 1;
        col = PSR9_NUM_STATES * (kmer_idx + 1) + next_ps;
    }
    free(o.fm); free(o.bm);
    if(err) return -1;
    /* std::reverse, r9.cpp:196 */
    int lim = n < cap ? n : cap;
    for(int i = 0, j = lim - 1; i < j; ++i, --j) {
        uint32_t te = out_event_idx[i]; out_event_idx[i] = out_event_idx[j]; out_event_idx[j] = te;
        uint32_t tk = out_kmer_idx[i]; out_kmer_idx[i] = out_kmer_idx[j]; out_kmer_idx[j] = tk;
        double tf = out_l_fm[i]; out_l_fm[i] = out_l_fm[j]; out_l_fm[j] = tf;
        char ts = out_state[i]; out_state[i] = out_state[j]; out_state[j] = ts;
    }
    return n;
}

/* profile_hmm_score_set, src/hmm/nanopolish_profile_hmm.cpp:32-56 */
float npo_combine_score_set(const float* scores, int n)
{
    double num_model_penalty = log((double)(size_t)n);
    double score = scores[0] - num_model_penalty;
    for(int i = 1; i < n; ++i) {
        double alt_score = scores[i] - num_model_penalty;
        score = add_logs(score, alt_score);
    }
    return score;
}

/* =====================================================================================
 * Raw loader -- src/nanopolish_raw_loader.cpp
 * ===================================================================================== */
/* estimate_scalings_using_mom, raw_loader.cpp:17-60 */
void npo_estimate_scalings_mom(const npo_model* m, const uint32_t* kmer_ranks, uint32_t n_kmers,
                               const float* event_mean, uint32_t n_events, double* shift_out, double* scale_out)
{
    double event_level_sum = 0.0f;
    for(uint32_t i = 0; i < n_events; ++i) event_level_sum += event_mean[i];

    double kmer_level_sum = 0.0f;
    double kmer_level_sq_sum = 0.0f;
    for(uint32_t i = 0; i < n_kmers; ++i) {
        double l = m->level_mean[kmer_ranks[i]];
        kmer_level_sum += l;
        kmer_level_sq_sum += pow(l, 2.0f);
    }
    double shift = event_level_sum / n_events - kmer_level_sum / n_kmers;

    double event_level_sq_sum = 0.0f;
    for(uint32_t i = 0; i < n_events; ++i) event_level_sq_sum += pow(event_mean[i] - shift, 2.0);

    double scale = (event_level_sq_sum / n_events) / (kmer_level_sq_sum / n_kmers);
    *shift_out = shift; *scale_out = scale;
}

/* adaptive_banded_simple_event_align, raw_loader.cpp:77-379 */
#define ALN_BANDWIDTH 100
int npo_adaptive_banded_simple_event_align(const npo_model* m, const npo_scalings* s,
                                           const float* event_mean, uint32_t n_events_u,
                                           const uint32_t* kmer_ranks, uint32_t n_kmers_u,
                                           int32_t* out_pairs, int cap)
{
    const size_t n_events = n_events_u, n_kmers = n_kmers_u;
    const uint8_t FROM_D = 0, FROM_U = 1, FROM_L = 2;
    double min_average_log_emission = -5.0;
    int max_gap_threshold = 50;
    int bandwidth = ALN_BANDWIDTH;
    int half_bandwidth = bandwidth / 2;

    /* transition penalties, :99-108 (all double) */
    double events_per_kmer = (double)n_events / n_kmers;
    double p_stay = 1 - (1 / (events_per_kmer + 1));
    double epsilon = 1e-10;
    double lp_skip = log(epsilon);
    double lp_stay = log(p_stay);
    double lp_step = log(1.0 - exp(lp_skip) - exp(lp_stay));
    double lp_trim = log(0.01);

    size_t n_rows = n_events + 1;
    size_t n_cols = n_kmers + 1;
    size_t n_bands = n_rows + n_cols;

    float* bands = (float*)malloc(sizeof(float) * n_bands * bandwidth);
    uint8_t* trace = (uint8_t*)malloc(sizeof(uint8_t) * n_bands * bandwidth);
    int* ll_event = (int*)malloc(sizeof(int) * n_bands);   /* band_lower_left[].event_idx */
    int* ll_kmer = (int*)malloc(sizeof(int) * n_bands);    /* band_lower_left[].kmer_idx  */
    if(!bands || !trace || !ll_event || !ll_kmer) { fprintf(stderr, "npo: allocation failed\n"); exit(1); }
#define BAND_ARRAY(r, c) (bands[((size_t)(r) * (ALN_BANDWIDTH) + (c))])
#define TRACE_ARRAY(r, c) (trace[((size_t)(r) * (ALN_BANDWIDTH) + (c))])
    for(size_t i = 0; i < n_bands; i++)
        for(int j = 0; j < bandwidth; j++) { BAND_ARRAY(i, j) = -INFINITY; TRACE_ARRAY(i, j) = 0; }

    /* first two bands, :152-167 */
    ll_event[0] = half_bandwidth - 1;
    ll_kmer[0] = -1 - half_bandwidth;
    ll_event[1] = ll_event[0] + 1; ll_kmer[1] = ll_kmer[0];             /* move_down */

    int start_cell_offset = (-1) - ll_kmer[0];
    BAND_ARRAY(0, start_cell_offset) = 0.0f;
    int first_trim_offset = ll_event[1] - 0;
    BAND_ARRAY(1, first_trim_offset) = lp_trim;
    TRACE_ARRAY(1, first_trim_offset) = FROM_U;

    /* fill, :176-290 */
    for(int band_idx = 2; band_idx < (int)n_bands; ++band_idx) {
        float ll = BAND_ARRAY(band_idx - 1, 0);
        float ur = BAND_ARRAY(band_idx - 1, bandwidth - 1);
        int ll_ob = ll == -INFINITY;
        int ur_ob = ur == -INFINITY;
        int right;
        if(ll_ob && ur_ob) right = band_idx % 2 == 1;
        else right = ll < ur;
        if(right) { ll_event[band_idx] = ll_event[band_idx - 1];     ll_kmer[band_idx] = ll_kmer[band_idx - 1] + 1; }
        else      { ll_event[band_idx] = ll_event[band_idx - 1] + 1; ll_kmer[band_idx] = ll_kmer[band_idx - 1]; }

        /* trim column, :216-225 */
        int trim_offset = (-1) - ll_kmer[band_idx];
        if(trim_offset >= 0 && trim_offset < bandwidth) {
            int event_idx = ll_event[band_idx] - trim_offset;
            if(event_idx >= 0 && event_idx < (int)n_events) {
                BAND_ARRAY(band_idx, trim_offset) = lp_trim * (event_idx + 1);
                TRACE_ARRAY(band_idx, trim_offset) = FROM_U;
            } else {
                BAND_ARRAY(band_idx, trim_offset) = -INFINITY;
            }
        }

        /* inner-loop limits, :229-238 */
        int kmer_min_offset = 0 - ll_kmer[band_idx];
        int kmer_max_offset = (int)n_kmers - ll_kmer[band_idx];
        int event_min_offset = ll_event[band_idx] - ((int)n_events - 1);
        int event_max_offset = ll_event[band_idx] - (-1);
        int min_offset = kmer_min_offset > event_min_offset ? kmer_min_offset : event_min_offset;
        min_offset = min_offset > 0 ? min_offset : 0;
        int max_offset = kmer_max_offset < event_max_offset ? kmer_max_offset : event_max_offset;
        max_offset = max_offset < bandwidth ? max_offset : bandwidth;

        for(int offset = min_offset; offset < max_offset; ++offset) {
            int event_idx = ll_event[band_idx] - offset;
            int kmer_idx = ll_kmer[band_idx] + offset;
            uint32_t kmer_rank = kmer_ranks[kmer_idx];

            int offset_up   = ll_event[band_idx - 1] - (event_idx - 1);
            int offset_left = (kmer_idx - 1) - ll_kmer[band_idx - 1];
            int offset_diag = (kmer_idx - 1) - ll_kmer[band_idx - 2];

            float up   = (offset_up >= 0 && offset_up < bandwidth)     ? BAND_ARRAY(band_idx - 1, offset_up)   : -INFINITY;
            float left = (offset_left >= 0 && offset_left < bandwidth) ? BAND_ARRAY(band_idx - 1, offset_left) : -INFINITY;
            float diag = (offset_diag >= 0 && offset_diag < bandwidth) ? BAND_ARRAY(band_idx - 2, offset_diag) : -INFINITY;

            float lp_emission = npo_log_probability_match_r9(m, s, kmer_rank, event_mean[event_idx], 0.0f);
            float score_d = diag + lp_step + lp_emission;   /* (double)diag + lp_step + (double)em -> float */
            float score_u = up + lp_stay + lp_emission;
            float score_l = left + lp_skip;

            float max_score = score_d;
            uint8_t from = FROM_D;
            max_score = score_u > max_score ? score_u : max_score;
            from = max_score == score_u ? FROM_U : from;
            max_score = score_l > max_score ? score_l : max_score;
            from = max_score == score_l ? FROM_L : from;

            BAND_ARRAY(band_idx, offset) = max_score;
            TRACE_ARRAY(band_idx, offset) = from;
        }
    }

    /* backtrack, :300-361 */
    double sum_emission = 0;
    double n_aligned_events = 0;
    int n_out = 0, ub = 0;

    float max_score = -INFINITY;
    int curr_event_idx = 0;
    int curr_kmer_idx = (int)n_kmers - 1;
    for(int event_idx = 0; event_idx < (int)n_events; ++event_idx) {
        int band_idx = (event_idx + 1) + (curr_kmer_idx + 1);
        int offset = ll_event[band_idx] - event_idx;
        if(offset >= 0 && offset < bandwidth) {
            float sc = BAND_ARRAY(band_idx, offset) + (n_events - event_idx) * lp_trim;
            if(sc > max_score) { max_score = sc; curr_event_idx = event_idx; }
        }
    }

    int curr_gap = 0, max_gap = 0;
    /* out is written backwards into a scratch list, then reversed (:363) */
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * 2 * (n_events + n_kmers + 2));
    while(curr_kmer_idx >= 0 && curr_event_idx >= 0) {
        tmp[2 * n_out] = curr_kmer_idx; tmp[2 * n_out + 1] = curr_event_idx; n_out++;
        uint32_t kmer_rank = kmer_ranks[curr_kmer_idx];
        sum_emission += npo_log_probability_match_r9(m, s, kmer_rank, event_mean[curr_event_idx], 0.0f);
        n_aligned_events += 1;

        int band_idx = (curr_event_idx + 1) + (curr_kmer_idx + 1);
        long long offset = (long long)ll_event[band_idx] - curr_event_idx;
        /* the reference reads TRACE_ARRAY(band_idx, offset) without a range check (:344-348) */
        long long flat = (long long)band_idx * ALN_BANDWIDTH + offset;
        if(flat < 0 || flat >= (long long)(n_bands * bandwidth)) { ub = 1; break; }
        uint8_t from = trace[flat];
        if(from == FROM_D) { curr_kmer_idx -= 1; curr_event_idx -= 1; curr_gap = 0; }
        else if(from == FROM_U) { curr_event_idx -= 1; curr_gap = 0; }
        else { curr_kmer_idx -= 1; curr_gap += 1; max_gap = curr_gap > max_gap ? curr_gap : max_gap; }
    }

    int result;
    if(ub) {
        result = -2;
    } else {
        /* QC, :365-372 */
        double avg_log_emission = sum_emission / n_aligned_events;
        int front_ref = tmp[2 * (n_out - 1)], back_ref = tmp[0];
        int spanned = front_ref == 0 && back_ref == (int)n_kmers - 1;
        if(avg_log_emission < min_average_log_emission || !spanned || max_gap > max_gap_threshold) {
            result = 0;
        } else {
            result = n_out;
            for(int i = 0; i < n_out && i < cap; ++i) {
                out_pairs[2 * i] = tmp[2 * (n_out - 1 - i)];
                out_pairs[2 * i + 1] = tmp[2 * (n_out - 1 - i) + 1];
            }
        }
    }
    free(tmp); free(bands); free(trace); free(ll_event); free(ll_kmer);
    return result;
#undef BAND_ARRAY
#undef TRACE_ARRAY
}

/* =====================================================================================
 * Read-level glue -- src/nanopolish_squiggle_read.cpp
 * ===================================================================================== */
/* base_to_event_map construction, squiggle_read.cpp:273-301 (strand 0) */
void npo_build_base_to_event_map(const int32_t* pairs, int n_pairs, uint32_t n_kmers,
                                 int32_t* start, int32_t* stop, double* events_per_base)
{
    for(uint32_t i = 0; i < n_kmers; ++i) { start[i] = -1; stop[i] = -1; }   /* IndexPair() */
    size_t max_event = 0;
    size_t min_event = (size_t)-1;
    size_t prev_event_idx = (size_t)-1;
    for(int i = 0; i < n_pairs; ++i) {
        size_t k_idx = pairs[2 * i];
        size_t event_idx = pairs[2 * i + 1];
        if(event_idx != prev_event_idx) {
            if(start[k_idx] == -1) start[k_idx] = (int32_t)event_idx;
            stop[k_idx] = (int32_t)event_idx;
        }
        max_event = max_event > event_idx ? max_event : event_idx;
        min_event = min_event < event_idx ? min_event : event_idx;
        prev_event_idx = event_idx;
    }
    *events_per_base = (double)(max_event - min_event) / n_kmers;
}

/* get_next_event, squiggle_read.cpp:161-171 */
static int get_next_event(const int32_t* start, int from, int stop, int stride)
{
    while(from != stop) {
        int ei = start[from];
        if(ei != -1) return ei;
        from += stride;
    }
    return -1;
}

/* get_closest_event_to, squiggle_read.cpp:174-186 */
int npo_get_closest_event_to(const int32_t* start, uint32_t n_kmers, int k_idx)
{
    int stop_before = 0 > k_idx - 1000 ? 0 : k_idx - 1000;
    int stop_after = k_idx + 1000 < (int)n_kmers - 1 ? k_idx + 1000 : (int)n_kmers - 1;
    int event_before = get_next_event(start, k_idx, stop_before, -1);
    int event_after = get_next_event(start, k_idx, stop_after, 1);
    if(event_before == -1) return event_after;
    return event_before;
}

/* =====================================================================================
 * call-methylation work-item rules
 * ===================================================================================== */
/* EventAlignmentRecord ctor, src/alignment/nanopolish_alignment_db.cpp:55-91 (strand_idx 0) */
int npo_event_alignment_record(const int32_t* aligned_bases, int n_bases, int read_length, int k, int seq_rc,
                               const int32_t* map_start, uint32_t n_kmers, int32_t* out)
{
    int n = 0;
    for(int i = 0; i < n_bases; ++i) {
        int ref_pos = aligned_bases[2 * i], read_pos = aligned_bases[2 * i + 1];
        if(read_pos < k) continue;
        if(read_pos + k >= read_length) continue;
        int kmer_pos_ref_strand = read_pos;
        int kmer_pos_read_strand = seq_rc ? read_length - kmer_pos_ref_strand - k : kmer_pos_ref_strand; /* flip_k_strand */
        int event_idx = npo_get_closest_event_to(map_start, n_kmers, kmer_pos_read_strand);
        out[2 * n] = ref_pos; out[2 * n + 1] = event_idx; n++;
    }
    if(n > 0 && out[1] == out[2 * (n - 1) + 1]) n = 0;   /* degenerate alignment, :85-87 */
    return n;
}

/* std::lower_bound with AlignedPairRefLBComp (src/alignment/nanopolish_anchor.h:24-27) */
static int lower_bound_ref(const int32_t* pairs, int n, int v)
{
    int lo = 0, hi = n;
    while(lo < hi) { int mid = lo + (hi - lo) / 2; if(pairs[2 * mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

/* get_aligned_segments, src/alignment/nanopolish_anchor.cpp:20-95 (read_stride 1), first segment only: a BAM_CREF_SKIP
 * starts a second segment, which SequenceAlignmentRecord rejects (alignment_db.cpp:43-47) -> -1 here. */
int npo_cigar_aligned_bases(const uint32_t* cigar, int n_cigar, int pos, int32_t* out_pairs, int cap)
{
    int read_pos = 0, ref_pos = pos, n = 0;
    for(int ci = 0; ci < n_cigar; ++ci) {
        int cigar_len = cigar[ci] >> 4;
        int cigar_op = cigar[ci] & 0xf;
        int read_inc = 0, ref_inc = 0, is_aligned = 0;
        if(cigar_op == 0 || cigar_op == 7 || cigar_op == 8) { is_aligned = 1; read_inc = 1; ref_inc = 1; }   /* M, =, X */
        else if(cigar_op == 2) { ref_inc = 1; }                                                             /* D */
        else if(cigar_op == 3) { return -1; }                                                               /* N */
        else if(cigar_op == 1) { read_inc = 1; }                                                            /* I */
        else if(cigar_op == 4) { read_inc = 1; }                                                            /* S */
        else if(cigar_op == 5) { read_inc = 0; }                                                            /* H */
        else { return -1; }                                        /* assert(false && "Unhandled cigar operation") */
        for(int j = 0; j < cigar_len; ++j) {
            if(is_aligned) {
                if(n < cap) { out_pairs[2 * n] = ref_pos; out_pairs[2 * n + 1] = read_pos; }
                n++;
            }
            read_pos += read_inc;
            ref_pos += ref_inc;
        }
    }
    return n;
}

/* _find_iter_by_ref_bounds + _find_by_ref_bounds, src/alignment/nanopolish_alignment_db.cpp:688-731 */
int npo_find_by_ref_bounds(const int32_t* pairs, int n, int ref_start, int ref_stop, int* read_start, int* read_stop)
{
    int si = lower_bound_ref(pairs, n, ref_start);
    int ti = lower_bound_ref(pairs, n, ref_stop);
    if(si == n || ti == n) return 0;
    int left_bounded = pairs[2 * si] <= ref_start || (si != 0 && pairs[2 * (si - 1)] <= ref_start);
    /* the reference's second clause dereferences stop_iter+1; it is only evaluated when the first
     * clause is false, which lower_bound makes impossible for a non-end iterator */
    int right_bounded = pairs[2 * ti] >= ref_stop;
    if(left_bounded && right_bounded) { *read_start = pairs[2 * si + 1]; *read_stop = pairs[2 * ti + 1]; return 1; }
    return 0;
}

/* motif scan + grouping, src/basemods/nanopolish_basemods.cpp:298-320 */
int npo_scan_motif_groups(int a, const char* ref_seq, int n, int min_separation,
                          int32_t* first_site, int32_t* last_site, int32_t* n_motif, int cap)
{
    int* sites = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    int ns = 0;
    for(int i = 0; i + 1 < n; ++i) if(npo_is_motif_match(a, ref_seq, n, i)) sites[ns++] = i;
    int ng = 0, curr = 0;
    while(curr < ns) {
        int end = curr + 1;
        while(end < ns) { if(sites[end] - sites[end - 1] > min_separation) break; end += 1; }
        if(ng < cap) { first_site[ng] = sites[curr]; last_site[ng] = sites[end - 1]; n_motif[ng] = end - curr; }
        ng++;
        curr = end;
    }
    free(sites);
    return ng;
}

/* =====================================================================================
 * Bounded CPU drivers for bench.py's cpu_baseline (kind="port").  OpenMP over reads/jobs mirrors
 * `#pragma omp parallel for schedule(dynamic)` of src/common/nanopolish_bam_processor.cpp:99.
 * ===================================================================================== */
void npo_align_many(const npo_model* m, int n_reads, const float* event_mean, const int64_t* event_off,
                    const uint32_t* ranks, const int64_t* rank_off, const double* shift, const double* scale,
                    int32_t* out_pairs, const int64_t* pair_off, int32_t* out_n, int n_threads)
{
    npo_flogsum_table();
    #pragma omp parallel for schedule(dynamic) num_threads(n_threads)
    for(int r = 0; r < n_reads; ++r) {
        npo_scalings s = npo_set4(shift[r], scale[r], 0.0, 1.0);
        out_n[r] = npo_adaptive_banded_simple_event_align(m, &s, event_mean + event_off[r],
                       (uint32_t)(event_off[r + 1] - event_off[r]), ranks + rank_off[r],
                       (uint32_t)(rank_off[r + 1] - rank_off[r]), out_pairs + 2 * pair_off[r],
                       (int)(pair_off[r + 1] - pair_off[r]));
    }
}

void npo_score_many(const npo_model* m, int64_t n_jobs, const int32_t* job_read,
                    const float* event_mean, const int64_t* event_off,
                    const double* shift, const double* scale, const double* var, const double* events_per_base,
                    const uint32_t* ranks, const int64_t* job_rank_off,
                    const uint32_t* e_start, const uint32_t* e_stop, const int8_t* stride,
                    double indel_bias, uint32_t flags, float* out, int n_threads)
{
    npo_flogsum_table();
    #pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads)
    for(int64_t j = 0; j < n_jobs; ++j) {
        int r = job_read[j];
        npo_scalings s = npo_set4(shift[r], scale[r], 0.0, var[r]);
        out[j] = npo_profile_hmm_score(m, &s, event_mean + event_off[r], ranks + job_rank_off[j],
                                       (uint32_t)(job_rank_off[j + 1] - job_rank_off[j]),
                                       e_start[j], e_stop[j], stride[j], events_per_base[r], indel_bias, flags);
    }
}

/* =====================================================================================
 * f1: calibration between the aligner and the HMM
 *   get_eventalignment_for_1d_basecalls  src/nanopolish_squiggle_read.cpp:339-389 (shift_offset 0)
 *   recalibrate_model(..., scale_var = true, scale_drift = false)  src/nanopolish_methyltrain.cpp:204-306
 * The 2x2 solve is Eigen 3.3.7 `A.fullPivLu().solve(b)` (methyltrain.cpp:283).  Eigen is NOT in this container, so
 * the restatement below follows Eigen's published algorithm (FullPivLU::computeInPlace + _solve_impl: complete
 * pivoting with the first maximum in column-major order, in-place elimination, rank from the default threshold
 * epsilon * diagonalSize, unit-lower then upper triangular solves, inverse column permutation).  Everything around the
 * solve is pinned against the reference's own recalibrate_model compiled in place (tests/test_oracle_vs_ref_full.py);
 * the solve itself is "parity unpinned" (the same restatement stands in for Eigen there).
 * ===================================================================================== */
/* Which Eigen?  The reference's Makefile pins 3.3.7 (Makefile:59), its README still names 3.2.5.  The two differ on this path in
 * ONE operation: `col(k).tail() /= pivot` is a true division in 3.3 and a multiplication by the rounded reciprocal in 3.2
 * (DenseBase::operator/=(Scalar) multiplied by Scalar(1)/other for floating types until 3.3).  Default (0) = 3.3.7, what the
 * device kernel implements; 1 = the 3.2 form, only so that tests/test_reflevel_golden.py can measure what the choice moves. */
static int g_eigen32_scalar_div = 0;
void npo_set_eigen32_scalar_div(int on) { g_eigen32_scalar_div = on; }

static void eigen_fullpivlu_solve_2x2(const double Ain[4] /* row-major a00 a01 a10 a11 */, const double bin[2], double x[2])
{
    double m[2][2] = {{Ain[0], Ain[1]}, {Ain[2], Ain[3]}};
    /* k = 0: biggest |coeff| of the whole matrix, first maximum in column-major visiting order */
    int pr = 0, pc = 0; double big = fabs(m[0][0]);
    if(fabs(m[1][0]) > big) { big = fabs(m[1][0]); pr = 1; pc = 0; }
    if(fabs(m[0][1]) > big) { big = fabs(m[0][1]); pr = 0; pc = 1; }
    if(fabs(m[1][1]) > big) { big = fabs(m[1][1]); pr = 1; pc = 1; }
    x[0] = x[1] = 0.0;
    if(big == 0.0) return;                                   /* zero matrix: rank 0, solution 0 */
    if(pr == 1) { double t; t = m[0][0]; m[0][0] = m[1][0]; m[1][0] = t; t = m[0][1]; m[0][1] = m[1][1]; m[1][1] = t; }
    if(pc == 1) { double t; t = m[0][0]; m[0][0] = m[0][1]; m[0][1] = t; t = m[1][0]; m[1][0] = m[1][1]; m[1][1] = t; }
    if(g_eigen32_scalar_div) m[1][0] *= (1.0 / m[0][0]); else m[1][0] /= m[0][0];
    m[1][1] -= m[1][0] * m[0][1];
    const double maxpivot = fabs(m[0][0]) > fabs(m[1][1]) ? fabs(m[0][0]) : fabs(m[1][1]);   /* m_maxpivot */
    const double thr = maxpivot * (2.220446049250313e-16 * 2);                                 /* epsilon * diagonalSize */
    const int rank = (fabs(m[0][0]) > thr) + (fabs(m[1][1]) > thr);
    double c0 = pr == 1 ? bin[1] : bin[0], c1 = pr == 1 ? bin[0] : bin[1];                     /* P b */
    c1 -= m[1][0] * c0;                                                                       /* unit-lower solve */
    double y0, y1 = 0.0;
    if(rank == 2) { y1 = c1 / m[1][1]; c0 -= y1 * m[0][1]; }                                   /* upper solve, top-left rank block */
    y0 = c0 / m[0][0];
    if(pc == 1) { x[0] = y1; x[1] = y0; } else { x[0] = y0; x[1] = y1; }                       /* Q y */
}

/* returns 1 if recalibrated (>= 200 'M' events), else 0 and leaves the outputs untouched */
int npo_recalibrate(const npo_model* m, const float* event_mean, const uint32_t* kmer_ranks, uint32_t n_kmers,
                    const int32_t* map_start, const int32_t* map_stop, double* shift_out, double* scale_out, double* var_out)
{
    /* get_eventalignment_for_1d_basecalls + the 'M' filter of recalibrate_model (methyltrain.cpp:221-240) */
    size_t cap = 0;
    for(uint32_t ki = 0; ki < n_kmers; ++ki) if(map_start[ki] != -1) cap += (size_t)(map_stop[ki] - map_start[ki] + 1);
    double* raw_events = (double*)malloc(sizeof(double) * (cap + 1));
    double* level_means = (double*)malloc(sizeof(double) * (cap + 1));
    double* level_stdvs = (double*)malloc(sizeof(double) * (cap + 1));
    size_t n = 0;
    size_t prev_kmer_rank = (size_t)-1;
    for(uint32_t ki = 0; ki < n_kmers; ++ki) {
        if(map_start[ki] == -1) continue;
        for(int32_t event_idx = map_start[ki]; event_idx <= map_stop[ki]; event_idx++) {
            size_t kmer_rank = kmer_ranks[ki];
            if(prev_kmer_rank != kmer_rank) {                  /* hmm_state 'M' */
                raw_events[n] = event_mean[event_idx];
                level_means[n] = m->level_mean[kmer_rank];
                level_stdvs[n] = m->level_stdv[kmer_rank];
                n++;
            }
            prev_kmer_rank = kmer_rank;
        }
    }
    int recalibrated = 0;
    if(n >= 200) {                                             /* minNumEventsToRescale */
        double A[4] = {0., 0., 0., 0.}, b[2] = {0., 0.};
        for(size_t i = 0; i < n; i++) {
            double inv_var = 1. / (level_stdvs[i] * level_stdvs[i]);
            double mu = level_means[i];
            double e = raw_events[i];
            A[0] += inv_var; A[1] += mu * inv_var;
            A[3] += mu * mu * inv_var;
            b[0] += e * inv_var;
            b[1] += mu * e * inv_var;
        }
        A[2] = A[1];
        double x[2];
        eigen_fullpivlu_solve_2x2(A, b, x);
        double shift = x[0], scale = x[1];
        double var = 0.;
        for(size_t i = 0; i < n; i++) {
            double yi = (raw_events[i] - shift - scale * level_means[i]);
            var += yi * yi / (level_stdvs[i] * level_stdvs[i]);
        }
        var /= n;
        var = sqrt(var);
        *shift_out = shift; *scale_out = scale; *var_out = var;
        recalibrated = 1;
    }
    free(raw_events); free(level_means); free(level_stdvs);
    return recalibrated;
}

/* =====================================================================================
 * f2: scrappie event detection -- src/thirdparty/scrappie/event_detection.c (vendored in the reference),
 * called on the whole raw table by SquiggleRead::load_from_raw (src/nanopolish_squiggle_read.cpp:229-236;
 * trim_and_segment_raw's result is discarded there, so no trimming takes effect).
 * Pinned against the reference's own object code (oracle/_ref) by tests/test_oracle_vs_ref.py.
 * ===================================================================================== */
typedef struct {
    int def_peak_pos; float def_peak_val;
    const float* signal; size_t signal_length; float threshold; size_t window_length;
    size_t masked_to; int peak_pos; float peak_value; int valid_peak;
} ed_detector;                                          /* Detector, event_detection.c:10-21 */

/* compute_tstat, event_detection.c:63-119 */
static void ed_compute_tstat(const double* sum, const double* sumsq, size_t d_length, size_t w_length, float* tstat)
{
    const float eta = 1.17549435e-38f;                  /* FLT_MIN */
    const float w_lengthf = (float)w_length;
    for(size_t i = 0; i < d_length; ++i) tstat[i] = 0.0f;                       /* calloc + quick return / boundaries */
    if(d_length < 2 * w_length || w_length < 2) return;
    for(size_t i = w_length; i <= d_length - w_length; ++i) {
        double sum1 = sum[i];
        double sumsq1 = sumsq[i];
        if(i > w_length) { sum1 -= sum[i - w_length]; sumsq1 -= sumsq[i - w_length]; }
        float sum2 = (float)(sum[i + w_length] - sum[i]);
        float sumsq2 = (float)(sumsq[i + w_length] - sumsq[i]);
        float mean1 = sum1 / w_lengthf;
        float mean2 = sum2 / w_lengthf;
        float combined_var = sumsq1 / w_lengthf - mean1 * mean1 + sumsq2 / w_lengthf - mean2 * mean2;
        combined_var = fmaxf(combined_var, eta);
        const float delta_mean = mean2 - mean1;
        tstat[i] = fabs(delta_mean) / sqrt(combined_var / w_lengthf);
    }
}

/* short_long_peak_detector, event_detection.c:126-207; returns the number of peaks written */
static size_t ed_peaks(ed_detector* sd, ed_detector* ld, float peak_height, size_t* peaks)
{
    ed_detector* detectors[2] = { sd, ld };
    size_t peak_count = 0;
    for(size_t i = 0; i < sd->signal_length; i++) {
        for(int k = 0; k < 2; k++) {
            ed_detector* d = detectors[k];
            if(d->masked_to >= i) continue;
            float current_value = d->signal[i];
            if(d->peak_pos == d->def_peak_pos) {
                if(current_value < d->peak_value) {
                    d->peak_value = current_value;
                } else if(current_value - d->peak_value > peak_height) {
                    d->peak_value = current_value;
                    d->peak_pos = (int)i;
                }
            } else {
                if(current_value > d->peak_value) { d->peak_value = current_value; d->peak_pos = (int)i; }
                if(d == sd) {
                    if(d->peak_value > d->threshold) {
                        ld->masked_to = d->peak_pos + d->window_length;
                        ld->peak_pos = ld->def_peak_pos;
                        ld->peak_value = ld->def_peak_val;
                        ld->valid_peak = 0;
                    }
                }
                if(d->peak_value - current_value > peak_height && d->peak_value > d->threshold) d->valid_peak = 1;
                if(d->valid_peak && (i - d->peak_pos) > d->window_length / 2) {
                    peaks[peak_count++] = (size_t)d->peak_pos;
                    d->peak_pos = d->def_peak_pos;
                    d->peak_value = current_value;
                    d->valid_peak = 0;
                }
            }
        }
    }
    return peak_count;
}

/* detect_events, event_detection.c:268-319 (compute_sum_sumsq :35-50, create_event :223-241, create_events :243-266).
 * Returns the number of events (1 + #peaks); 0 when there is no peak at all -- the reference then reads peaks[-1]
 * (undefined behaviour), which no read with real signal reaches. */
int npo_detect_events(const float* raw, size_t n, size_t w1, size_t w2, float t1, float t2, float peak_height,
                      uint64_t* out_start, float* out_length, float* out_mean, float* out_stdv, size_t cap)
{
    double* sums = (double*)calloc(n + 1, sizeof(double));
    double* sumsqs = (double*)calloc(n + 1, sizeof(double));
    float* ts1 = (float*)malloc(sizeof(float) * (n ? n : 1));
    float* ts2 = (float*)malloc(sizeof(float) * (n ? n : 1));
    size_t* peaks = (size_t*)calloc(n ? n : 1, sizeof(size_t));
    sums[0] = 0.0f; sumsqs[0] = 0.0f;
    for(size_t i = 0; i < n; ++i) {
        sums[i + 1] = sums[i] + raw[i];
        sumsqs[i + 1] = sumsqs[i] + raw[i] * raw[i];          /* float product, double accumulation (:47) */
    }
    ed_compute_tstat(sums, sumsqs, n, w1, ts1);
    ed_compute_tstat(sums, sumsqs, n, w2, ts2);
    ed_detector sd = { -1, 3.40282347e+38f, ts1, n, t1, w1, 0, -1, 3.40282347e+38f, 0 };
    ed_detector ld = { -1, 3.40282347e+38f, ts2, n, t2, w2, 0, -1, 3.40282347e+38f, 0 };
    const size_t n_peaks = ed_peaks(&sd, &ld, peak_height, peaks);
    int n_ev = 0;
    if(n_peaks > 0) {
        n_ev = (int)n_peaks + 1;
        for(size_t ev = 0; ev < (size_t)n_ev && ev < cap; ++ev) {
            const size_t start = ev == 0 ? 0 : peaks[ev - 1];
            const size_t end = ev == (size_t)n_ev - 1 ? n : peaks[ev];
            const float length = (float)(end - start);
            const float mean = (float)(sums[end] - sums[start]) / length;
            const float deltasqr = (sumsqs[end] - sumsqs[start]);
            const float var = deltasqr / length - mean * mean;
            out_start[ev] = (uint64_t)start; out_length[ev] = length; out_mean[ev] = mean; out_stdv[ev] = sqrtf(fmaxf(var, 0.0f));
        }
    }
    free(sums); free(sumsqs); free(ts1); free(ts2); free(peaks);
    return n_ev;
}

/* bounded driver for bench.py's cpu_baseline (kind = "port") */
void npo_detect_events_many(int n_reads, const float* raw, const int64_t* raw_off, float* out_mean, const int64_t* ev_off,
                            int32_t* out_n, int n_threads)
{
#pragma omp parallel for schedule(dynamic) num_threads(n_threads)
    for(int r = 0; r < n_reads; ++r) {
        const size_t n = (size_t)(raw_off[r + 1] - raw_off[r]);
        const size_t cap = (size_t)(ev_off[r + 1] - ev_off[r]);
        uint64_t* st = (uint64_t*)malloc(sizeof(uint64_t) * (cap ? cap : 1));
        float* ln = (float*)malloc(sizeof(float) * (cap ? cap : 1));
        float* sv = (float*)malloc(sizeof(float) * (cap ? cap : 1));
        out_n[r] = npo_detect_events(raw + raw_off[r], n, 3, 6, 1.4f, 9.0f, 0.2f, st, ln, out_mean + ev_off[r], sv, cap);
        free(st); free(ln); free(sv);
    }
}

/* the aligner's per-read constants on their own (raw_loader.cpp:99-108), for checking the device's restated log/exp */
void npo_aligner_constants(uint32_t n_events, uint32_t n_kmers, double out[4])
{
    double events_per_kmer = (double)n_events / n_kmers;
    double p_stay = 1 - (1 / (events_per_kmer + 1));
    double epsilon = 1e-10;
    out[0] = log(epsilon);
    out[1] = log(p_stay);
    out[2] = log(1.0 - exp(out[0]) - exp(out[1]));
    out[3] = log(0.01);
}
