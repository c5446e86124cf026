// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A C-ABI veneer over the *unmodified* nanopolish reference sources, which are
// compiled in place from /root/reference by oracle/Makefile into
// oracle/_ref/libnp_ref.so.  Nothing from the reference is copied here: this
// file only (1) supplies the three out-of-line members the hot path needs from
// src/nanopolish_squiggle_read.cpp (a TU that cannot be built here because it
// drags in HDF5/slow5/Eigen I/O), and (2) flattens the C++ entry points
//   profile_hmm_score / profile_hmm_score_set / profile_hmm_align
//       (src/hmm/nanopolish_profile_hmm.h:24-31)
//   adaptive_banded_simple_event_align, estimate_scalings_using_mom
//       (src/nanopolish_raw_loader.h:16-24)
// into plain-pointer functions that python/ctypes can drive.  It is used to
//   * pin oracle/np_oracle.c (the portable C restatement) against the real code,
//   * generate tests/golden/*.npz (tests/gen_golden.py),
//   * optionally serve as bench.py's cpu_baseline (kind = "reference").
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
#include <cstring>
#include <string>
#include <vector>
#include <cmath>
#include <omp.h>
#include <malloc.h>
#include "nanopolish_profile_hmm.h"
#include "nanopolish_raw_loader.h"
#include "nanopolish_pore_model_set.h"
#include "nanopolish_alphabet.h"
#include "logsum.h"

extern double hmm_indel_bias_factor; // src/hmm/nanopolish_profile_hmm_r9.cpp:19

// --- the three members that live in src/nanopolish_squiggle_read.cpp --------
// :155-158 (empty destructor)
SquiggleRead::~SquiggleRead() {}
// :38-44
void SquiggleScalings::set4(double _shift, double _scale, double _drift, double _var)
{
    set6(_shift, _scale, _drift, _var, 1.0, 1.0);
}
// :46-65
void SquiggleScalings::set6(double _shift, double _scale, double _drift, double _var,
                            double _scale_sd, double _var_sd)
{
    shift = _shift; scale = _scale; drift = _drift; var = _var;
    scale_sd = _scale_sd; var_sd = _var_sd;
    log_var = log(var);
    scaled_var = var / scale;
    log_scaled_var = log(scaled_var);
}

namespace {

const Alphabet* alphabet_by_name(const char* name)
{
    std::string n(name);
    if(n == "nucleotide") return &gDNAAlphabet;
    if(n == "cpg") return &gMCpGAlphabet;
    if(n == "gpc") return &gMethylGpCAlphabet;
    if(n == "dam") return &gMethylDamAlphabet;
    if(n == "dcm") return &gMethylDcmAlphabet;
    if(n == "u_to_t_rna") return &gUtoTRNAAlphabet;
    return NULL;
}

// Populate a SquiggleRead by hand, following the reference's own "scalings"
// unit test (src/test/nanopolish_test.cpp:279-312).
struct ReadHolder
{
    SquiggleRead sr;
    ReadHolder(const char* kit, const float* event_mean, int n_events,
               const char* read_sequence, double shift, double scale, double drift,
               double var, double events_per_base)
    {
        sr.pore_type = PORETYPE_R9;
        sr.read_type = SRT_TEMPLATE;
        sr.nucleotide_type = SRNT_DNA;
        sr.base_model[0] = PoreModelSet::get_model(kit, "nucleotide", "template", 6);
        sr.base_model[1] = NULL;
        sr.scalings[0].set4(shift, scale, drift, var);
        sr.events[0].resize(n_events);
        for(int i = 0; i < n_events; ++i) {
            SquiggleEvent& e = sr.events[0][i];
            e.mean = event_mean[i];
            e.stdv = 1.0f;
            e.start_time = 0.002 * i;
            e.duration = 0.002f;
            e.log_stdv = 0.0f;
        }
        sr.events_per_base[0] = events_per_base;
        sr.events_per_base[1] = 0.0;
        if(read_sequence) sr.read_sequence = read_sequence;
    }
};

} // namespace

extern "C" {

// ---- tables -----------------------------------------------------------------
void npref_flogsum_table(float* out)
{
    extern float flogsum_lookup[p7_LOGSUM_TBL];
    memcpy(out, flogsum_lookup, sizeof(float) * p7_LOGSUM_TBL);
}

float npref_add_logs(float a, float b) { return (float)add_logs(a, b); }

int npref_model_size(const char* kit, const char* alphabet, int k)
{
    const PoreModel* m = PoreModelSet::get_model(kit, alphabet, "template", k);
    return m ? (int)m->states.size() : -1;
}

int npref_model_get(const char* kit, const char* alphabet, int k,
                    double* level_mean, double* level_stdv, double* level_log_stdv)
{
    const PoreModel* m = PoreModelSet::get_model(kit, alphabet, "template", k);
    if(!m) return -1;
    for(size_t i = 0; i < m->states.size(); ++i) {
        level_mean[i] = m->states[i].level_mean;
        level_stdv[i] = m->states[i].level_stdv;
        level_log_stdv[i] = m->states[i].level_log_stdv;
    }
    return (int)m->states.size();
}

// ---- alphabets ---------------------------------------------------------------
void npref_set_indel_bias(double v) { hmm_indel_bias_factor = v; }

// Test hook: change a registered model IN PLACE (same PoreModel address), which is what PoreModelSet::register_model does for
// an existing key (src/pore_model/nanopolish_pore_model_set.cpp:70; methyltrain's add_model every training round).
void npref_shift_model(const char* kit, const char* alphabet, int k, double delta)
{
    PoreModel* m = const_cast<PoreModel*>(PoreModelSet::get_model(kit, alphabet, "template", k));
    for(size_t i = 0; i < m->states.size(); ++i) m->states[i].level_mean += delta;
}

int npref_kmer_rank(const char* alphabet, const char* kmer, int k)
{
    const Alphabet* a = alphabet_by_name(alphabet);
    return a ? (int)a->kmer_rank(kmer, k) : -1;
}
static int copy_out(const std::string& s, char* out) { memcpy(out, s.c_str(), s.size() + 1); return (int)s.size(); }
int npref_reverse_complement(const char* alphabet, const char* in, char* out)
{ return copy_out(alphabet_by_name(alphabet)->reverse_complement(in), out); }
int npref_methylate(const char* alphabet, const char* in, char* out)
{ return copy_out(alphabet_by_name(alphabet)->methylate(in), out); }
int npref_unmethylate(const char* alphabet, const char* in, char* out)
{ return copy_out(alphabet_by_name(alphabet)->unmethylate(in), out); }
int npref_disambiguate(const char* alphabet, const char* in, char* out)
{ return copy_out(alphabet_by_name(alphabet)->disambiguate(in), out); }
int npref_is_motif_match(const char* alphabet, const char* str, int i)
{ return alphabet_by_name(alphabet)->is_motif_match(str, i) ? 1 : 0; }

// ---- emission (src/hmm/nanopolish_emissions.h:57-68) --------------------------
float npref_log_probability_match_r9(const char* kit, const char* alphabet, int rank, float event_mean,
                                     double shift, double scale, double drift, double var)
{
    ReadHolder h(kit, &event_mean, 1, NULL, shift, scale, drift, var, 1.5);
    const PoreModel* m = PoreModelSet::get_model(kit, alphabet, "template", 6);
    return log_probability_match_r9(h.sr, *m, rank, 0, 0);
}

// ---- MoM scaling estimate (src/nanopolish_raw_loader.cpp:17-60) ---------------
void npref_estimate_scalings_mom(const char* kit, const char* sequence, const float* event_mean, int n_events,
                                 double* shift, double* scale)
{
    const PoreModel* m = PoreModelSet::get_model(kit, "nucleotide", "template", 6);
    std::vector<event_t> ev(n_events);
    for(int i = 0; i < n_events; ++i) { memset(&ev[i], 0, sizeof(event_t)); ev[i].mean = event_mean[i]; }
    event_table et; et.n = n_events; et.start = 0; et.end = n_events; et.event = ev.data();
    SquiggleScalings s = estimate_scalings_using_mom(sequence, *m, et);
    *shift = s.shift; *scale = s.scale;
}

// ---- adaptive banded event alignment (src/nanopolish_raw_loader.cpp:77-379) ----
// out_pairs: interleaved (ref_pos, read_pos). Returns #pairs (0 == QC failure / empty vector).
int npref_event_align(const char* kit, const float* event_mean, int n_events, const char* sequence,
                      double shift, double scale, double drift, double var,
                      int32_t* out_pairs, int cap)
{
    ReadHolder h(kit, event_mean, n_events, sequence, shift, scale, drift, var, 0.0);
    std::vector<AlignedPair> r = adaptive_banded_simple_event_align(h.sr, *h.sr.base_model[0], sequence);
    int n = (int)r.size();
    for(int i = 0; i < n && i < cap; ++i) { out_pairs[2*i] = r[i].ref_pos; out_pairs[2*i+1] = r[i].read_pos; }
    return n;
}

// ---- profile HMM ----------------------------------------------------------------
static HMMInputData make_data(ReadHolder& h, const char* kit, const char* alphabet,
                              uint32_t e_start, uint32_t e_stop, int stride, int rc)
{
    HMMInputData d;
    d.read = &h.sr;
    d.pore_model = PoreModelSet::get_model(kit, alphabet, "template", 6);
    d.event_start_idx = e_start; d.event_stop_idx = e_stop;
    d.strand = 0; d.event_stride = (int8_t)stride; d.rc = (uint8_t)rc;
    return d;
}

// profile_hmm_score (src/hmm/nanopolish_profile_hmm.cpp:23-30)
float npref_hmm_score(const char* kit, const char* alphabet, const char* seq, const char* rc_seq,
                      const float* event_mean, int n_events_total,
                      uint32_t e_start, uint32_t e_stop, int stride, int rc,
                      double shift, double scale, double var, double events_per_base,
                      double indel_bias, uint32_t flags)
{
    ReadHolder h(kit, event_mean, n_events_total, NULL, shift, scale, 0.0, var, events_per_base);
    HMMInputData d = make_data(h, kit, alphabet, e_start, e_stop, stride, rc);
    const Alphabet* a = alphabet_by_name(alphabet);
    hmm_indel_bias_factor = indel_bias;
    float s;
    if(rc_seq) { HMMInputSequence hs(seq, rc_seq, a); s = profile_hmm_score(hs, d, flags); }
    else       { HMMInputSequence hs(seq, a);         s = profile_hmm_score(hs, d, flags); }
    hmm_indel_bias_factor = 1.0;
    return s;
}

// Batched form sharing one read: scores n_jobs windows of the same read (OpenMP over jobs off).
void npref_hmm_score_many(const char* kit, const char* alphabet, int n_jobs,
                          const char* const* seqs, const char* const* rc_seqs,
                          const float* event_mean, int n_events_total,
                          const uint32_t* e_start, const uint32_t* e_stop, const int* stride, const int* rc,
                          double shift, double scale, double var, double events_per_base,
                          double indel_bias, uint32_t flags, float* out)
{
    ReadHolder h(kit, event_mean, n_events_total, NULL, shift, scale, 0.0, var, events_per_base);
    const Alphabet* a = alphabet_by_name(alphabet);
    hmm_indel_bias_factor = indel_bias;
    for(int j = 0; j < n_jobs; ++j) {
        HMMInputData d = make_data(h, kit, alphabet, e_start[j], e_stop[j], stride[j], rc[j]);
        HMMInputSequence hs(seqs[j], rc_seqs[j], a);
        out[j] = profile_hmm_score(hs, d, flags);
    }
    hmm_indel_bias_factor = 1.0;
}

// The vector overload (src/hmm/nanopolish_profile_hmm.cpp:14-21): ONE sequence against several reads' event windows,
// the fp32 sum of the per-read scores in index order.  Read i: events event_mean[event_off[i] .. event_off[i+1]).
float npref_hmm_score_vec(const char* kit, const char* alphabet, const char* seq, int n_data,
                          const float* event_mean, const int64_t* event_off,
                          const uint32_t* e_start, const uint32_t* e_stop, const int* stride, const int* rc,
                          const double* shift, const double* scale, const double* var, const double* events_per_base,
                          double indel_bias, uint32_t flags)
{
    std::vector<ReadHolder*> hs;
    std::vector<HMMInputData> data;
    for(int i = 0; i < n_data; ++i) {
        hs.push_back(new ReadHolder(kit, event_mean + event_off[i], (int)(event_off[i+1] - event_off[i]), NULL,
                                    shift[i], scale[i], 0.0, var[i], events_per_base[i]));
        data.push_back(make_data(*hs.back(), kit, alphabet, e_start[i], e_stop[i], stride[i], rc[i]));
    }
    HMMInputSequence sq(seq, alphabet_by_name(alphabet));
    hmm_indel_bias_factor = indel_bias;
    float s = profile_hmm_score(sq, data, flags);
    hmm_indel_bias_factor = 1.0;
    for(size_t i = 0; i < hs.size(); ++i) delete hs[i];
    return s;
}

// profile_hmm_score_set (src/hmm/nanopolish_profile_hmm.cpp:32-56): sequences[0] nucleotide, others by alphabet name
float npref_hmm_score_set(const char* kit, int n_seqs, const char* const* seqs, const char* const* alphabets,
                          const float* event_mean, int n_events_total,
                          uint32_t e_start, uint32_t e_stop, int stride, int rc,
                          double shift, double scale, double var, double events_per_base,
                          double indel_bias, uint32_t flags)
{
    ReadHolder h(kit, event_mean, n_events_total, NULL, shift, scale, 0.0, var, events_per_base);
    HMMInputData d = make_data(h, kit, "nucleotide", e_start, e_stop, stride, rc);
    std::vector<HMMInputSequence> v;
    for(int i = 0; i < n_seqs; ++i) v.push_back(HMMInputSequence(seqs[i], alphabet_by_name(alphabets[i])));
    hmm_indel_bias_factor = indel_bias;
    float s = profile_hmm_score_set(v, d, flags);
    hmm_indel_bias_factor = 1.0;
    return s;
}

// profile_hmm_align (src/hmm/nanopolish_profile_hmm.cpp:58-65)
int npref_hmm_align(const char* kit, const char* alphabet, const char* seq, const char* rc_seq,
                    const float* event_mean, int n_events_total,
                    uint32_t e_start, uint32_t e_stop, int stride, int rc,
                    double shift, double scale, double var, double events_per_base,
                    double indel_bias, uint32_t flags,
                    uint32_t* out_event_idx, uint32_t* out_kmer_idx, double* out_l_fm, char* out_state, int cap)
{
    ReadHolder h(kit, event_mean, n_events_total, NULL, shift, scale, 0.0, var, events_per_base);
    HMMInputData d = make_data(h, kit, alphabet, e_start, e_stop, stride, rc);
    const Alphabet* a = alphabet_by_name(alphabet);
    hmm_indel_bias_factor = indel_bias;
    std::vector<HMMAlignmentState> r;
    if(rc_seq) { HMMInputSequence hs(seq, rc_seq, a); r = profile_hmm_align(hs, d, flags); }
    else       { HMMInputSequence hs(seq, a);         r = profile_hmm_align(hs, d, flags); }
    hmm_indel_bias_factor = 1.0;
    int n = (int)r.size();
    for(int i = 0; i < n && i < cap; ++i) {
        out_event_idx[i] = r[i].event_idx; out_kmer_idx[i] = r[i].kmer_idx;
        out_l_fm[i] = r[i].l_fm; out_state[i] = r[i].state;
    }
    return n;
}

// ---- allocator settings for the CPU baseline (process-wide, harness only; the reference code is untouched) ----
// adaptive_banded_simple_event_align mallocs and frees ~6.5 MB of band + trace arrays per read
// (src/nanopolish_raw_loader.cpp:123-138).  With glibc's defaults every one of them is an mmap/munmap pair plus a page
// fault per 4 KB, and on a many-core host the OpenMP threads serialise on the process's address-space lock: 256 threads
// then align 1.5 reads/s/core instead of ~30.  Keeping those blocks in the per-thread arenas (no mmap for them, no
// trimming on free) removes the allocator from the measurement.
void npref_tune_malloc(void)
{
    mallopt(M_MMAP_THRESHOLD, 32 << 20);       // DEFAULT_MMAP_THRESHOLD_MAX on 64-bit
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 16 << 20);
}

// ---- CPU baseline driver: align + (2 x score per job) over many reads, OpenMP over reads ----
// Mirrors `#pragma omp parallel for schedule(dynamic)` of src/common/nanopolish_bam_processor.cpp:99.
// Reads are given SoA/CSR style. Jobs (CpG groups) are given per read with window sequences.
// Only the two hot-path calls are timed by the caller (this function does no I/O).
void npref_align_many(const char* kit, int n_reads, const float* event_mean, const int64_t* event_off,
                      const char* const* sequences, const double* shift, const double* scale,
                      int32_t* out_pairs, const int64_t* pair_off, int32_t* out_n, int n_threads)
{
    #pragma omp parallel for schedule(dynamic) num_threads(n_threads)
    for(int r = 0; r < n_reads; ++r) {
        int ne = (int)(event_off[r+1] - event_off[r]);
        int cap = (int)(pair_off[r+1] - pair_off[r]);
        out_n[r] = npref_event_align(kit, event_mean + event_off[r], ne, sequences[r], shift[r], scale[r], 0.0, 1.0,
                                     out_pairs + 2 * pair_off[r], cap);
    }
}

void npref_score_many_reads(const char* kit, const char* alphabet, int n_reads,
                            const float* event_mean, const int64_t* event_off,
                            const double* shift, const double* scale, const double* var, const double* events_per_base,
                            const int64_t* job_off, const char* const* seqs, const char* const* rc_seqs,
                            const uint32_t* e_start, const uint32_t* e_stop, const int* stride, const int* rc,
                            uint32_t flags, float* out, int n_threads)
{
    #pragma omp parallel for schedule(dynamic) num_threads(n_threads)
    for(int r = 0; r < n_reads; ++r) {
        int64_t j0 = job_off[r], j1 = job_off[r+1];
        if(j1 == j0) continue;
        int ne = (int)(event_off[r+1] - event_off[r]);
        ReadHolder h(kit, event_mean + event_off[r], ne, NULL, shift[r], scale[r], 0.0, var[r], events_per_base[r]);
        const Alphabet* a = alphabet_by_name(alphabet);
        for(int64_t j = j0; j < j1; ++j) {
            HMMInputData d = make_data(h, kit, alphabet, e_start[j], e_stop[j], stride[j], rc[j]);
            HMMInputSequence hs(seqs[j], rc_seqs[j], a);
            out[j] = profile_hmm_score(hs, d, flags);
        }
    }
}

// ---- f2: scrappie event detection (src/thirdparty/scrappie/event_detection.c:268-319), as called by
//      SquiggleRead::load_from_raw (src/nanopolish_squiggle_read.cpp:229-236) on the whole raw table -----------------
extern "C" {
#include "event_detection.h"
}
int npref_detect_events(const float* raw, size_t n, size_t w1, size_t w2, float t1, float t2, float peak_height,
                        uint64_t* out_start, float* out_length, float* out_mean, float* out_stdv, size_t cap)
{
    raw_table rt; rt.n = n; rt.start = 0; rt.end = n; rt.raw = const_cast<float*>(raw);
    detector_param p; p.window_length1 = w1; p.window_length2 = w2; p.threshold1 = t1; p.threshold2 = t2; p.peak_height = peak_height;
    event_table et = detect_events(rt, p);
    if (et.event == NULL) return -1;
    const size_t m = et.n < cap ? et.n : cap;
    for (size_t i = 0; i < m; ++i) {
        out_start[i] = et.event[i].start; out_length[i] = et.event[i].length; out_mean[i] = et.event[i].mean; out_stdv[i] = et.event[i].stdv;
    }
    const int n_ev = (int)et.n;
    free(et.event);
    return n_ev;
}
void npref_detect_events_many(int n_reads, const float* raw, const int64_t* raw_off, float* out_mean, const int64_t* ev_off, int32_t* out_n,
                              int n_threads)
{
#pragma omp parallel for schedule(dynamic) num_threads(n_threads)
    for (int r = 0; r < n_reads; ++r) {
        raw_table rt; rt.n = (size_t)(raw_off[r + 1] - raw_off[r]); rt.start = 0; rt.end = rt.n; rt.raw = const_cast<float*>(raw + raw_off[r]);
        event_table et = detect_events(rt, event_detection_defaults);
        const size_t cap = (size_t)(ev_off[r + 1] - ev_off[r]);
        for (size_t i = 0; i < et.n && i < cap; ++i) out_mean[ev_off[r] + i] = et.event[i].mean;
        out_n[r] = (int32_t)et.n;
        free(et.event);
    }
}

} // extern "C"
