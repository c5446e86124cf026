// np_capi.hip -- the C ABI (include/np_hmm.h): context, model registry, host-buffer entry points that pack /
// upload / launch / download, and device-resident entry points that only enqueue kernels.
// No CPU fallback exists: every compute entry point needs a live gfx950 device and fails loudly otherwise.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>
#include <algorithm>
#include "np_kernels.h"
#include "np_logf.h"

#define NP_VERSION_STR "nanopolish_amd 0.1 (gfx950)"
#define NP_FLANK_LEN NP_MAX_WINDOW_EVENTS
#define NP_NUM_FAMILIES 9      // 0 event align, 1 forward HMM, 2 glue, 3 Viterbi, 4 event detection, 5 MoM scalings, 6 eventalign chain
                               // (7 and 8 belonged to round 3's split-aligner / side-stream experiments: kept as empty slots so that np_kernel_time's
                               // family numbers do not move)

namespace {

struct dev_buf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct model_t {
    int k = 0, n_states = 0;
    np_state_dev* d_states = nullptr;
    std::vector<double> level_mean;
};

struct timing_t {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> pool;
    double total_ms = 0.0;
    long launches = 0;
    float last_ms = 0.f;
};

} // namespace

struct np_ctx {
    int device = 0;
    int n_cu = 256;
    np_params params;
    hipStream_t stream = nullptr;
    hipStream_t last_stream = nullptr;   // identity of the stream of the most recent call (use_stream); never dereferenced later:
                                         // the caller may have destroyed it (torch stream pools, short-lived streams)
    bool tail_recorded = false;          // switch_ev marks the tail of the most recent call's work
    hipEvent_t switch_ev = nullptr;
    bool lse_oor = true;                 // forward kernel: clamp-free log-sum lookups (cleared when probe_hardware fails)
    bool lse_probe_ok = false;           // probe_hardware found the LDS out-of-range rule to hold: only then may option "lse_oor" select the clamp-free kernel
    std::string info;                    // np_ctx_info(): the probe's findings
    std::vector<model_t> models;
    float* d_logsum = nullptr;
    std::vector<float> h_logsum;
    float* d_flank = nullptr;
    uint32_t* d_counters = nullptr;   // [0..7] class counts, [8..15] work-queue heads, [16] align queue head, [17] chain queue head, [18] back-track queue head, [32] self-test, [1024 .. 1024 + 2 * 4096) work-item bins (np_launch_classify)
    dev_buf site_scan;                // np_genome_site_index_dev: the chunk totals of its prefix scan
    dev_buf order, trace, kparams, align_order, recal_order;   // (recal_order: the recalibration's own issue order -- the persistent aligner of another batch may still be pulling from align_order)
    void* small_h = nullptr; size_t small_h_cap = 0; dev_buf small_d;     // np_hmm_score_host's small-batch path: one pinned blob, its device twin
    int small_batch_path = 1;         // np_hmm_score_host: batches of <= NP_SMALL_BATCH items as one pinned blob (0: the general path; tests compare the two)
    // host-API staging
    dev_buf b_jobs, b_reads, b_events, b_ranks, b_out, b_pair_off, b_pairs, b_pair_begin, b_n_pairs,
            b_vm, b_bp, b_cell_off, b_state_off, b_states, b_n_states;
    dev_buf ed_status, ed_tstat;      // event detection scratch
    dev_buf cm_group_rank_off, cm_cigar_scratch;        // work-item generation scratch
    dev_buf ea_bp, ea_path, ea_args;                    // eventalign chain: per-wave back-pointer rows and path lists; a device copy of the launch arguments
    int ea_rows_cap = 4096, ea_waves_per_cu = 20;
    int ea_walk_prio = 0;            // two-read chain kernel: wave priority 3 during back-track + emission (helped the scalar walks, not the vector walk)
    dev_buf b_raw, b_raw_off, b_ev_off, b_ev_start, b_ev_len, b_ev_mean, b_ev_stdv, b_n_events;
    timing_t timing[NP_NUM_FAMILIES];
    std::mutex lock;
    std::string err;
    int align_blocks_per_cu = 8, hmm_blocks_per_cu = 2;
    int hmm_prio = 0;                 // wave priority of the forward kernels
    bool host_constants = false;      // the per-read constants that go through libm are computed on the HOST with the process's own log / exp / logf (np_create)
    double* d_log_n = nullptr;        // host_constants: log(1 .. 64) for profile_hmm_score_set's penalty
    np_slots lay = {nullptr, nullptr, 0}; int64_t lay_total = 0;   // np_set_job_layout: the slot layout of the work-item arrays of the calls that follow
    int recal_shape = 3;              // the recalibration kernel's workgroup shape (3: round 6's half-wave form, the default; 0 ... 2: rounds 5's shapes, same results)
    int align_lpt = 1;                // issue the event aligner's reads longest first
    int stream_switch_wait = 1;       // a call on a new stream waits for the tail of the stream the context used before (0: the caller orders its streams itself)
    int ed_warmup = -1;               // parallel peak walk: samples of warm-up per segment (< 0: the kernel's default)
    bool ed_ratio_exact = false;      // the fused walk's (float)(|dm| / sqrt(cvw)) by the exact sequence for every value (np_create's probe failed, or the option)
    int64_t last_align_blocks = 0, last_align_scratch = 0;
    int ed_last_reads = 0;            // np_get_stat("ed_serial_reads"): reads of the most recent event-detection call      // np_get_stat: grid and scratch of the most recent event-align launch
};

namespace {

std::string g_create_err;

#define NP_HIP(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                   \
            return NP_ERR_DEVICE;                                                              \
        }                                                                                      \
    } while (0)

struct family_timer {
    np_ctx* c; int which; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    family_timer(np_ctx* ctx, int w, hipStream_t st) : c(ctx), which(w), s(st)
    {
        timing_t& t = c->timing[which];
        auto get = [&]() { hipEvent_t e = nullptr; if (!t.pool.empty()) { e = t.pool.back(); t.pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
        a = get(); b = get();
        (void)hipEventRecord(a, s);
    }
    ~family_timer() { (void)hipEventRecord(b, s); c->timing[which].pending.push_back({a, b}); }
};

void drain_timing(np_ctx* c)
{
    for (int w = 0; w < NP_NUM_FAMILIES; ++w) {
        timing_t& t = c->timing[w];
        std::vector<std::pair<hipEvent_t, hipEvent_t>> later;
        for (auto& pr : t.pending) {
            float ms = 0.f;
            const hipError_t e = hipEventElapsedTime(&ms, pr.first, pr.second);
            if (e == hipErrorNotReady) { later.push_back(pr); continue; }      // enqueued on a stream the caller has not synchronised: next time
            if (e == hipSuccess) { t.total_ms += ms; t.launches += 1; t.last_ms = ms; }
            t.pool.push_back(pr.first); t.pool.push_back(pr.second);
        }
        if (!later.empty()) (void)hipGetLastError();
        t.pending.swap(later);
    }
}

hipStream_t pick_stream(np_ctx* c, void* s) { return s ? (hipStream_t)s : c->stream; }

// The stream an entry point enqueues on.  A context's work queues, counters and scratch are shared by all of its calls, so
// work on a new stream must not overtake what was enqueued on the previous one (include/np_hmm.h, "ONE stream at a time per
// context").  Every call records the context's switch event at the tail of what it enqueued -- on its own stream, while that
// stream is certainly alive -- and a call that arrives on a different stream makes its stream wait for that event.  The previous
// stream's handle is only compared, never used: the caller may have destroyed it in the meantime.
struct stream_scope {
    np_ctx* c; hipStream_t s;
    stream_scope(np_ctx* ctx, hipStream_t st) : c(ctx), s(st) {}
    stream_scope(const stream_scope&) = delete;
    stream_scope& operator=(const stream_scope&) = delete;
    operator hipStream_t() const { return s; }
    ~stream_scope() { c->tail_recorded = c->switch_ev && hipEventRecord(c->switch_ev, s) == hipSuccess; }
};
stream_scope use_stream(np_ctx* c, void* s)
{
    hipStream_t st = pick_stream(c, s);
    if (c->stream_switch_wait && c->tail_recorded && st != c->last_stream) (void)hipStreamWaitEvent(st, c->switch_ev, 0);
    c->last_stream = st;
    return stream_scope(c, st);
}

int persistent_blocks(np_ctx* c, int64_t work_items, int per_block, int blocks_per_cu)
{
    int64_t need = (work_items + per_block - 1) / per_block;
    int64_t maxb = (int64_t)c->n_cu * blocks_per_cu;
    if (need < 1) need = 1;
    return (int)std::min(need, maxb);
}

// ---- host-constants mode (VERDICT r4 Missing 6) ----------------------------------------------------------------------------------
// The device forms the aligner constants (raw_loader.cpp:99-108), set4's log(var) (squiggle_read.cpp:38-65), the HMM transitions
// (r9.inl:17-76) and profile_hmm_score_set's log(n) with a restatement of THIS image's glibc 2.35 (np_log.h, np_logf.h).  A reference
// built against another libm computes those constants with ITS log / exp / logf, possibly one ulp away, and "event indices bit-exact"
// would be gone with no way back short of a rebuild.  np_create therefore compares the restatement with the process's libm
// (np_selftest_libm); on a mismatch -- or with NP_HOST_CONSTANTS=1 -- the fused pass stops at the three places a constant is formed,
// reads back what it depends on (two integers, one double per read), computes it on the host with the process's own libm exactly as
// the reference's expressions do, and uploads it: a few stream synchronisations per batch instead of none, bit-identical to the
// process's libm whatever it is (tests/test_gpu_libm.py runs the chain under an LD_PRELOADed libm whose log / exp / logf are skewed).
void transitions_libm(double events_per_base, double indel_bias, float out[10])
{
    double read_events_per_base = events_per_base;
    read_events_per_base *= indel_bias;
    read_events_per_base = read_events_per_base > 1.25 ? read_events_per_base : 1.25;
    const float p_stay = (float)(1 - (1 / read_events_per_base));
    const float p_skip = 0.0025f, p_bad = 0.001f, p_bad_self = p_bad, p_skip_self = 0.3f;
    const float p_mk = p_skip, p_mb = p_bad, p_mm_self = p_stay;
    const float p_mm_next = 1.0f - p_mm_self - p_mk - p_mb;
    const float p_bb = p_bad_self;
    const float p_bk = (1.0f - p_bb) / 3;
    const float p_bm_next = p_bk, p_bm_self = p_bk;
    const float p_kk = p_skip_self;
    const float p_km = 1.0f - p_kk;
    const float p[10] = {p_mm_self, p_mb, p_mk, p_mm_next, p_bb, p_bk, p_bm_next, p_bm_self, p_kk, p_km};
    for (int i = 0; i < 10; ++i) out[i] = logf(p[i]);                       // the float overload of log, as r9.inl:61-72
}

// what: 1 = the aligner constants from (n_events, n_kmers); 2 = the transitions from events_per_base; 4 = log_var from var
int host_constants_fix(np_ctx* c, hipStream_t s, int n_reads, np_read_dev* reads, const double* events_per_base, int what)
{
    if (n_reads <= 0) return NP_OK;
    std::vector<np_read_dev> h((size_t)n_reads);
    std::vector<double> epb;
    NP_HIP(c, hipMemcpyAsync(h.data(), reads, h.size() * sizeof(np_read_dev), hipMemcpyDeviceToHost, s));
    if (what & 2) { epb.resize((size_t)n_reads); NP_HIP(c, hipMemcpyAsync(epb.data(), events_per_base, epb.size() * sizeof(double), hipMemcpyDeviceToHost, s)); }
    NP_HIP(c, hipStreamSynchronize(s));
    for (int r = 0; r < n_reads; ++r) {
        np_read_dev& q = h[(size_t)r];
        if (what & 1) {                                                     // raw_loader.cpp:99-108
            const double events_per_kmer = (double)q.n_events / (double)q.n_kmers;
            const double p_stay = 1 - (1 / (events_per_kmer + 1));
            const double epsilon = 1e-10;
            q.lp_skip = log(epsilon);
            q.lp_stay = log(p_stay);
            q.lp_step = log(1.0 - exp(q.lp_skip) - exp(q.lp_stay));
            q.lp_trim = log(0.01);
        }
        if (what & 2) transitions_libm(epb[(size_t)r], c->params.hmm_indel_bias_factor, q.trans);
        if (what & 4) q.log_var = log(q.var);                               // set4 / set6, squiggle_read.cpp:38-65
    }
    NP_HIP(c, hipMemcpyAsync(reads, h.data(), h.size() * sizeof(np_read_dev), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipStreamSynchronize(s));                                     // (h is this call's)
    return NP_OK;
}

// the slot layout a consumer of `n_jobs` work items may use: the declared one when it describes exactly this array, none otherwise
// (a declared layout with another item count is the caller's mistake: refused, not guessed at)
int layout_for(np_ctx* c, int64_t n_jobs, np_slots* out)
{
    *out = np_slots{nullptr, nullptr, 0};
    if (c->lay.n_reads <= 0 || n_jobs <= 0) return NP_OK;
    if (n_jobs != 2 * c->lay_total) { c->err = "np_set_job_layout describes another work-item array (2 x total_slots != n_jobs): clear it with n_reads = 0"; return NP_ERR_INVALID; }
    *out = c->lay;
    return NP_OK;
}

// ---- kernel B driver: classify + one persistent launch per non-empty size class ----------------------
// class_mask: bit cls set = the size class may hold work items (the *_dev callers do not know: all eight; the host entry points
// see the items and launch only the classes that occur -- a per-call round is launch-bound, and six empty persistent launches each
// still load their 64 KB table)
// use_layout: the *_dev caller's work items may be the array np_set_job_layout describes; the host entry points pack their OWN dense array and
// must never see a layout another thread declared for its *_dev calls on this context (a per-record np_hmm_score_host arriving
// between the packer's np_set_job_layout and its clearing call was refused -- or, with a matching count, scored through foreign slots)
int run_hmm_forward(np_ctx* c, hipStream_t s, int64_t n_jobs, const np_hmm_job_dev* jobs, const np_read_dev* reads,
                    const float* event_mean, const uint16_t* ranks, int model, float* out, unsigned class_mask = 0xffu, bool use_layout = true)
{
    if (n_jobs <= 0) return NP_OK;
    if (model < 0 || model >= (int)c->models.size()) { c->err = "bad model id"; return NP_ERR_INVALID; }
    if (n_jobs > 0xffffffffll) { c->err = "too many jobs in one call"; return NP_ERR_UNSUPPORTED; }
    NP_HIP(c, c->order.reserve((size_t)NP_NUM_CLASSES * (size_t)n_jobs * sizeof(uint32_t)));
    NP_HIP(c, hipMemsetAsync(c->d_counters, 0, 16 * sizeof(uint32_t), s));
    family_timer tm(c, 1, s);
    np_slots lay{nullptr, nullptr, 0};
    if (use_layout) { const int rc = layout_for(c, n_jobs, &lay); if (rc != NP_OK) return rc; }
    NP_HIP(c, np_launch_classify(jobs, n_jobs, c->d_counters, c->order.as<uint32_t>(), out, NP_FLANK_LEN, c->d_counters + 1024, lay, s));
    for (int cls = 0; cls < NP_NUM_CLASSES; ++cls) {
        if (!(class_mask >> cls & 1u)) continue;
        np_hmm_args a{};
        a.jobs = jobs; a.order = c->order.as<uint32_t>() + (size_t)cls * (size_t)n_jobs; a.n_class_jobs = c->d_counters + cls;
        a.reads = reads; a.event_mean = event_mean; a.ranks = ranks; a.model = c->models[model].d_states;
        a.logsum = c->d_logsum; a.flank = c->d_flank; a.counter = c->d_counters + 8 + cls; a.out = out; a.prio = c->hmm_prio;
        const int threads = np_hmm_block_threads(cls);
        const int jobs_per_block = (threads / 64) * (64 / NP_CLASS_SEG[cls]);
        const int nb = persistent_blocks(c, n_jobs, jobs_per_block, c->hmm_blocks_per_cu);
        NP_HIP(c, np_launch_hmm_forward(cls, a, nb, c->lse_oor, s));
    }
    return NP_OK;
}

// fill and back-track of a read in one kernel (trace scratch per resident wave)
int run_event_align(np_ctx* c, hipStream_t s, int n_reads, const np_read_dev* reads, const float* event_mean,
                    const uint16_t* ranks, int model, int64_t max_bands, const int64_t* pair_off, np_pair* pairs,
                    int32_t* pair_begin, int32_t* n_pairs)
{
    if (n_reads <= 0) return NP_OK;
    if (model < 0 || model >= (int)c->models.size()) { c->err = "bad model id"; return NP_ERR_INVALID; }
    const int waves_per_block = np_align_block_threads() / 64;
    int nb = persistent_blocks(c, n_reads, waves_per_block, c->align_blocks_per_cu);
    // per-resident-wave scratch: packed trace (32 B per band) and the k-mer parameter slab (16 B per k-mer).
    // Ultra-long reads make the slabs large, so the persistent grid shrinks to keep the scratch under a budget
    // (a 1M-event read needs ~56 MB per wave: 48 GB would hold ~850 resident waves instead of 5120).
    const uint64_t stride = (((uint64_t)max_bands + 7) / 8) * 32;        // u64 units: one 256-byte row per 8 bands
    const uint64_t kp_stride = ((uint64_t)max_bands + 63) & ~63ull;      // k-mers per read < bands per read
    const uint64_t per_block = (uint64_t)waves_per_block * (stride * sizeof(uint64_t) + kp_stride * sizeof(float4));
    const uint64_t budget = 48ull << 30;
    if ((uint64_t)nb * per_block > budget) nb = (int)std::max<uint64_t>(1, budget / per_block);
    c->last_align_blocks = nb; c->last_align_scratch = (int64_t)((uint64_t)nb * per_block);
    NP_HIP(c, c->trace.reserve((size_t)nb * waves_per_block * stride * sizeof(uint64_t)));
    uint32_t* counter = c->d_counters + 16;
    NP_HIP(c, c->kparams.reserve((size_t)nb * waves_per_block * kp_stride * sizeof(float4)));
    NP_HIP(c, c->align_order.reserve((size_t)(2048 + n_reads) * sizeof(uint32_t)));
    NP_HIP(c, hipMemsetAsync(counter, 0, sizeof(uint32_t), s));
    np_align_args a{};
    a.reads = reads; a.event_mean = event_mean; a.ranks = ranks; a.model = c->models[model].d_states;
    a.pair_off = pair_off; a.pairs = pairs; a.pair_begin = pair_begin; a.n_pairs = n_pairs;
    a.trace = c->trace.as<uint64_t>(); a.trace_stride = stride; a.kparams = c->kparams.as<float4>(); a.kp_stride = kp_stride;
    a.counter = counter;
    a.n_reads = n_reads; a.max_gap_threshold = c->params.max_gap_threshold;
    a.min_average_log_emission = c->params.min_average_log_emission;
    family_timer tm(c, 0, s);
    if (c->align_lpt && n_reads > nb * waves_per_block) {       // more reads than resident waves: the issue order matters
        NP_HIP(c, np_launch_align_order(n_reads, reads, c->align_order.as<uint32_t>(), s));
        a.order = c->align_order.as<uint32_t>() + 2048;
    }
    NP_HIP(c, np_launch_event_align(a, nb, s));
    return NP_OK;
}

} // namespace

// The emission's exact division (np_device.h:np_div_exact: reciprocal + two fused corrections, no range scaling) is verified
// against the IEEE divide for divisors in [2^-5, 2^6) and numerators of magnitude < 2^15 (np_selftest_division): a model
// whose scaled sigma can leave that range is refused here rather than scored with an unverified quotient.  (A read's `var`
// multiplies sigma by 1 .. 2.5 before the calibration gate drops the read; levels are pA, |x - mean| < 2^15.)
static bool model_in_range(np_ctx* c, int n_states, const double* level_mean, const double* level_stdv)
{
    for (int i = 0; i < n_states; ++i) {
        if (!(level_stdv[i] >= 0.0625 && level_stdv[i] <= 16.0) || !(fabs(level_mean[i]) < 8192.0)) {
            c->err = "np_register_model: level_stdv outside [1/16, 16] or |level_mean| >= 8192: outside the range the exact emission division is verified for";
            return false;
        }
    }
    return true;
}


// Hardware / toolchain assumptions the clamp-free fast paths rest on, checked once per context (np_hmm_kernels.hip, "Hardware
// probes").  (a) every forward kernel's static LDS is exactly the log-sum table, (b) LDS reads past the allocation return 0 and
// np_lse_oor agrees with np_lse bit for bit, (c) range-checked buffer loads return 0 / stores are dropped outside the descriptor.
// (a) or (b) failing selects the clamped log-sum (slower, same results); (c) failing is fatal: the event aligner and the
// eventalign chain have no other form.
static bool probe_hardware(np_ctx* c)
{
    char line[512];
    bool lds_size_ok = true;
    size_t lds_seen = 0;
    for (int cls = 0; cls < NP_NUM_CLASSES; ++cls) {
        size_t b = 0;
        if (np_hmm_forward_lds_bytes(cls, &b) != hipSuccess || b != NP_LOGSUM_TBL * sizeof(float)) lds_size_ok = false;
        lds_seen = std::max(lds_seen, b);
    }
    float* d_buf = nullptr; uint16_t* d_sbuf = nullptr; uint32_t* d_out = c->d_counters + 64;     // 8 words, zeroed at np_create
    std::vector<float> ones(32, 1.0f);
    std::vector<uint16_t> pat(32, (uint16_t)0xa5a5u);
    uint32_t out[8] = {0};
    bool ran = hipMalloc((void**)&d_buf, ones.size() * sizeof(float)) == hipSuccess && hipMalloc((void**)&d_sbuf, pat.size() * sizeof(uint16_t)) == hipSuccess;
    ran = ran && hipMemcpy(d_buf, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    ran = ran && hipMemcpy(d_sbuf, pat.data(), pat.size() * sizeof(uint16_t), hipMemcpyHostToDevice) == hipSuccess;
    ran = ran && hipMemset(d_out, 0, sizeof(out)) == hipSuccess;
    ran = ran && np_launch_probe(c->d_logsum, d_buf, d_sbuf, d_out, 4 * c->n_cu, c->stream) == hipSuccess;
    ran = ran && hipStreamSynchronize(c->stream) == hipSuccess;
    ran = ran && hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost) == hipSuccess;
    ran = ran && hipMemcpy(pat.data(), d_sbuf, pat.size() * sizeof(uint16_t), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipMemset(d_out, 0, sizeof(out));
    if (d_buf) (void)hipFree(d_buf);
    if (d_sbuf) (void)hipFree(d_sbuf);
    (void)hipGetLastError();
    bool store_ok = ran;
    for (int i = 0; i < 16 && store_ok; ++i) store_ok = pat[i] == 0x1234u && pat[16 + i] == 0xa5a5u;
    const bool ran_all = ran && out[3] == (uint32_t)(4 * c->n_cu);
    const bool lds_ok = ran_all && lds_size_ok && out[0] == 0 && out[1] == 0;
    const bool buf_ok = ran_all && out[2] == 0 && store_ok;
    const char* forced = getenv("NP_LSE_CLAMP");
    c->lse_probe_ok = lds_ok;
    c->lse_oor = lds_ok && !(forced && atoi(forced) != 0);
    snprintf(line, sizeof(line), "probe: forward-kernel LDS %zu B (%s), reads past the LDS allocation %s (%u non-zero words, %u log-sum mismatches), "
             "range-checked buffer access %s (%u bad loads, stores %s); log-sum lookup: %s",
             lds_seen, lds_size_ok ? "the table only" : "NOT the table only", ran_all && out[1] == 0 ? "return 0" : "DO NOT return 0", out[1], out[0],
             buf_ok ? "ok" : "BROKEN", out[2], store_ok ? "dropped out of range" : "NOT dropped out of range",
             c->lse_oor ? "clamp-free (np_lse_oor)" : (lds_ok ? "clamped (NP_LSE_CLAMP)" : "clamped (probe failed)"));
    c->info = line;
    if (const char* v = getenv("NP_VERBOSE")) if (atoi(v) != 0) fprintf(stderr, "nanopolish_amd: %s\n", line);
    if (!buf_ok) { g_create_err = std::string("np_create: hardware probe failed: ") + line; return false; }
    return true;
}

extern "C" {

const char* np_version(void) { return NP_VERSION_STR; }

const char* np_ctx_info(const np_ctx* ctx) { return ctx ? ctx->info.c_str() : ""; }

const char* np_last_error(const np_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

np_ctx* np_create(int device, const np_params* params)
{
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_create_err = std::string("np_create: no HIP device (") + hipGetErrorString(e) + "); this library has no CPU fallback";
        return nullptr;
    }
    if (device < 0 || device >= n_dev) { g_create_err = "np_create: bad device ordinal"; return nullptr; }
    if ((e = hipSetDevice(device)) != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { g_create_err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return nullptr; }
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_create_err = std::string("np_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
        return nullptr;
    }
    np_ctx* c = new np_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    // tuning knobs (persistent-grid sizes); defaults fill the CU up to the kernels' register-limited occupancy
    if (const char* v = getenv("NP_ALIGN_BLOCKS_PER_CU")) c->align_blocks_per_cu = std::max(1, atoi(v));
    if (const char* v = getenv("NP_HMM_BLOCKS_PER_CU")) c->hmm_blocks_per_cu = std::max(1, atoi(v));
    if (const char* v = getenv("NP_ALIGN_LPT")) c->align_lpt = atoi(v) != 0;
    if (const char* v = getenv("NP_RECAL_SHAPE")) c->recal_shape = std::max(0, std::min(3, atoi(v)));
    if (const char* v = getenv("NP_ED_WARMUP")) c->ed_warmup = atoi(v);
    if (const char* v = getenv("NP_EA_WAVES_PER_CU")) c->ea_waves_per_cu = std::max(1, atoi(v));
    if (const char* v = getenv("NP_EA_WALK_PRIO")) c->ea_walk_prio = atoi(v);
    if (params) c->params = *params; else np_default_params(&c->params);
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->switch_ev, hipEventDisableTiming) == hipSuccess;

    // p7_FLogsum table, src/common/logsum.cpp:57-69 (host libm, as the reference's static initialiser)
    std::vector<float> tbl(NP_LOGSUM_TBL);
    for (int i = 0; i < NP_LOGSUM_TBL; i++) tbl[i] = (float)log(1. + exp((double)-i / 1000.f));
    // universal clip-flank table: pre_flank[i] (src/hmm/nanopolish_profile_hmm_r9.inl:204-226); post_flank[i] of an
    // e-event window is the same sequence read backwards, post_flank[i] == flank[e-1-i] (r9.inl:236-259).
    std::vector<float> flank(NP_FLANK_LEN);
    {
        const double TRANS_CLIP_SELF = 0.9, TRANS_START_TO_CLIP = 0.5;
        const float bg = -3.0f;   // log_probability_background, src/hmm/nanopolish_emissions.h:98-103
        flank[0] = (float)log(1 - TRANS_START_TO_CLIP);
        flank[1] = (float)(log(TRANS_START_TO_CLIP) + bg + log(1 - TRANS_CLIP_SELF));
        for (size_t i = 2; i < flank.size(); ++i) flank[i] = (float)(log(TRANS_CLIP_SELF) + bg + flank[i - 1]);
    }
    // the device code carries (float)log(0.3989422804014327) as a literal: check it against this host's libm
    if ((float)log(0.3989422804014327) != -0.918938518f) { g_create_err = "np_create: log_inv_sqrt_2pi literal mismatch"; ok = false; }

    c->h_logsum = tbl;
    ok = ok && hipMalloc((void**)&c->d_logsum, tbl.size() * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_flank, flank.size() * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_counters, 16384 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemcpy(c->d_logsum, tbl.data(), tbl.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(c->d_flank, flank.data(), flank.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(c->d_counters, 0, 16384 * sizeof(uint32_t)) == hipSuccess;
    if (!ok) {
        if (g_create_err.empty()) g_create_err = "np_create: device allocation failed";
        np_destroy(c);
        return nullptr;
    }
    g_create_err.clear();
    if (!probe_hardware(c)) { np_destroy(c); return nullptr; }
    {
        // the fused detector walk trusts a once-refined v_rsq_f64 for the t-statistic's last step wherever the result stays clear of the float
        // rounding boundaries (np_events_kernels.hip:ed_ratio_filtered); that rests on the instruction's accuracy: checked here on 2^22 operand
        // pairs -- a device on which a trusted value differs from the exact sequence runs the exact sequence for every value instead
        uint64_t bad = 0, near = 0, far = 0;
        const int rc = np_selftest_tstat_ratio(c, 1ull << 22, 20260930, &bad, &near, &far);
        c->ed_ratio_exact = rc != NP_OK || bad != 0 || far >= 4096;
        if (const char* v = getenv("NP_ED_RATIO_EXACT")) c->ed_ratio_exact = atoi(v) != 0;
        char line[200];
        snprintf(line, sizeof(line), "; t-statistic ratio: %s (probe: %llu trusted values differ, farthest unfiltered disagreement %llu of a band of 16384)",
                 c->ed_ratio_exact ? "exact sequence for every value" : "filtered", (unsigned long long)bad, (unsigned long long)far);
        c->info += line;
    }
    {
        uint64_t bad = 0;
        (void)np_selftest_libm(100000, 20260925, &bad);
        const char* hc = getenv("NP_HOST_CONSTANTS");
        c->host_constants = hc ? atoi(hc) != 0 : bad != 0;
        if (c->host_constants) {
            double ln[65]; ln[0] = 0.0;
            for (int i = 1; i <= 64; ++i) ln[i] = log((double)i);
            if (hipMalloc((void**)&c->d_log_n, sizeof(ln)) != hipSuccess || hipMemcpy(c->d_log_n, ln, sizeof(ln), hipMemcpyHostToDevice) != hipSuccess) {
                g_create_err = "np_create: device allocation failed"; np_destroy(c); return nullptr;
            }
        }
        char line[256];
        snprintf(line, sizeof(line), "; libm check: %llu of 300000 log / exp / logf values differ from the restatement -> per-read constants on the %s",
                 (unsigned long long)bad, c->host_constants ? "HOST (process libm)" : "device (glibc 2.35 restated)");
        c->info += line;
        if (const char* v = getenv("NP_VERBOSE")) if (atoi(v) != 0) fprintf(stderr, "nanopolish_amd: %s\n", line + 2);
    }
    return c;
}

void np_destroy(np_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& m : c->models) if (m.d_states) (void)hipFree(m.d_states);
    if (c->d_logsum) (void)hipFree(c->d_logsum);
    if (c->d_flank) (void)hipFree(c->d_flank);
    if (c->d_log_n) (void)hipFree(c->d_log_n);
    if (c->d_counters) (void)hipFree(c->d_counters);
    dev_buf* bufs[] = {&c->order, &c->trace, &c->kparams, &c->b_jobs, &c->b_reads, &c->b_events, &c->b_ranks, &c->b_out, &c->b_pair_off,
                       &c->b_pairs, &c->b_pair_begin, &c->b_n_pairs, &c->b_vm, &c->b_bp, &c->b_cell_off, &c->b_state_off,
                       &c->b_states, &c->b_n_states, &c->ed_status, &c->ed_tstat, &c->b_raw, &c->b_raw_off, &c->b_ev_off, &c->b_ev_start,
                       &c->b_ev_len, &c->b_ev_mean, &c->b_ev_stdv, &c->b_n_events, &c->cm_group_rank_off, &c->cm_cigar_scratch, &c->ea_bp, &c->ea_path, &c->ea_args, &c->align_order, &c->recal_order, &c->site_scan};
    for (dev_buf* b : bufs) b->release();
    c->small_d.release();
    if (c->small_h) (void)hipHostFree(c->small_h);
    for (auto& t : c->timing) {
        for (auto& pr : t.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        for (auto& ev : t.pool) (void)hipEventDestroy(ev);
    }
    if (c->switch_ev) (void)hipEventDestroy(c->switch_ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int np_register_model(np_ctx* c, int k, int n_states, const double* level_mean, const double* level_stdv,
                      const double* level_log_stdv)
{
    if (!c || n_states <= 0 || n_states > 65536 || !level_mean || !level_stdv || !level_log_stdv) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    if (!model_in_range(c, n_states, level_mean, level_stdv)) return NP_ERR_UNSUPPORTED;
    std::vector<np_state_dev> st(n_states);
    for (int i = 0; i < n_states; ++i) { st[i].level_mean = level_mean[i]; st[i].level_stdv = level_stdv[i]; st[i].level_log_stdv = level_log_stdv[i]; st[i].pad = 0; }
    model_t m; m.k = k; m.n_states = n_states; m.level_mean.assign(level_mean, level_mean + n_states);
    NP_HIP(c, hipMalloc((void**)&m.d_states, st.size() * sizeof(np_state_dev)));
    NP_HIP(c, hipMemcpy(m.d_states, st.data(), st.size() * sizeof(np_state_dev), hipMemcpyHostToDevice));
    c->models.push_back(std::move(m));
    return (int)c->models.size() - 1;
}

int np_update_model(np_ctx* c, int model, int n_states, const double* level_mean, const double* level_stdv,
                    const double* level_log_stdv)
{
    if (!c || !level_mean || !level_stdv || !level_log_stdv) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    if (model < 0 || model >= (int)c->models.size() || n_states != c->models[model].n_states) { c->err = "np_update_model: bad model id or size"; return NP_ERR_INVALID; }
    if (!model_in_range(c, n_states, level_mean, level_stdv)) return NP_ERR_UNSUPPORTED;
    NP_HIP(c, hipSetDevice(c->device));
    if (c->tail_recorded) NP_HIP(c, hipEventSynchronize(c->switch_ev));           // kernels in flight still read the old table
    NP_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<np_state_dev> st(n_states);
    for (int i = 0; i < n_states; ++i) { st[i].level_mean = level_mean[i]; st[i].level_stdv = level_stdv[i]; st[i].level_log_stdv = level_log_stdv[i]; st[i].pad = 0; }
    NP_HIP(c, hipMemcpy(c->models[model].d_states, st.data(), st.size() * sizeof(np_state_dev), hipMemcpyHostToDevice));
    c->models[model].level_mean.assign(level_mean, level_mean + n_states);
    return NP_OK;
}

// Device self-test of the exact fast division used by the emission (np_device.h:np_div_exact) against `/`.
int np_selftest_division(np_ctx* c, uint64_t n_samples, uint64_t seed, uint64_t* n_mismatch)
{
    if (!c || !n_mismatch) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    unsigned long long* d = (unsigned long long*)(c->d_counters + 32);
    stream_scope scope = use_stream(c, nullptr);
    NP_HIP(c, hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream));
    NP_HIP(c, np_launch_selftest_div(n_samples, seed, d, c->stream));
    unsigned long long h = 0;
    NP_HIP(c, hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    NP_HIP(c, hipStreamSynchronize(c->stream));
    *n_mismatch = h;
    return NP_OK;
}

int np_selftest_tstat_ratio(np_ctx* c, uint64_t n_samples, uint64_t seed, uint64_t* n_mismatch, uint64_t* n_sent_to_exact, uint64_t* farthest_disagreement)
{
    if (!c || !n_mismatch) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    unsigned long long* d = (unsigned long long*)(c->d_counters + 32);
    stream_scope scope = use_stream(c, nullptr);
    NP_HIP(c, hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), c->stream));
    NP_HIP(c, np_launch_selftest_ratio(n_samples, seed, d, c->stream));
    unsigned long long h[3] = {0, 0, 0};
    NP_HIP(c, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    NP_HIP(c, hipStreamSynchronize(c->stream));
    *n_mismatch = h[0];
    if (n_sent_to_exact) *n_sent_to_exact = h[1];
    if (farthest_disagreement) *farthest_disagreement = h[2];
    return NP_OK;
}

int np_selftest_division_small(np_ctx* c, int w, uint64_t n_f64, uint64_t* n_mismatch_f32, uint64_t* n_mismatch_f64, uint64_t* n_f32_compared)
{
    if (!c || w < 2 || w > 1024 || !n_mismatch_f32 || !n_mismatch_f64) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    unsigned long long* d = (unsigned long long*)(c->d_counters + 32);
    stream_scope scope = use_stream(c, nullptr);
    NP_HIP(c, hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), c->stream));
    // ch = RN(1 / w), cl = RN(1 / w - ch): the residual 1 - w ch is exact in one fma, its quotient by w is the low part
    const double wd = (double)w, chd = 1.0 / wd, cld = std::fma(-wd, chd, 1.0) / wd;
    const float chf = (float)chd, clf = (float)((chd - (double)chf) + cld);
    NP_HIP(c, np_launch_selftest_div_small(w, chd, cld, chf, clf, n_f64, d, c->stream));
    unsigned long long h[3] = {0, 0, 0};
    NP_HIP(c, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    NP_HIP(c, hipStreamSynchronize(c->stream));
    *n_mismatch_f32 = h[0]; *n_mismatch_f64 = h[1];
    if (n_f32_compared) *n_f32_compared = h[2];
    return NP_OK;
}

void* np_dev_alloc(np_ctx* c, size_t bytes)
{
    if (!c) return nullptr;
    std::lock_guard<std::mutex> g(c->lock);
    void* p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    const hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { c->err = std::string("np_dev_alloc: ") + hipGetErrorString(e); return nullptr; }
    return p;
}

void np_dev_free(np_ctx* c, void* p)
{
    if (!c || !p) return;
    std::lock_guard<std::mutex> g(c->lock);
    (void)hipSetDevice(c->device);
    (void)hipFree(p);
}

// Copies and fills are plain stream operations: they touch none of the context's shared scratch, so they take no part in the
// one-stream-at-a-time rule (no switch wait, no tail event) -- a caller that uploads on one stream while the context computes on
// another orders the two with its own events (np_event_record / np_stream_wait_event), as NpBatchPipeline does.
int np_copy_to_device(np_ctx* c, void* stream, void* dst, const void* src, size_t bytes)
{
    if (!c || (bytes && (!dst || !src))) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    if (bytes) NP_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, pick_stream(c, stream)));
    return NP_OK;
}

int np_copy_to_host(np_ctx* c, void* stream, void* dst, const void* src, size_t bytes)
{
    if (!c || (bytes && (!dst || !src))) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    if (bytes) NP_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, pick_stream(c, stream)));
    return NP_OK;
}

int np_memset_dev(np_ctx* c, void* stream, void* dst, int value, size_t bytes)
{
    if (!c || (bytes && !dst)) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    if (bytes) NP_HIP(c, hipMemsetAsync(dst, value, bytes, pick_stream(c, stream)));
    return NP_OK;
}

// ---- streams, events and pinned host memory for bindings that are not HIP programs themselves ------------------------------
void* np_host_alloc(np_ctx* c, size_t bytes)
{
    if (!c) return nullptr;
    std::lock_guard<std::mutex> g(c->lock);
    void* p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { c->err = std::string("np_host_alloc: ") + hipGetErrorString(e); return nullptr; }
    return p;
}

void np_host_free(np_ctx* c, void* p)
{
    if (!c || !p) return;
    std::lock_guard<std::mutex> g(c->lock);
    (void)hipSetDevice(c->device);
    (void)hipHostFree(p);
}

void* np_stream_create(np_ctx* c)
{
    if (!c) return nullptr;
    std::lock_guard<std::mutex> g(c->lock);
    hipStream_t s = nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { c->err = std::string("np_stream_create: ") + hipGetErrorString(e); return nullptr; }
    return (void*)s;
}

void np_stream_destroy(np_ctx* c, void* stream)
{
    if (!c || !stream) return;
    std::lock_guard<std::mutex> g(c->lock);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize((hipStream_t)stream);
    if (c->last_stream == (hipStream_t)stream) { c->last_stream = nullptr; }      // (its tail event stays valid: events outlive streams)
    (void)hipStreamDestroy((hipStream_t)stream);
}

void* np_event_create(np_ctx* c)
{
    if (!c) return nullptr;
    std::lock_guard<std::mutex> g(c->lock);
    hipEvent_t e = nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync);
    if (rc != hipSuccess) { c->err = std::string("np_event_create: ") + hipGetErrorString(rc); return nullptr; }
    return (void*)e;
}

void np_event_destroy(np_ctx* c, void* ev)
{
    if (!c || !ev) return;
    std::lock_guard<std::mutex> g(c->lock);
    (void)hipSetDevice(c->device);
    (void)hipEventDestroy((hipEvent_t)ev);
}

int np_event_record(np_ctx* c, void* ev, void* stream)
{
    if (!c || !ev) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    NP_HIP(c, hipEventRecord((hipEvent_t)ev, pick_stream(c, stream)));
    return NP_OK;
}

int np_stream_wait_event(np_ctx* c, void* stream, void* ev)
{
    if (!c || !ev) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    NP_HIP(c, hipStreamWaitEvent(pick_stream(c, stream), (hipEvent_t)ev, 0));
    return NP_OK;
}

int np_event_sync(np_ctx* c, void* ev)
{
    if (!c || !ev) return NP_ERR_INVALID;
    NP_HIP(c, hipEventSynchronize((hipEvent_t)ev));       // (no lock: other threads keep enqueueing while this one waits)
    return NP_OK;
}

int np_event_query(np_ctx* c, void* ev)
{
    if (!c || !ev) return NP_ERR_INVALID;
    const hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipSuccess) return NP_OK;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 1; }
    c->err = std::string("np_event_query: ") + hipGetErrorString(e);
    return NP_ERR_DEVICE;
}

int np_sync(np_ctx* c, void* stream)
{
    if (!c) return NP_ERR_INVALID;
    NP_HIP(c, hipStreamSynchronize(pick_stream(c, stream)));
    std::lock_guard<std::mutex> g(c->lock);          // (other threads may be enqueueing: the timers' event lists are shared)
    drain_timing(c);
    return NP_OK;
}

int np_kernel_time(np_ctx* c, int which, double* total_ms, int64_t* launches, int reset)
{
    if (!c || which < 0 || which >= NP_NUM_FAMILIES) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    drain_timing(c);
    if (total_ms) *total_ms = c->timing[which].total_ms;
    if (launches) *launches = c->timing[which].launches;
    if (reset) { c->timing[which].total_ms = 0; c->timing[which].launches = 0; }
    return NP_OK;
}

int np_last_kernel_ms(np_ctx* c, int which, float* ms)
{
    if (!c || which < 0 || which >= NP_NUM_FAMILIES || !ms) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    drain_timing(c);
    *ms = c->timing[which].last_ms;
    return NP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// device-resident entry points
// ---------------------------------------------------------------------------------------------------------
int np_event_align_dev(np_ctx* c, void* stream, int n_reads, const np_read_dev* reads, const float* event_mean,
                       const uint16_t* kmer_rank, int model, int64_t max_bands, const int64_t* pair_off,
                       np_pair* pairs_out, int32_t* pair_begin, int32_t* n_pairs)
{
    if (!c) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    return run_event_align(c, use_stream(c, stream), n_reads, reads, event_mean, kmer_rank, model, max_bands,
                           pair_off, pairs_out, pair_begin, n_pairs);
}

int np_hmm_score_dev(np_ctx* c, void* stream, int64_t n_jobs, const np_hmm_job_dev* jobs, const np_read_dev* reads,
                     const float* event_mean, const uint16_t* job_kmer_rank, int model, float* out_scores)
{
    if (!c) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    return run_hmm_forward(c, use_stream(c, stream), n_jobs, jobs, reads, event_mean, job_kmer_rank, model, out_scores);
}

// (verdict: device int32[n_reads] out, or null.  The exactness verdict of every read is an explicit by-product the caller hands to
//  np_detect_events_checked_dev; round 5 kept it as hidden context state matched by pointer identity -- VERDICT r5 Weak 9)
static int adc_to_pa_locked(np_ctx* c, void* stream, int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples,
                            const float* offset, const float* raw_unit, float* raw_pa, int32_t* verdict)
{
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    family_timer tm(c, 4, s);
    NP_HIP(c, np_launch_adc_to_pa(n_reads, adc, raw_off, max_samples, offset, raw_unit, raw_pa, (n_reads > 0 && max_samples > 0) ? verdict : nullptr, s));
    return NP_OK;
}

int np_adc_to_pa_dev(np_ctx* c, void* stream, int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples,
                     const float* offset, const float* raw_unit, float* raw_pa)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!adc || !raw_off || !offset || !raw_unit || !raw_pa))) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    return adc_to_pa_locked(c, stream, n_reads, adc, raw_off, max_samples, offset, raw_unit, raw_pa, nullptr);
}

int np_adc_to_pa_checked_dev(np_ctx* c, void* stream, int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples,
                             const float* offset, const float* raw_unit, float* raw_pa, int32_t* verdict)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!adc || !raw_off || !offset || !raw_unit || !raw_pa || !verdict))) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    return adc_to_pa_locked(c, stream, n_reads, adc, raw_off, max_samples, offset, raw_unit, raw_pa, verdict);
}

int np_site_table_dev(np_ctx* c, void* stream, int64_t n_groups, const float* scores, const int32_t* first_site,
                      const int32_t* n_motif, const np_hmm_job_dev* jobs, const int64_t* read_base, double call_threshold,
                      int64_t n_pos, int32_t* table)
{
    if (!c || n_groups < 0 || n_pos < 0 || (n_groups > 0 && (!scores || !first_site || !n_motif || !table)) || (read_base && !jobs)) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    family_timer tm(c, 2, s);
    NP_HIP(c, np_launch_site_table(n_groups, scores, first_site, n_motif, jobs, read_base, call_threshold, n_pos, table, s));
    return NP_OK;
}

static int site_table_genome_locked(np_ctx* c, void* stream, int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* last_site,
                                    const int32_t* n_motif, const np_hmm_job_dev* jobs, const int64_t* read_base, const char* genome,
                                    const int64_t* contig_off, int n_contigs, int alphabet, int min_separation, double call_threshold, int64_t n_pos,
                                    int32_t* table, uint64_t* n_overflow, const uint64_t* site_mask, const uint32_t* word_rank)
{
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    family_timer tm(c, 2, s);
    NP_HIP(c, np_launch_site_table_genome(n_groups, scores, first_site, last_site, n_motif, jobs, read_base, genome, contig_off, n_contigs, alphabet,
                                          min_separation, call_threshold, n_pos, table, (unsigned long long*)n_overflow, site_mask, word_rank, s));
    return NP_OK;
}

int np_site_table_genome_dev(np_ctx* c, void* stream, int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* last_site,
                             const int32_t* n_motif, const np_hmm_job_dev* jobs, const int64_t* read_base, const char* genome,
                             const int64_t* contig_off, int n_contigs, int alphabet, int min_separation, double call_threshold, int64_t n_pos,
                             int32_t* table, uint64_t* n_overflow)
{
    if (!c || n_groups < 0 || n_pos < 0 || n_contigs < 1 || min_separation < 0 || alphabet < 1 || alphabet > 4 ||
        (n_groups > 0 && (!scores || !first_site || !last_site || !n_motif || !jobs || !read_base || !genome || !contig_off || !table || !n_overflow)))
        return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    return site_table_genome_locked(c, stream, n_groups, scores, first_site, last_site, n_motif, jobs, read_base, genome, contig_off, n_contigs, alphabet,
                                    min_separation, call_threshold, n_pos, table, n_overflow, nullptr, nullptr);
}

int np_site_table_genome_indexed_dev(np_ctx* c, void* stream, int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* last_site,
                                     const int32_t* n_motif, const np_hmm_job_dev* jobs, const int64_t* read_base, const char* genome,
                                     const int64_t* contig_off, int n_contigs, int alphabet, int min_separation, double call_threshold, int64_t n_pos,
                                     const uint64_t* site_mask, const uint32_t* word_rank, int32_t* table, uint64_t* n_overflow)
{
    if (!c || n_groups < 0 || n_pos < 0 || n_contigs < 1 || min_separation < 0 || alphabet < 1 || alphabet > 4 || !site_mask || !word_rank ||
        (n_groups > 0 && (!scores || !first_site || !last_site || !n_motif || !jobs || !read_base || !genome || !contig_off || !table || !n_overflow)))
        return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    return site_table_genome_locked(c, stream, n_groups, scores, first_site, last_site, n_motif, jobs, read_base, genome, contig_off, n_contigs, alphabet,
                                    min_separation, call_threshold, n_pos, table, n_overflow, site_mask, word_rank);
}

int np_genome_site_index_dev(np_ctx* c, void* stream, const char* genome, const int64_t* contig_off, int n_contigs, int alphabet, int64_t n_pos,
                             uint64_t* site_mask, uint32_t* word_rank, int64_t* n_sites)
{
    if (!c || n_pos < 0 || n_contigs < 1 || alphabet < 1 || alphabet > 4 || !n_sites || (n_pos > 0 && (!genome || !contig_off || !site_mask || !word_rank)))
        return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    NP_HIP(c, c->site_scan.reserve((size_t)std::max<int64_t>(1, np_site_rank_chunks(n_pos)) * sizeof(uint32_t)));
    family_timer tm(c, 2, s);
    NP_HIP(c, np_launch_genome_site_index(genome, contig_off, n_contigs, alphabet, n_pos, site_mask, word_rank, n_sites, c->site_scan.as<uint32_t>(), s));
    return NP_OK;
}

int np_hmm_score_set_combine_dev(np_ctx* c, void* stream, int64_t n_sets, const int64_t* set_off, const int64_t* member_idx,
                                 const float* member_scores, float* out_scores)
{
    if (!c || n_sets < 0 || (n_sets > 0 && (!set_off || !member_scores || !out_scores))) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    family_timer tm(c, 1, s);
    NP_HIP(c, np_launch_score_set_combine(n_sets, set_off, member_idx, member_scores, c->d_logsum, c->host_constants ? c->d_log_n : nullptr, out_scores, s));
    return NP_OK;
}

int np_resolve_jobs_dev(np_ctx* c, void* stream, int n_reads, np_read_dev* reads, const int64_t* pair_off,
                        const np_pair* pairs, const int32_t* pair_begin, const int32_t* n_pairs, int32_t* map_start,
                        double* events_per_base, int64_t n_jobs, np_hmm_job_dev* jobs, const int32_t* kpos)
{
    if (!c) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    family_timer tm(c, 2, s);
    NP_HIP(c, np_launch_build_map(n_reads, reads, pair_off, pairs, pair_begin, n_pairs, map_start, nullptr, events_per_base,
                                  c->params.hmm_indel_bias_factor, s));
    if (c->host_constants) { const int rc = host_constants_fix(c, s, n_reads, reads, events_per_base, 2); if (rc != NP_OK) return rc; }
    np_slots lay;
    { const int rc = layout_for(c, n_jobs, &lay); if (rc != NP_OK) return rc; }
    NP_HIP(c, np_launch_resolve(n_jobs, jobs, reads, n_pairs, events_per_base, nullptr, map_start, kpos, lay, s));
    return NP_OK;
}

// The same glue with the calibration step of load_from_raw in between: event map ->
// recalibrate_model on the device -> work-item bounds.  `reads` comes in with any scalings and leaves with the
// calibrated shift/scale/var/log_var; reads that are not calibrated (< 200 'M' events) or exceed
// MIN_CALIBRATION_VAR get calibrated[r] = 0 and all their work items are skipped, as their events are cleared in
// the reference (squiggle_read.cpp:320-323).
int np_calibrate_resolve_dev(np_ctx* c, void* stream, int n_reads, np_read_dev* reads, const float* event_mean,
                             const uint16_t* kmer_rank, int model, const int64_t* pair_off, const np_pair* pairs,
                             const int32_t* pair_begin, const int32_t* n_pairs, int32_t* map_start, int32_t* map_stop,
                             double* events_per_base, int32_t* calibrated, int64_t n_jobs, np_hmm_job_dev* jobs,
                             const int32_t* kpos)
{
    if (!c || !calibrated) return NP_ERR_INVALID;
    if (model < 0 || model >= (int)c->models.size()) { c->err = "bad model id"; return NP_ERR_INVALID; }
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    family_timer tm(c, 2, s);
    NP_HIP(c, np_launch_build_map(n_reads, reads, pair_off, pairs, pair_begin, n_pairs, map_start, map_stop, events_per_base,
                                  c->params.hmm_indel_bias_factor, s));
    // the calibration kernel takes several reads per wave and runs for the longest of them: groups of similar length (the aligner's order)
    const uint32_t* order = nullptr;
    if (n_reads > 64) {
        NP_HIP(c, c->recal_order.reserve((size_t)(2048 + n_reads) * sizeof(uint32_t)));
        NP_HIP(c, np_launch_align_order(n_reads, reads, c->recal_order.as<uint32_t>(), s));
        order = c->recal_order.as<uint32_t>() + 2048;
    }
    NP_HIP(c, np_launch_recalibrate(n_reads, reads, event_mean, kmer_rank, c->models[model].d_states, c->models[model].n_states, n_pairs, map_start,
                                    calibrated, order, c->recal_shape, s));
    if (c->host_constants) { const int rc = host_constants_fix(c, s, n_reads, reads, events_per_base, 2 | 4); if (rc != NP_OK) return rc; }
    np_slots lay;
    { const int rc = layout_for(c, n_jobs, &lay); if (rc != NP_OK) return rc; }
    NP_HIP(c, np_launch_resolve(n_jobs, jobs, reads, n_pairs, events_per_base, calibrated, map_start, kpos, lay, s));
    return NP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// host-buffer ("drop-in") entry points: pack -> upload -> kernels -> download, synchronous
// ---------------------------------------------------------------------------------------------------------
static int pack_hmm_jobs(np_ctx* c, int n_jobs, const np_hmm_job* jobs, std::vector<np_hmm_job_dev>& dj,
                         std::vector<np_read_dev>& dr, std::vector<float>& ev, std::vector<uint16_t>& rk, int* model_out)
{
    int model = -1;
    dj.resize(n_jobs); dr.resize(n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        const np_hmm_job& q = jobs[j];
        if (!q.event_mean || !q.kmer_rank || q.n_kmers == 0 || q.n_kmers > NP_MAX_KMERS ||
            (q.stride != 1 && q.stride != -1) || q.e_start >= q.n_events_total || q.e_stop >= q.n_events_total) {
            c->err = "np_hmm_job: invalid field"; return q.n_kmers > NP_MAX_KMERS ? NP_ERR_UNSUPPORTED : NP_ERR_INVALID;
        }
        // the reference asserts rc <=> stride == -1 and walks events from e_start by stride (r9.inl:275,342)
        if ((q.stride == 1 && q.e_stop < q.e_start) || (q.stride == -1 && q.e_stop > q.e_start)) { c->err = "np_hmm_job: stride disagrees with e_start/e_stop"; return NP_ERR_INVALID; }
        if (model < 0) model = q.model; else if (model != q.model) { c->err = "one model per batch"; return NP_ERR_UNSUPPORTED; }
        const uint32_t lo = std::min(q.e_start, q.e_stop), hi = std::max(q.e_start, q.e_stop);
        if ((uint64_t)(hi - lo) + 1 > NP_FLANK_LEN) { c->err = "event window too long"; return NP_ERR_UNSUPPORTED; }
        np_read_dev& r = dr[j];
        memset(&r, 0, sizeof(r));
        r.scale = q.scale; r.shift = q.shift; r.var = q.var; r.log_var = log(q.var);
        r.event_off = (int64_t)ev.size() - (int64_t)lo;      // ev[event_off + event_idx] addresses the packed window
        r.n_events = q.n_events_total;
        if (c->host_constants) transitions_libm(q.events_per_base, q.indel_bias != 0.0 ? q.indel_bias : c->params.hmm_indel_bias_factor, r.trans);
        else np_transitions(q.events_per_base, q.indel_bias != 0.0 ? q.indel_bias : c->params.hmm_indel_bias_factor, r.trans);
        ev.insert(ev.end(), q.event_mean + lo, q.event_mean + hi + 1);
        np_hmm_job_dev& d = dj[j];
        d.rank_off = (int64_t)rk.size(); d.n_kmers = q.n_kmers; d.read = (uint32_t)j;
        d.e_start = q.e_start; d.e_stop = q.e_stop; d.stride = q.stride; d.flags = q.flags;
        rk.insert(rk.end(), q.kmer_rank, q.kmer_rank + q.n_kmers);
    }
    if (n_jobs > 0 && (model < 0 || model >= (int)c->models.size())) { c->err = "bad model id"; return NP_ERR_INVALID; }
    *model_out = model;
    return NP_OK;
}

// A SMALL batch through the host entry point (the per-call shim: one work item, or one round of the callers it combined) is bound by
// API calls, not by the device: four pageable uploads, a memset, the three binning kernels, the forward launch, a pageable read-back
// and two timer events cost ~230 us a round.  Here the packed arrays, the per-class order the binning kernels would have produced
// (np_job_bin, sorted on the host) and the zeroed tickets travel as ONE pinned blob (one upload), only the size classes that occur
// are launched, and the scores come back into the blob's pinned tail: two copies, the launches, one synchronisation.
#define NP_SMALL_BATCH 4096
static int score_small(np_ctx* c, hipStream_t s, int n_jobs, const std::vector<np_hmm_job_dev>& dj, const std::vector<np_read_dev>& dr,
                       const std::vector<float>& ev, const std::vector<uint16_t>& rk, int model, float* out_scores)
{
    if (model < 0 || model >= (int)c->models.size()) { c->err = "bad model id"; return NP_ERR_INVALID; }
    size_t off = 0;
    auto add = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t n = (size_t)n_jobs;
    const size_t o_cnt = add(16 * sizeof(uint32_t)), o_order = add((size_t)NP_NUM_CLASSES * n * sizeof(uint32_t)), o_jobs = add(n * sizeof(np_hmm_job_dev)),
                 o_reads = add(dr.size() * sizeof(np_read_dev)), o_ev = add(ev.size() * sizeof(float)), o_rk = add(rk.size() * sizeof(uint16_t));
    const size_t up = off;
    const size_t o_out = add(n * sizeof(float));
    if (off > c->small_h_cap) {
        if (c->small_h) (void)hipHostFree(c->small_h);
        c->small_h = nullptr; c->small_h_cap = 0;
        const size_t want = off + off / 2 + 65536;
        NP_HIP(c, hipHostMalloc(&c->small_h, want, hipHostMallocDefault));
        c->small_h_cap = want;
    }
    NP_HIP(c, c->small_d.reserve(c->small_h_cap));
    char* H = (char*)c->small_h; char* D = (char*)c->small_d.p;
    uint32_t* cnt = (uint32_t*)(H + o_cnt);
    memset(cnt, 0, 16 * sizeof(uint32_t));                      // [0..8): items per class, [8..16): the kernels' tickets
    uint32_t* order = (uint32_t*)(H + o_order);
    float* out_h = (float*)(H + o_out);
    std::vector<std::pair<int, uint32_t>> key(n);
    for (size_t j = 0; j < n; ++j) key[j] = std::make_pair(np_job_bin(dj[j], NP_FLANK_LEN), (uint32_t)j);
    std::sort(key.begin(), key.end());
    // A round of a few items (the per-call shim: up to one per calling thread) is one wave's worth of work per class, and the classes'
    // launches run one after the other: three classes, three times the latency.  A size class is a CAPACITY (lanes x blocks per lane
    // >= the item's k-mers; blocks per lane and lanes used follow the item at run time), so such a round goes to the largest class any
    // of its items needs, in ONE launch (same arithmetic per cell: tests/test_gpu_parity.py::test_host_scoring_...).
    int one_class = -1;
    if (n <= 64) for (size_t q = 0; q < n; ++q) if (key[q].first >= 0) one_class = std::max(one_class, key[q].first / (NP_CPL * NP_EBUCKETS));
    bool skipped = false;
    for (size_t q = 0; q < n; ++q) {
        if (key[q].first < 0) { skipped = true; continue; }
        const int cls = one_class >= 0 ? one_class : key[q].first / (NP_CPL * NP_EBUCKETS);
        order[(size_t)cls * n + cnt[cls]++] = key[q].second;
    }
    memcpy(H + o_jobs, dj.data(), n * sizeof(np_hmm_job_dev));
    memcpy(H + o_reads, dr.data(), dr.size() * sizeof(np_read_dev));
    memcpy(H + o_ev, ev.data(), ev.size() * sizeof(float));
    memcpy(H + o_rk, rk.data(), rk.size() * sizeof(uint16_t));
    // From here on the stream holds copies out of / into the shared pinned blob: whatever fails, the stream is drained before this call
    // returns -- the combiner retries the requests of a failed round one by one straight away, and the next call rewrites (or frees and
    // re-allocates) small_h.
    auto enqueue = [&]() -> int {
        NP_HIP(c, hipMemcpyAsync(D, H, up, hipMemcpyHostToDevice, s));
        {
            family_timer tm(c, 1, s);
            for (int cls = 0; cls < NP_NUM_CLASSES; ++cls) {
                if (!cnt[cls]) continue;
                np_hmm_args a{};
                a.jobs = (const np_hmm_job_dev*)(D + o_jobs); a.order = (const uint32_t*)(D + o_order) + (size_t)cls * n; a.n_class_jobs = (const uint32_t*)(D + o_cnt) + cls;
                a.reads = (const np_read_dev*)(D + o_reads); a.event_mean = (const float*)(D + o_ev); a.ranks = (const uint16_t*)(D + o_rk);
                a.model = c->models[model].d_states; a.logsum = c->d_logsum; a.flank = c->d_flank; a.counter = (uint32_t*)(D + o_cnt) + 8 + cls;
                a.out = (float*)(D + o_out); a.prio = c->hmm_prio;
                const int jobs_per_block = (np_hmm_block_threads(cls) / 64) * (64 / NP_CLASS_SEG[cls]);
                NP_HIP(c, np_launch_hmm_forward(cls, a, persistent_blocks(c, cnt[cls], jobs_per_block, c->hmm_blocks_per_cu), c->lse_oor, s));
            }
        }
        NP_HIP(c, hipMemcpyAsync(out_h, D + o_out, n * sizeof(float), hipMemcpyDeviceToHost, s));
        NP_HIP(c, hipStreamSynchronize(s));
        return NP_OK;
    };
    const int erc = enqueue();
    if (erc != NP_OK) { (void)hipStreamSynchronize(s); (void)hipGetLastError(); return erc; }
    memcpy(out_scores, out_h, n * sizeof(float));
    if (skipped) for (size_t q = 0; q < n && key[q].first < 0; ++q) out_scores[key[q].second] = __builtin_nanf("");      // (classify's rule for an item no class takes)
    drain_timing(c);
    return NP_OK;
}

int np_hmm_score_host(np_ctx* c, int n_jobs, const np_hmm_job* jobs, float* out_scores)
{
    if (!c || n_jobs < 0 || (n_jobs > 0 && (!jobs || !out_scores))) return NP_ERR_INVALID;
    if (n_jobs == 0) return NP_OK;
    // jobs under different pore models (profile_hmm_score_set scores each alphabet's sequence under that alphabet's
    // model, profile_hmm.cpp:44-52) are launched model by model: a kernel launch binds one model table
    {
        bool mixed = false;
        for (int j = 1; j < n_jobs && !mixed; ++j) mixed = jobs[j].model != jobs[0].model;
        if (mixed) {
            std::vector<int> models;
            for (int j = 0; j < n_jobs; ++j) if (std::find(models.begin(), models.end(), jobs[j].model) == models.end()) models.push_back(jobs[j].model);
            for (int m : models) {
                std::vector<np_hmm_job> sub; std::vector<int> idx;
                for (int j = 0; j < n_jobs; ++j) if (jobs[j].model == m) { sub.push_back(jobs[j]); idx.push_back(j); }
                std::vector<float> so(sub.size());
                const int rc = np_hmm_score_host(c, (int)sub.size(), sub.data(), so.data());
                if (rc != NP_OK) return rc;
                for (size_t q = 0; q < idx.size(); ++q) out_scores[idx[q]] = so[q];
            }
            return NP_OK;
        }
    }
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    std::vector<np_hmm_job_dev> dj; std::vector<np_read_dev> dr; std::vector<float> ev; std::vector<uint16_t> rk;
    int model = -1;
    int rc = pack_hmm_jobs(c, n_jobs, jobs, dj, dr, ev, rk, &model);
    if (rc != NP_OK) return rc;
    stream_scope scope = use_stream(c, nullptr); hipStream_t s = scope.s;
    if (n_jobs <= NP_SMALL_BATCH && c->small_batch_path) return score_small(c, s, n_jobs, dj, dr, ev, rk, model, out_scores);
    NP_HIP(c, c->b_jobs.reserve(dj.size() * sizeof(np_hmm_job_dev)));
    NP_HIP(c, c->b_reads.reserve(dr.size() * sizeof(np_read_dev)));
    NP_HIP(c, c->b_events.reserve(ev.size() * sizeof(float)));
    NP_HIP(c, c->b_ranks.reserve(rk.size() * sizeof(uint16_t)));
    NP_HIP(c, c->b_out.reserve((size_t)n_jobs * sizeof(float)));
    NP_HIP(c, hipMemcpyAsync(c->b_jobs.p, dj.data(), dj.size() * sizeof(np_hmm_job_dev), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_reads.p, dr.data(), dr.size() * sizeof(np_read_dev), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_events.p, ev.data(), ev.size() * sizeof(float), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_ranks.p, rk.data(), rk.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    unsigned class_mask = 0;
    for (int j = 0; j < n_jobs; ++j) {           // np_glue_kernels.hip:size_class
        const uint32_t n = jobs[j].n_kmers;
        class_mask |= 1u << (n <= 16 ? 0 : n <= 24 ? 1 : n <= 32 ? 2 : n <= 64 ? 3 : n <= 128 ? 4 : n <= 256 ? 5 : n <= 512 ? 6 : 7);
    }
    rc = run_hmm_forward(c, s, n_jobs, c->b_jobs.as<np_hmm_job_dev>(), c->b_reads.as<np_read_dev>(),
                         c->b_events.as<float>(), c->b_ranks.as<uint16_t>(), model, c->b_out.as<float>(), class_mask, /*use_layout=*/false);
    if (rc != NP_OK) return rc;
    NP_HIP(c, hipMemcpyAsync(out_scores, c->b_out.p, (size_t)n_jobs * sizeof(float), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipStreamSynchronize(s));
    drain_timing(c);
    return NP_OK;
}

// profile_hmm_score_set (src/hmm/nanopolish_profile_hmm.cpp:32-56): the per-sequence forward scores come from the
// device; the (tiny) combination  score = (+)_j (score_j - log n)  is done here in double with the same table.
int np_hmm_score_set_host(np_ctx* c, int n_sets, const int32_t* set_off, const np_hmm_job* jobs, float* out_scores)
{
    if (!c || n_sets < 0 || (n_sets > 0 && (!set_off || !jobs || !out_scores))) return NP_ERR_INVALID;
    if (n_sets == 0) return NP_OK;
    const int n_jobs = set_off[n_sets];
    std::vector<float> sc(n_jobs);
    const int rc = np_hmm_score_host(c, n_jobs, jobs, sc.data());
    if (rc != NP_OK) return rc;
    const float* tbl = c->h_logsum.data();
    auto add_logs = [&](double a, double b) -> double {      // add_logs -> p7_FLogsum, nanopolish_common.h:97-104
        const float fa = (float)a, fb = (float)b;
        const float mx = fa > fb ? fa : fb, mn = fa < fb ? fa : fb;
        return (mn == -INFINITY || (mx - mn) >= 15.7f) ? mx : mx + tbl[(int)((mx - mn) * 1000.f)];
    };
    for (int q = 0; q < n_sets; ++q) {
        const int b = set_off[q], n = set_off[q + 1] - b;
        if (n <= 0) { out_scores[q] = -INFINITY; continue; }
        const double pen = log((double)(size_t)n);
        double score = sc[b] - pen;
        for (int t = 1; t < n; ++t) { const double alt = sc[b + t] - pen; score = add_logs(score, alt); }
        out_scores[q] = (float)score;
    }
    return NP_OK;
}

int np_hmm_align_host(np_ctx* c, int n_jobs, const np_hmm_job* jobs, np_hmm_state* out_states, int64_t cap, int64_t* out_off)
{
    if (!c || n_jobs < 0 || (n_jobs > 0 && (!jobs || !out_states || !out_off))) return NP_ERR_INVALID;
    if (n_jobs == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    std::vector<np_hmm_job_dev> dj; std::vector<np_read_dev> dr; std::vector<float> ev; std::vector<uint16_t> rk;
    int model = -1;
    int rc = pack_hmm_jobs(c, n_jobs, jobs, dj, dr, ev, rk, &model);
    if (rc != NP_OK) return rc;
    std::vector<int64_t> cell_off(n_jobs + 1, 0), state_off(n_jobs + 1, 0);
    for (int j = 0; j < n_jobs; ++j) {
        const int64_t e = (int64_t)(dj[j].e_stop > dj[j].e_start ? dj[j].e_stop - dj[j].e_start : dj[j].e_start - dj[j].e_stop) + 1;
        cell_off[j + 1] = cell_off[j] + e * 3 * (int64_t)dj[j].n_kmers;
        state_off[j + 1] = state_off[j] + e + (int64_t)dj[j].n_kmers + 1;     // path length bound: e rows + n silent K hops
    }
    stream_scope scope = use_stream(c, nullptr); hipStream_t s = scope.s;
    NP_HIP(c, c->b_jobs.reserve(dj.size() * sizeof(np_hmm_job_dev)));
    NP_HIP(c, c->b_reads.reserve(dr.size() * sizeof(np_read_dev)));
    NP_HIP(c, c->b_events.reserve(ev.size() * sizeof(float)));
    NP_HIP(c, c->b_ranks.reserve(rk.size() * sizeof(uint16_t)));
    NP_HIP(c, c->b_vm.reserve((size_t)cell_off[n_jobs] * sizeof(float)));
    NP_HIP(c, c->b_bp.reserve((size_t)cell_off[n_jobs]));
    NP_HIP(c, c->b_cell_off.reserve(cell_off.size() * sizeof(int64_t)));
    NP_HIP(c, c->b_state_off.reserve(state_off.size() * sizeof(int64_t)));
    NP_HIP(c, c->b_states.reserve((size_t)state_off[n_jobs] * sizeof(np_hmm_state)));
    NP_HIP(c, c->b_n_states.reserve((size_t)n_jobs * sizeof(int32_t)));
    NP_HIP(c, c->order.reserve((size_t)NP_NUM_CLASSES * (size_t)n_jobs * sizeof(uint32_t)));
    NP_HIP(c, hipMemcpyAsync(c->b_jobs.p, dj.data(), dj.size() * sizeof(np_hmm_job_dev), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_reads.p, dr.data(), dr.size() * sizeof(np_read_dev), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_events.p, ev.data(), ev.size() * sizeof(float), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_ranks.p, rk.data(), rk.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_cell_off.p, cell_off.data(), cell_off.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_state_off.p, state_off.data(), state_off.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemsetAsync(c->d_counters, 0, 16 * sizeof(uint32_t), s));
    {
        family_timer tm(c, 3, s);
        NP_HIP(c, np_launch_classify(c->b_jobs.as<np_hmm_job_dev>(), n_jobs, c->d_counters, c->order.as<uint32_t>(), nullptr, NP_FLANK_LEN, c->d_counters + 1024, np_slots{nullptr, nullptr, 0}, s));
        np_hmm_args a{};
        a.jobs = c->b_jobs.as<np_hmm_job_dev>(); a.reads = c->b_reads.as<np_read_dev>(); a.event_mean = c->b_events.as<float>();
        a.ranks = c->b_ranks.as<uint16_t>(); a.model = c->models[model].d_states; a.logsum = c->d_logsum; a.flank = c->d_flank;
        a.vm = c->b_vm.as<float>(); a.bp = c->b_bp.as<uint8_t>(); a.cell_off = c->b_cell_off.as<int64_t>();
        a.states = c->b_states.as<np_hmm_state>(); a.state_off = c->b_state_off.as<int64_t>(); a.n_states = c->b_n_states.as<int32_t>();
        for (int cls = 0; cls < NP_NUM_CLASSES; ++cls) {
            a.order = c->order.as<uint32_t>() + (size_t)cls * (size_t)n_jobs; a.n_class_jobs = c->d_counters + cls;
            a.counter = c->d_counters + 8 + cls;
            const int jobs_per_block = (np_vit_block_threads() / 64) * (64 / NP_CLASS_SEG[cls]);
            NP_HIP(c, np_launch_hmm_viterbi(cls, a, persistent_blocks(c, n_jobs, jobs_per_block, c->hmm_blocks_per_cu), s));
        }
        NP_HIP(c, np_launch_hmm_backtrack(a, n_jobs, s));
    }
    std::vector<int32_t> ns(n_jobs);
    std::vector<np_hmm_state> st((size_t)state_off[n_jobs]);
    NP_HIP(c, hipMemcpyAsync(ns.data(), c->b_n_states.p, ns.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipMemcpyAsync(st.data(), c->b_states.p, st.size() * sizeof(np_hmm_state), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipStreamSynchronize(s));
    drain_timing(c);
    int64_t w = 0;
    for (int j = 0; j < n_jobs; ++j) {
        out_off[j] = w;
        if (w + ns[j] > cap) { c->err = "np_hmm_align_host: output capacity too small"; return NP_ERR_NOMEM; }
        memcpy(out_states + w, st.data() + state_off[j], (size_t)ns[j] * sizeof(np_hmm_state));
        w += ns[j];
    }
    out_off[n_jobs] = w;
    return NP_OK;
}

int np_event_align_host(np_ctx* c, int n_jobs, const np_align_job* jobs, np_pair* out_pairs, int64_t cap, int64_t* out_off)
{
    if (!c || n_jobs < 0 || (n_jobs > 0 && (!jobs || !out_pairs || !out_off))) return NP_ERR_INVALID;
    if (n_jobs == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    std::vector<np_read_dev> dr(n_jobs); std::vector<float> ev; std::vector<uint16_t> rk;
    std::vector<int64_t> pair_off(n_jobs + 1, 0);
    int model = -1; int64_t max_bands = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const np_align_job& q = jobs[j];
        if (!q.event_mean || !q.kmer_rank || q.n_events == 0 || q.n_kmers == 0) { c->err = "np_align_job: invalid field"; return NP_ERR_INVALID; }
        if (model < 0) model = q.model; else if (model != q.model) { c->err = "one model per batch"; return NP_ERR_UNSUPPORTED; }
        np_fill_read_host(&dr[j], q.shift, q.scale, q.var, (int64_t)ev.size(), q.n_events, (int64_t)rk.size(), q.n_kmers);
        ev.insert(ev.end(), q.event_mean, q.event_mean + q.n_events);
        rk.insert(rk.end(), q.kmer_rank, q.kmer_rank + q.n_kmers);
        const int64_t nb = (int64_t)q.n_events + q.n_kmers + 2;
        pair_off[j + 1] = pair_off[j] + nb;
        max_bands = std::max(max_bands, nb);
    }
    stream_scope scope = use_stream(c, nullptr); hipStream_t s = scope.s;
    NP_HIP(c, c->b_reads.reserve(dr.size() * sizeof(np_read_dev)));
    NP_HIP(c, c->b_events.reserve(ev.size() * sizeof(float)));
    NP_HIP(c, c->b_ranks.reserve(rk.size() * sizeof(uint16_t)));
    NP_HIP(c, c->b_pair_off.reserve(pair_off.size() * sizeof(int64_t)));
    NP_HIP(c, c->b_pairs.reserve((size_t)pair_off[n_jobs] * sizeof(np_pair)));
    NP_HIP(c, c->b_pair_begin.reserve((size_t)n_jobs * sizeof(int32_t)));
    NP_HIP(c, c->b_n_pairs.reserve((size_t)n_jobs * sizeof(int32_t)));
    NP_HIP(c, hipMemcpyAsync(c->b_reads.p, dr.data(), dr.size() * sizeof(np_read_dev), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_events.p, ev.data(), ev.size() * sizeof(float), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_ranks.p, rk.data(), rk.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_pair_off.p, pair_off.data(), pair_off.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    int rc = run_event_align(c, s, n_jobs, c->b_reads.as<np_read_dev>(), c->b_events.as<float>(), c->b_ranks.as<uint16_t>(),
                             model, max_bands, c->b_pair_off.as<int64_t>(), c->b_pairs.as<np_pair>(),
                             c->b_pair_begin.as<int32_t>(), c->b_n_pairs.as<int32_t>());
    if (rc != NP_OK) return rc;
    std::vector<np_pair> hp((size_t)pair_off[n_jobs]);
    std::vector<int32_t> hb(n_jobs), hn(n_jobs);
    NP_HIP(c, hipMemcpyAsync(hp.data(), c->b_pairs.p, hp.size() * sizeof(np_pair), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipMemcpyAsync(hb.data(), c->b_pair_begin.p, hb.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipMemcpyAsync(hn.data(), c->b_n_pairs.p, hn.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipStreamSynchronize(s));
    drain_timing(c);
    int64_t w = 0;
    for (int j = 0; j < n_jobs; ++j) {
        out_off[j] = w;
        if (w + hn[j] > cap) { c->err = "np_event_align_host: output capacity too small"; return NP_ERR_NOMEM; }
        memcpy(out_pairs + w, hp.data() + pair_off[j] + hb[j], (size_t)hn[j] * sizeof(np_pair));
        w += hn[j];
    }
    out_off[n_jobs] = w;
    return NP_OK;
}

// ---- f3: work-item generation on the device ---------------------------------------------------------------------------
int np_cm_build_jobs_identity_dev(np_ctx* c, void* stream, int n_reads, const char* ref_seq, const int64_t* seq_off, const uint8_t* read_rc,
                                  int alphabet, uint32_t k, int min_separation, int min_flank, const int64_t* group_off,
                                  int64_t total_group_slots, const int64_t* rank_off, np_hmm_job_dev* jobs, int32_t* kpos,
                                  uint16_t* job_ranks, int32_t* first_site, int32_t* last_site, int32_t* n_motif, int32_t* n_groups)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!ref_seq || !seq_off || !read_rc || !group_off || !rank_off || !jobs || !kpos || !job_ranks ||
                                             !first_site || !last_site || !n_motif || !n_groups))) return NP_ERR_INVALID;
    if (alphabet < 1 || alphabet > 4) { c->err = "np_cm_build_jobs_identity_dev: cpg, gpc, dam or dcm"; return NP_ERR_UNSUPPORTED; }
    if (k < 1 || k > 6) { c->err = "np_cm_build_jobs_identity_dev: k must be 1..6 (uint16 ranks over 5 letters)"; return NP_ERR_UNSUPPORTED; }
    if (n_reads == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    NP_HIP(c, c->cm_group_rank_off.reserve((size_t)total_group_slots * sizeof(int64_t)));
    {
        family_timer tm(c, 2, s);
        NP_HIP(c, np_launch_cm_build_jobs(n_reads, ref_seq, seq_off, read_rc, alphabet, (int)k, min_separation, min_flank, group_off, rank_off, jobs,
                                          kpos, job_ranks, first_site, last_site, n_motif, c->cm_group_rank_off.as<int64_t>(), n_groups,
                                          !(c->lay.n_reads == n_reads && c->lay.group_off == group_off && c->lay.n_groups == n_groups), s));
    }
    return NP_OK;
}

int np_cm_build_jobs_cigar_dev(np_ctx* c, void* stream, int n_reads, const char* genome, const int64_t* ref_begin, const int32_t* ref_len,
                               const uint32_t* cigar, const int64_t* cigar_off, int64_t total_cigar_ops, const int32_t* read_len,
                               const uint8_t* read_rc, int alphabet, uint32_t k, int min_separation, int min_flank, const int64_t* group_off,
                               int64_t total_group_slots, const int64_t* rank_off, np_hmm_job_dev* jobs, int32_t* kpos, uint16_t* job_ranks,
                               int32_t* first_site, int32_t* last_site, int32_t* n_motif, int32_t* n_groups, int32_t* deg_kpos)
{
    if (!c || n_reads < 0 || total_cigar_ops < 0 ||
        (n_reads > 0 && (!genome || !ref_begin || !ref_len || !cigar || !cigar_off || !read_len || !read_rc || !group_off || !rank_off || !jobs ||
                         !kpos || !job_ranks || !first_site || !last_site || !n_motif || !n_groups || !deg_kpos))) return NP_ERR_INVALID;
    if (alphabet < 1 || alphabet > 4) { c->err = "np_cm_build_jobs_cigar_dev: cpg, gpc, dam or dcm"; return NP_ERR_UNSUPPORTED; }
    if (k < 1 || k > 6) { c->err = "np_cm_build_jobs_cigar_dev: k must be 1..6 (uint16 ranks over 5 letters)"; return NP_ERR_UNSUPPORTED; }
    if (n_reads == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    const size_t n_idx = (size_t)total_cigar_ops + (size_t)n_reads;
    NP_HIP(c, c->cm_group_rank_off.reserve((size_t)total_group_slots * sizeof(int64_t)));
    NP_HIP(c, c->cm_cigar_scratch.reserve(2 * n_idx * sizeof(int32_t) + (size_t)n_reads * 16 + 2 * (size_t)total_group_slots * sizeof(int32_t)));
    int32_t* op_ref = c->cm_cigar_scratch.as<int32_t>();
    int32_t* op_read = op_ref + n_idx;
    int32_t* cig_reads = op_read + n_idx;                      // 16 B per read
    int32_t* group_kpos = cig_reads + 4 * (size_t)n_reads;
    {
        family_timer tm(c, 2, s);
        NP_HIP(c, np_launch_cm_build_jobs_cigar(n_reads, genome, ref_begin, ref_len, cigar, cigar_off, read_len, read_rc, alphabet, (int)k,
                                                min_separation, min_flank, group_off, rank_off, jobs, kpos, job_ranks, first_site, last_site, n_motif,
                                                c->cm_group_rank_off.as<int64_t>(), n_groups, deg_kpos, op_ref, op_read, cig_reads, group_kpos,
                                                !(c->lay.n_reads == n_reads && c->lay.group_off == group_off && c->lay.n_groups == n_groups), s));
    }
    return NP_OK;
}

int np_set_job_layout(np_ctx* c, int n_reads, const int64_t* group_off, const int32_t* n_groups, int64_t total_slots)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!group_off || !n_groups || total_slots < 0))) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    if (n_reads == 0) { c->lay = np_slots{nullptr, nullptr, 0}; c->lay_total = 0; }
    else { c->lay = np_slots{group_off, n_groups, n_reads}; c->lay_total = total_slots; }
    return NP_OK;
}

int np_cm_discard_degenerate_dev(np_ctx* c, void* stream, const np_read_dev* reads, const int32_t* map_start, const int32_t* deg_kpos,
                                 int64_t n_jobs, np_hmm_job_dev* jobs)
{
    if (!c || n_jobs < 0 || (n_jobs > 0 && (!reads || !map_start || !deg_kpos || !jobs))) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    family_timer tm(c, 2, s);
    np_slots lay;
    { const int rc = layout_for(c, n_jobs, &lay); if (rc != NP_OK) return rc; }
    NP_HIP(c, np_launch_discard_degenerate(n_jobs, jobs, reads, map_start, deg_kpos, lay, s));
    return NP_OK;
}

// ---- eventalign segment chain -----------------------------------------------------------------------------------------------
int np_eventalign_dev(np_ctx* c, void* stream, int n_reads, const np_read_dev* reads, const float* event_mean, const int32_t* map_start,
                      const int32_t* n_pairs, const double* events_per_base, const int32_t* calibrated, int model, const char* genome,
                      const int64_t* ref_begin, const int32_t* ref_len, const uint32_t* cigar, const int64_t* cigar_off,
                      int64_t total_cigar_ops, const int32_t* read_len, const uint8_t* read_rc, uint32_t k, const int64_t* out_off,
                      int32_t* out_ref, int32_t* out_event, uint8_t* out_state, int32_t* n_out, int32_t* status, int32_t* n_calls)
{
    if (!c || n_reads < 0 || total_cigar_ops < 0 ||
        (n_reads > 0 && (!reads || !event_mean || !map_start || !n_pairs || !events_per_base || !genome || !ref_begin || !ref_len || !cigar ||
                         !cigar_off || !read_len || !read_rc || !out_off || !out_ref || !out_event || !out_state || !n_out || !status || !n_calls)))
        return NP_ERR_INVALID;
    if (model < 0 || model >= (int)c->models.size()) { c->err = "bad model id"; return NP_ERR_INVALID; }
    // the base model of a DNA read (4096 states, k = 6) or of a direct-RNA read (u_to_t_rna: 1024 states, k = 5); a segment holds up to
    // 102 - k k-mers (eventalign.cpp:695-735)
    if (!((c->models[model].n_states == 4096 && k == 6) || (c->models[model].n_states == 1024 && k == 5))) {
        c->err = "np_eventalign_dev: a four-letter base model with k = 6 (DNA) or k = 5 (direct RNA)"; return NP_ERR_UNSUPPORTED;
    }
    if (n_reads == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    const size_t n_idx = (size_t)total_cigar_ops + (size_t)n_reads;
    NP_HIP(c, c->cm_cigar_scratch.reserve(2 * n_idx * sizeof(int32_t) + (size_t)n_reads * 16));
    int32_t* op_ref = c->cm_cigar_scratch.as<int32_t>();
    int32_t* op_read = op_ref + n_idx;
    int32_t* cig_reads = op_read + n_idx;
    const int rows_cap = c->ea_rows_cap;
    const int variant = k == 5 ? 4 : (c->ea_waves_per_cu > 16 ? 3 : 2);        // (k = 5: four blocks per lane; else the register budget of 5 or 4 waves per SIMD)
    const int waves_per_cu = c->ea_waves_per_cu;
    const int nb = persistent_blocks(c, (n_reads + 1) / 2, 1, waves_per_cu);
    const size_t bp_stride = ((size_t)rows_cap + 64) * (size_t)np_eventalign_line_bytes(variant), path_stride = 4 * ((size_t)rows_cap + 256);     // one line per sweep step: e + 63 at most; two path lists (one per half-wave), a 64-bit word per burst of walk steps, e + n bursts at most
    NP_HIP(c, c->ea_bp.reserve((size_t)nb * bp_stride));
    NP_HIP(c, c->ea_path.reserve((size_t)nb * path_stride * sizeof(uint32_t)));
    family_timer tm(c, 6, s);
    NP_HIP(c, np_launch_cigar_index(n_reads, cigar, cigar_off, read_len, (int)k, op_ref, op_read, cig_reads, s));
    NP_HIP(c, hipMemsetAsync(c->d_counters + 17, 0, sizeof(uint32_t), s));
    NP_HIP(c, hipMemsetAsync(c->d_counters + 40, 0, 6 * sizeof(unsigned long long), s));
    np_ea_args a{};
    a.stats = (unsigned long long*)(c->d_counters + 40);
    a.walk_prio = c->ea_walk_prio;
    a.n_reads = n_reads; a.reads = reads; a.event_mean = event_mean; a.map_start = map_start; a.n_pairs = n_pairs;
    a.events_per_base = events_per_base; a.calibrated = calibrated;
    a.model = c->models[model].d_states; a.flank = c->d_flank; a.genome = genome; a.ref_begin = ref_begin; a.ref_len = ref_len;
    a.cigar = cigar; a.cigar_off = cigar_off; a.op_ref = op_ref; a.op_read = op_read; a.cig_reads = cig_reads;
    a.read_len = read_len; a.read_rc = read_rc; a.k = (int)k;
    a.bp = c->ea_bp.as<uint8_t>(); a.bp_stride = bp_stride; a.rows_cap = rows_cap; a.max_kmers = np_eventalign_max_kmers(variant);
    a.path = c->ea_path.as<uint32_t>(); a.path_stride = path_stride;
    a.out_off = out_off; a.out_ref = out_ref; a.out_event = out_event; a.out_state = out_state; a.n_out = n_out; a.status = status;
    a.n_calls = n_calls; a.counter = c->d_counters + 17;
    NP_HIP(c, c->ea_args.reserve(sizeof(np_ea_args)));
    NP_HIP(c, hipMemcpyAsync(c->ea_args.p, &a, sizeof(np_ea_args), hipMemcpyHostToDevice, s));      // (pageable source: staged before the call returns)
    NP_HIP(c, np_launch_eventalign_chain(a, c->ea_args.as<np_ea_args>(), nb, variant, s));
    return NP_OK;
}

int64_t np_get_stat(np_ctx* c, const char* name)
{
    if (!c || !name) return -1;
    std::lock_guard<std::mutex> g(c->lock);
    const std::string k(name);
    if (k == "align_blocks") return c->last_align_blocks;
    if (k == "align_scratch_bytes") return c->last_align_scratch;
    if (k == "align_blocks_max") return (int64_t)c->n_cu * c->align_blocks_per_cu;
    if (k == "lse_oor") return c->lse_oor ? 1 : 0;
    if (k == "lse_probe_ok") return c->lse_probe_ok ? 1 : 0;
    if (k == "host_constants") return c->host_constants ? 1 : 0;
    if (k == "n_cu") return c->n_cu;
    if (k == "ed_serial_reads" || k == "ed_refused_reads") {     // of the most recent np_detect_events_* call (waits for it): reads whose
        // prefix sums were accumulated serially (exactness bound not provable), resp. refused (NP_ED_INEXACT: non-finite samples)
        if (c->ed_last_reads <= 0 || !c->ed_status.p) return 0;
        std::vector<int32_t> st((size_t)c->ed_last_reads);
        if (hipSetDevice(c->device) != hipSuccess) return -1;
        if (c->tail_recorded && hipEventSynchronize(c->switch_ev) != hipSuccess) return -1;
        if (hipMemcpy(st.data(), c->ed_status.p, st.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        int64_t n = 0;
        for (int32_t v : st) n += k == "ed_serial_reads" ? v == 1 : v == NP_ED_INEXACT;
        return n;
    }
    if (k.rfind("ea_", 0) == 0) {        // statistics of the most recent np_eventalign_dev call (waits for it)
        unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
        if (hipSetDevice(c->device) != hipSuccess) return -1;
        if (c->tail_recorded && hipEventSynchronize(c->switch_ev) != hipSuccess) return -1;
        if (hipMemcpy(h, c->d_counters + 40, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        static const char* names[6] = {"ea_lattice_cells", "ea_lattice_rows", "ea_lattice_kmers", "ea_cycles_geometry", "ea_cycles_fill", "ea_cycles_backtrack"};
        for (int i = 0; i < 6; ++i) if (k == names[i]) return (int64_t)h[i];
        return -1;
    }
    return -1;
}

// tuning / test knobs
int np_set_option(np_ctx* c, const char* name, int64_t value)
{
    if (!c || !name) return NP_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->lock);
    const std::string k(name);
    if (k == "align_blocks_per_cu") c->align_blocks_per_cu = (int)std::max<int64_t>(1, value);
    else if (k == "hmm_prio") c->hmm_prio = (int)std::min<int64_t>(2, std::max<int64_t>(0, value));
    else if (k == "hmm_blocks_per_cu") c->hmm_blocks_per_cu = (int)std::max<int64_t>(1, value);
    else if (k == "align_lpt") c->align_lpt = value != 0;
    else if (k == "recal_shape") c->recal_shape = (int)std::min<int64_t>(3, std::max<int64_t>(0, value));
    else if (k == "stream_switch_wait") c->stream_switch_wait = value != 0;
    else if (k == "small_batch_path") c->small_batch_path = value != 0;
    else if (k == "ed_warmup") c->ed_warmup = (int)value;
    else if (k == "ed_ratio_exact") c->ed_ratio_exact = value != 0;
    else if (k == "ea_rows_cap") c->ea_rows_cap = (int)std::min<int64_t>(65535, std::max<int64_t>(16, value));
    else if (k == "lse_oor") {                                   // tests: both log-sum lookups must give the same scores
        if (value != 0 && !c->lse_probe_ok) { c->err = "lse_oor: the hardware probe of this context failed; the clamp-free log-sum is not available"; return NP_ERR_UNSUPPORTED; }
        c->lse_oor = value != 0;
    }
    else if (k == "ea_kernel") { if (value != 2) { c->err = "ea_kernel: the one-read chain kernel was removed in round 4"; return NP_ERR_INVALID; } }
    else if (k == "ea_waves_per_cu") c->ea_waves_per_cu = (int)std::max<int64_t>(1, value);
    else { c->err = "np_set_option: unknown option " + k; return NP_ERR_INVALID; }
    return NP_OK;
}

// ---- f2: event detection + MoM scalings ------------------------------------------------------------------------------
void np_event_detection_params(np_detector_param* p, int rna)
{
    if (!p) return;
    if (!rna) { p->window_length1 = 3; p->window_length2 = 6; p->threshold1 = 1.4f; p->threshold2 = 9.0f; p->peak_height = 0.2f; }   // event_detection.h:15-21
    else { p->window_length1 = 7; p->window_length2 = 14; p->threshold1 = 2.5f; p->threshold2 = 9.0f; p->peak_height = 1.0f; }      // :23-29
}

static int detect_events_locked(np_ctx* c, hipStream_t s, int n_reads, const float* raw, const int64_t* raw_off, int64_t max_samples,
                                const np_detector_param* params, float* tstat, int64_t total_samples_hint, const int64_t* event_off,
                                int64_t max_events, uint32_t* event_start, float* event_length, float* event_mean, float* event_stdv,
                                int32_t* n_events, const int32_t* verdict = nullptr)
{
    np_detector_param p;
    if (params) p = *params; else np_event_detection_params(&p, 0);
    if (p.window_length1 > 16 || p.window_length2 > 16) { c->err = "np_detect_events: window length > 16"; return NP_ERR_UNSUPPORTED; }
    if (tstat && ((uintptr_t)tstat & 63u)) { c->err = "np_detect_events: tstat scratch must be 64-byte aligned"; return NP_ERR_INVALID; }
    NP_HIP(c, c->ed_status.reserve((size_t)n_reads * sizeof(int32_t)));
    c->ed_last_reads = n_reads;
    if (!tstat) {
        NP_HIP(c, c->ed_tstat.reserve((size_t)total_samples_hint * sizeof(float2) + 64));
        tstat = c->ed_tstat.as<float>();
    }
    // the caller's verdicts (np_adc_to_pa_checked_dev on exactly these samples) replace the detector's own pass over them: the status
    // words stay the context's (np_get_stat "ed_serial_reads" reads them), n_reads x 4 bytes copied on the stream
    const bool checked = verdict != nullptr && n_reads > 0;
    if (checked) NP_HIP(c, hipMemcpyAsync(c->ed_status.p, verdict, (size_t)n_reads * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    family_timer tm(c, 4, s);
    NP_HIP(c, np_launch_detect_events(n_reads, raw, raw_off, max_samples, p, (float2*)tstat, c->ed_status.as<int32_t>(), event_off,
                                      max_events, event_start, event_length, event_mean, event_stdv, n_events, c->ed_warmup, checked, s, c->ed_ratio_exact ? 1 : 0));
    return NP_OK;
}

int np_detect_events_dev(np_ctx* c, void* stream, int n_reads, const float* raw, const int64_t* raw_off, int64_t max_samples,
                         const np_detector_param* params, float* tstat, const int64_t* event_off, int64_t max_events,
                         uint32_t* event_start, float* event_length, float* event_mean, float* event_stdv, int32_t* n_events)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!raw || !raw_off || !tstat || !event_off || !event_start || !event_length || !event_mean ||
                                             !event_stdv || !n_events))) return NP_ERR_INVALID;
    if (n_reads == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    return detect_events_locked(c, use_stream(c, stream), n_reads, raw, raw_off, max_samples, params, tstat, 0, event_off, max_events,
                                event_start, event_length, event_mean, event_stdv, n_events);
}

int np_detect_events_checked_dev(np_ctx* c, void* stream, int n_reads, const float* raw, const int64_t* raw_off, int64_t max_samples,
                                 const np_detector_param* params, float* tstat, const int64_t* event_off, int64_t max_events,
                                 uint32_t* event_start, float* event_length, float* event_mean, float* event_stdv, int32_t* n_events,
                                 const int32_t* verdict)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!raw || !raw_off || !tstat || !event_off || !event_start || !event_length || !event_mean ||
                                             !event_stdv || !n_events))) return NP_ERR_INVALID;
    if (n_reads == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    return detect_events_locked(c, use_stream(c, stream), n_reads, raw, raw_off, max_samples, params, tstat, 0, event_off, max_events,
                                event_start, event_length, event_mean, event_stdv, n_events, verdict);
}

int np_detect_events_adc_dev(np_ctx* c, void* stream, int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples,
                             const float* offset, const float* raw_unit, float* raw_pa, const np_detector_param* params, float* tstat,
                             const int64_t* event_off, int64_t max_events, uint32_t* event_start, float* event_length, float* event_mean,
                             float* event_stdv, int32_t* n_events)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!adc || !raw_off || !offset || !raw_unit || !raw_pa || !tstat || !event_off || !event_start || !event_length ||
                                             !event_mean || !event_stdv || !n_events))) return NP_ERR_INVALID;
    if (n_reads == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    if (((uintptr_t)adc & 3u) || ((uintptr_t)tstat & 63u)) { c->err = "np_detect_events_adc_dev: adc must be 4-byte aligned, tstat 64-byte aligned"; return NP_ERR_INVALID; }
    NP_HIP(c, hipSetDevice(c->device));
    np_detector_param p;
    if (params) p = *params; else np_event_detection_params(&p, 0);
    if (p.window_length1 > 16 || p.window_length2 > 16) { c->err = "np_detect_events: window length > 16"; return NP_ERR_UNSUPPORTED; }
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    NP_HIP(c, c->ed_status.reserve((size_t)n_reads * sizeof(int32_t)));
    c->ed_last_reads = n_reads;
    family_timer tm(c, 4, s);
    NP_HIP(c, np_launch_detect_events_adc(n_reads, adc, raw_off, max_samples, offset, raw_unit, raw_pa, p, (float2*)tstat, c->ed_status.as<int32_t>(),
                                          event_off, max_events, event_start, event_length, event_mean, event_stdv, n_events, c->ed_warmup, s, c->ed_ratio_exact ? 1 : 0));
    return NP_OK;
}

int np_detect_events_host(np_ctx* c, int n_reads, const float* const* raw, const uint32_t* n_samples, const np_detector_param* params,
                          uint32_t* out_start, float* out_length, float* out_mean, float* out_stdv, int64_t cap, int64_t* out_off)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!raw || !n_samples || !out_start || !out_length || !out_mean || !out_stdv || !out_off)))
        return NP_ERR_INVALID;
    if (n_reads == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    std::vector<int64_t> raw_off(n_reads + 1, 0), ev_off(n_reads + 1, 0);
    int64_t max_samples = 0, max_events = 0;
    for (int r = 0; r < n_reads; ++r) {
        if (!raw[r] && n_samples[r]) { c->err = "np_detect_events_host: null raw table"; return NP_ERR_INVALID; }
        raw_off[r + 1] = raw_off[r] + n_samples[r];
        const int64_t ecap = (int64_t)n_samples[r] / 2 + 2;
        ev_off[r + 1] = ev_off[r] + ecap;
        max_samples = std::max<int64_t>(max_samples, n_samples[r]); max_events = std::max(max_events, ecap);
    }
    stream_scope scope = use_stream(c, nullptr); hipStream_t s = scope.s;
    const size_t ns = (size_t)raw_off[n_reads], ne = (size_t)ev_off[n_reads];
    NP_HIP(c, c->b_raw.reserve(ns * sizeof(float) + 16));
    NP_HIP(c, c->b_raw_off.reserve(raw_off.size() * sizeof(int64_t)));
    NP_HIP(c, c->b_ev_off.reserve(ev_off.size() * sizeof(int64_t)));
    NP_HIP(c, c->b_ev_start.reserve(ne * sizeof(uint32_t))); NP_HIP(c, c->b_ev_len.reserve(ne * sizeof(float)));
    NP_HIP(c, c->b_ev_mean.reserve(ne * sizeof(float))); NP_HIP(c, c->b_ev_stdv.reserve(ne * sizeof(float)));
    NP_HIP(c, c->b_n_events.reserve((size_t)n_reads * sizeof(int32_t)));
    for (int r = 0; r < n_reads; ++r)
        if (n_samples[r])
            NP_HIP(c, hipMemcpyAsync(c->b_raw.as<float>() + raw_off[r], raw[r], (size_t)n_samples[r] * sizeof(float), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_raw_off.p, raw_off.data(), raw_off.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    NP_HIP(c, hipMemcpyAsync(c->b_ev_off.p, ev_off.data(), ev_off.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    int rc = detect_events_locked(c, s, n_reads, c->b_raw.as<float>(), c->b_raw_off.as<int64_t>(), max_samples, params, nullptr, (int64_t)ns,
                                  c->b_ev_off.as<int64_t>(), max_events, c->b_ev_start.as<uint32_t>(), c->b_ev_len.as<float>(),
                                  c->b_ev_mean.as<float>(), c->b_ev_stdv.as<float>(), c->b_n_events.as<int32_t>());
    if (rc != NP_OK) return rc;
    std::vector<int32_t> hn(n_reads);
    std::vector<uint32_t> hs(ne); std::vector<float> hl(ne), hm(ne), hd(ne);
    NP_HIP(c, hipMemcpyAsync(hn.data(), c->b_n_events.p, hn.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipMemcpyAsync(hs.data(), c->b_ev_start.p, ne * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipMemcpyAsync(hl.data(), c->b_ev_len.p, ne * sizeof(float), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipMemcpyAsync(hm.data(), c->b_ev_mean.p, ne * sizeof(float), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipMemcpyAsync(hd.data(), c->b_ev_stdv.p, ne * sizeof(float), hipMemcpyDeviceToHost, s));
    NP_HIP(c, hipStreamSynchronize(s));
    drain_timing(c);
    int64_t w = 0;
    for (int r = 0; r < n_reads; ++r) {
        out_off[r] = w;
        if (hn[r] == NP_ED_INEXACT) { c->err = "np_detect_events_host: a read holds a non-finite sample (NP_ED_INEXACT)"; return NP_ERR_UNSUPPORTED; }
        if (hn[r] < 0) { c->err = "np_detect_events_host: event capacity exceeded"; return NP_ERR_NOMEM; }
        if (w + hn[r] > cap) { c->err = "np_detect_events_host: output capacity too small"; return NP_ERR_NOMEM; }
        const size_t o = (size_t)ev_off[r];
        memcpy(out_start + w, hs.data() + o, (size_t)hn[r] * sizeof(uint32_t)); memcpy(out_length + w, hl.data() + o, (size_t)hn[r] * sizeof(float));
        memcpy(out_mean + w, hm.data() + o, (size_t)hn[r] * sizeof(float)); memcpy(out_stdv + w, hd.data() + o, (size_t)hn[r] * sizeof(float));
        w += hn[r];
    }
    out_off[n_reads] = w;
    return NP_OK;
}

int np_mom_fill_dev(np_ctx* c, void* stream, int n_reads, np_read_dev* reads, np_read_dev* reads_b, const float* event_mean,
                    const int32_t* n_events, const uint16_t* kmer_rank, int model)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!reads || !event_mean || !n_events || !kmer_rank))) return NP_ERR_INVALID;
    if (model < 0 || model >= (int)c->models.size()) { c->err = "bad model id"; return NP_ERR_INVALID; }
    if (n_reads == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    family_timer tm(c, 5, s);
    NP_HIP(c, np_launch_mom_fill(n_reads, reads, reads_b, event_mean, n_events, kmer_rank, c->models[model].d_states, c->models[model].n_states, s));
    if (c->host_constants) return host_constants_fix(c, s, n_reads, reads, nullptr, 1);
    return NP_OK;
}

// Direct-RNA reads: the detected events of every read reversed in place (squiggle_read.cpp:260-263), AFTER np_mom_fill_dev (the MoM
// sums run over the events in detection order) and before np_event_align_dev.  Any of the four arrays may be null.
int np_reverse_events_dev(np_ctx* c, void* stream, int n_reads, const int64_t* event_off, const int32_t* n_events, uint32_t* event_start,
                          float* event_length, float* event_mean, float* event_stdv)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!event_off || !n_events))) return NP_ERR_INVALID;
    if (n_reads == 0) return NP_OK;
    std::lock_guard<std::mutex> g(c->lock);
    NP_HIP(c, hipSetDevice(c->device));
    stream_scope scope = use_stream(c, stream); hipStream_t s = scope.s;
    family_timer tm(c, 4, s);
    NP_HIP(c, np_launch_reverse_events(n_reads, event_off, n_events, event_start, event_length, event_mean, event_stdv, s));
    return NP_OK;
}

} // extern "C"
