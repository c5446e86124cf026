// np_dropin.cpp -- the reference-side shim: the reference's own entry points, same signatures, forwarding to the
// C ABI (include/np_hmm.h).  It is compiled INSIDE a nanopolish build (it includes nanopolish's headers) in place of
//     src/hmm/nanopolish_profile_hmm.cpp      (profile_hmm_score x2, profile_hmm_score_set, profile_hmm_align)
//     src/nanopolish_raw_loader.cpp           (adaptive_banded_simple_event_align, estimate_scalings_using_mom)
// so that every caller -- basemods.cpp:374,382, nanopolish_variant.cpp:249,789-790,834, nanopolish_call_variants.cpp,
// nanopolish_eventalign.cpp:740, nanopolish_scorereads.cpp:164,191, nanopolish_phase_reads.cpp:289,292,
// nanopolish_squiggle_read.cpp:270 -- links unchanged.  See INTEGRATION.md.  oracle/Makefile builds exactly this
// configuration (`make -C oracle dropin`) and tests/test_gpu_dropin.py runs the reference's harness through it.
//
// One call = one tiny batch: correct, but launch-bound -- and the reference calls these functions from inside OpenMP loops
// (src/nanopolish_scorereads.cpp:164,388, src/nanopolish_phase_reads.cpp:289-292, basemods.cpp:374,382 under bam_processor.cpp:99):
// sixteen threads each paying an upload, a launch, a read-back and a synchronisation, one after the other on the context's lock.
// The scoring entry points therefore COMBINE concurrent callers (flat combining): a caller queues its work items; whoever finds
// no flush in progress becomes the combiner, takes everything queued so far -- what arrived while the previous flush was on the
// device -- and scores it with ONE np_hmm_score_host; the others sleep until their scores are there.  One thread still costs one
// launch per call; sixteen threads cost one launch per round (tests/bench_percall_dropin.py, profiles/r04_percall_dropin.md).
// Errors: the reference's callers link UNCHANGED, so none of them reads an error count -- a failed device call therefore ends the
// process (message on stderr, exit status 1: what the reference itself does when its own allocation fails, raw_loader.cpp:124-131),
// as it must not be possible to write NaN / -inf log-likelihoods into a TSV and exit 0.  A caller that HAS been adapted to check
// np_dropin_error_count() after its loop sets NP_DROPIN_ERRORS_IN_BAND=1: the call then reports (once per message), counts, and
// returns the reference's in-band "no result" (-INFINITY / an empty vector); even then an atexit handler turns the exit status of a
// process that ends with a non-zero count into 1.
// The throughput path remains the *_dev batch API fed at the BamProcessor batch boundary (np_batch_dropin.cpp).
#include <algorithm>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include "nanopolish_profile_hmm.h"
#include "nanopolish_profile_hmm_r9.h"
#include "nanopolish_profile_hmm_r7.h"
#include "nanopolish_raw_loader.h"
#include "np_hmm.h"
#include "np_shim_common.h"

extern double hmm_indel_bias_factor;   // src/hmm/nanopolish_profile_hmm_r9.cpp:19 (still the caller-visible knob)

using np_shim::shim;

extern "C" void np_dropin_invalidate_models(void) { shim().invalidate(); }

namespace {

std::atomic<long> g_errors(0);

bool errors_in_band()
{
    static const bool v = [] { const char* e = getenv("NP_DROPIN_ERRORS_IN_BAND"); return e && atoi(e) != 0; }();
    return v;
}

// process exit with failed calls on the books (in-band mode): an unmodified caller would exit 0 with corrupt output
void exit_status_guard()
{
    const long n = g_errors.load();
    if (n > 0) {
        fprintf(stderr, "nanopolish_amd: %ld device call(s) failed during this run; results are incomplete (exit status 1)\n", n);
        fflush(stderr);
        _exit(EXIT_FAILURE);
    }
}

// a failed device call: message, count, and the end of the process -- or, with NP_DROPIN_ERRORS_IN_BAND=1, back to the caller, who
// returns the entry point's in-band failure value (message once per distinct text)
void fail(const char* what, int rc)
{
    static std::mutex m;
    static std::set<std::string> seen;
    static std::once_flag guard;
    const std::string msg = std::string(what) + " failed (" + std::to_string(rc) + "): " + np_last_error(shim().get());
    g_errors.fetch_add(1);
    {
        std::lock_guard<std::mutex> g(m);
        if (seen.insert(msg).second) fprintf(stderr, "nanopolish_amd: %s\n", msg.c_str());
    }
    if (!errors_in_band()) { fflush(stderr); _exit(EXIT_FAILURE); }       // (_exit: no static destructors under the caller's running OpenMP team)
    std::call_once(guard, [] { atexit(exit_status_guard); });
}

// Per-item callers (scorereads, phase-reads: SURVEY section 2 OUT OF SCOPE, no batched binding) get SLOWER through this shim -- a
// synchronous device round trip per item, ~12 x the CPU function from sixteen threads (profiles/r04_percall_dropin.md).  Say so once,
// after enough calls that it is a throughput caller and not a test; NP_DROPIN_QUIET=1 silences it.
void note_per_call_use()
{
    static std::atomic<long> calls(0);
    if (calls.fetch_add(1, std::memory_order_relaxed) + 1 != 4096) return;
    const char* q = getenv("NP_DROPIN_QUIET");
    if (q && atoi(q) != 0) return;
    fprintf(stderr, "nanopolish_amd: note: profile_hmm_score* has been called 4096 times one item at a time.  Each call is a synchronous device round\n"
                    "  trip (~35 us of dependent work against ~20 us for the CPU function): per-item callers (scorereads, phase-reads) run slower\n"
                    "  through this shim than on the host.  Throughput callers use the batched bindings: np_calculate_methylation_for_batch /\n"
                    "  NpBatchPipeline (call-methylation), np_score_variants_thresholded (variants), np_realign_reads_batch (eventalign);\n"
                    "  INTEGRATION.md section 1.  NP_DROPIN_QUIET=1 silences this note.\n");
}

// Flat combining of concurrent scoring calls (see the header).  A request is a run of np_hmm_job with room for their scores.
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

class ScoreCombiner {
public:
    ScoreCombiner() : flushing_(false) {}
    // scores of jobs[0..n) into out[0..n); NP_OK or the library's error for THIS request
    int score(const np_hmm_job* jobs, int n, float* out)
    {
        if (n <= 0) return NP_OK;
        Req r; r.jobs = jobs; r.n = n; r.out = out; r.rc = NP_OK; r.done = false;
        std::unique_lock<std::mutex> g(m_);
        queue_.push_back(&r);
        queued_.store(queue_.size(), std::memory_order_release);
        while (!r.done) {
            if (flushing_) {
                // a round is ~100 us: spin for the hand-over first (a futex sleep and wake-up costs as much as the round itself), sleep
                // only when the device keeps the combiner for long
                g.unlock();
                for (int spin = 0; spin < 20000 && flushing_.load(std::memory_order_acquire) && !r.done.load(std::memory_order_acquire); ++spin) cpu_relax();
                g.lock();
                if (flushing_ && !r.done) cv_.wait_for(g, std::chrono::microseconds(200));
                continue;
            }
            flushing_ = true;                                   // this thread combines
            // The callers a round has just released need a few microseconds to come back with their next item; a combiner that collects
            // at once only ever sees the other half of the threads (two groups taking turns: 8 calls per round from 16 threads).  With
            // company in the last round it holds the door for up to ~12 us, or until as many items wait as the last two rounds carried.
            if (last_round_ >= 2) {
                const size_t target = last_round_ + prev_round_;
                g.unlock();
                const std::chrono::steady_clock::time_point until = std::chrono::steady_clock::now() + std::chrono::microseconds(12);
                while (queued_.load(std::memory_order_acquire) < target && std::chrono::steady_clock::now() < until) cpu_relax();
                g.lock();
            }
            std::vector<Req*> batch; batch.swap(queue_);
            queued_.store(0, std::memory_order_release);
            prev_round_ = last_round_; last_round_ = batch.size();
            g.unlock();
            flush(batch);
            g.lock();
            for (size_t i = 0; i < batch.size(); ++i) batch[i]->done = true;
            flushing_ = false;
            cv_.notify_all();
        }
        return r.rc;
    }
    long rounds() const { return rounds_.load(); }
    long requests() const { return requests_.load(); }
    long flush_ns() const { return flush_ns_.load(); }
private:
    struct Req { const np_hmm_job* jobs; int n; float* out; int rc; std::atomic<bool> done; };
    void flush(std::vector<Req*>& batch)
    {
        struct Clock { std::atomic<long>& acc; std::chrono::steady_clock::time_point t0; Clock(std::atomic<long>& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
                       ~Clock() { acc.fetch_add((long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count()); } } clk(flush_ns_);
        rounds_.fetch_add(1); requests_.fetch_add((long)batch.size());
        np_ctx* c = shim().get();
        if (batch.size() == 1) { batch[0]->rc = np_hmm_score_host(c, batch[0]->n, batch[0]->jobs, batch[0]->out); return; }
        size_t total = 0;
        for (size_t i = 0; i < batch.size(); ++i) total += (size_t)batch[i]->n;
        all_.resize(total); sc_.resize(total);
        size_t at = 0;
        for (size_t i = 0; i < batch.size(); ++i) { memcpy(&all_[at], batch[i]->jobs, (size_t)batch[i]->n * sizeof(np_hmm_job)); at += (size_t)batch[i]->n; }
        const int rc = np_hmm_score_host(c, (int)total, all_.data(), sc_.data());
        if (rc == NP_OK) {
            at = 0;
            for (size_t i = 0; i < batch.size(); ++i) { memcpy(batch[i]->out, &sc_[at], (size_t)batch[i]->n * sizeof(float)); batch[i]->rc = NP_OK; at += (size_t)batch[i]->n; }
            return;
        }
        // one request of the round is outside what the library covers (NP_ERR_UNSUPPORTED: more than NP_MAX_KMERS k-mers, ...): the
        // others must not fail with it -- every request again, on its own
        for (size_t i = 0; i < batch.size(); ++i) batch[i]->rc = np_hmm_score_host(c, batch[i]->n, batch[i]->jobs, batch[i]->out);
    }
    std::mutex m_; std::condition_variable cv_;
    std::vector<Req*> queue_;
    std::atomic<bool> flushing_;
    std::atomic<size_t> queued_{0};
    size_t last_round_ = 0, prev_round_ = 0;       // (under m_)
    std::vector<np_hmm_job> all_; std::vector<float> sc_;      // the combiner's buffers (one combiner at a time)
    std::atomic<long> rounds_{0}, requests_{0}, flush_ns_{0};
};
ScoreCombiner& combiner() { static ScoreCombiner c; return c; }

// (HMMInputSequence, HMMInputData) -> np_hmm_job.  `ev` and `ranks` own the flattened buffers.
struct FlatJob {
    std::vector<float> ev;
    std::vector<uint16_t> ranks;
    np_hmm_job job;
};

void flatten(const HMMInputSequence& sequence, const HMMInputData& data, uint32_t flags, FlatJob& f)
{
    const SquiggleRead* read = data.read;
    const uint8_t strand = data.strand;
    const SquiggleScalings& sc = read->scalings[strand];
    assert(sc.drift == 0.0);                 // always 0 on the R9 path (squiggle_read.cpp:310, raw_loader.cpp:52)
    assert((data.rc && data.event_stride == -1) || (!data.rc && data.event_stride == 1));   // r9.inl:275
    const uint32_t k = data.pore_model->k;
    const uint32_t n_kmers = sequence.length() - k + 1;
    assert(data.pore_model->states.size() == sequence.get_num_kmer_ranks(k));               // r9.inl:305
    const uint32_t lo = std::min(data.event_start_idx, data.event_stop_idx), hi = std::max(data.event_start_idx, data.event_stop_idx);
    f.ev.resize(hi - lo + 1);
    for (uint32_t e = lo; e <= hi; ++e) f.ev[e - lo] = read->events[strand][e].mean;
    f.ranks.resize(n_kmers);
    for (uint32_t i = 0; i < n_kmers; ++i) f.ranks[i] = (uint16_t)sequence.get_kmer_rank(i, k, data.rc);
    np_hmm_job& j = f.job;
    j.event_mean = f.ev.data() - lo;         // the library only touches indices [lo, hi]
    j.n_events_total = (uint32_t)read->events[strand].size();
    j.e_start = data.event_start_idx; j.e_stop = data.event_stop_idx; j.stride = data.event_stride;
    j.kmer_rank = f.ranks.data(); j.n_kmers = n_kmers;
    j.model = shim().model_id(data.pore_model);
    j.scale = sc.scale; j.shift = sc.shift; j.var = sc.var;
    j.events_per_base = read->events_per_base[strand];
    j.flags = flags; j.reserved = 0;
    j.indel_bias = hmm_indel_bias_factor;
}

} // namespace

// src/hmm/nanopolish_profile_hmm.cpp:23-30
float profile_hmm_score(const HMMInputSequence& sequence, const HMMInputData& data, const uint32_t flags)
{
    if (data.read->pore_type != PORETYPE_R9) return profile_hmm_score_r7(sequence, data, flags);
    FlatJob f;
    flatten(sequence, data, flags, f);
    float out = 0.0f;
    note_per_call_use();
    const int rc = combiner().score(&f.job, 1, &out);
    if (rc != NP_OK) { fail("profile_hmm_score: np_hmm_score_host", rc); return -INFINITY; }
    return out;
}

// src/hmm/nanopolish_profile_hmm.cpp:14-21 -- one device batch over all reads, summed in index order in float
float profile_hmm_score(const HMMInputSequence& sequence, const std::vector<HMMInputData>& data, const uint32_t flags)
{
    std::vector<FlatJob> f(data.size());
    std::vector<np_hmm_job> jobs(data.size());
    for (size_t i = 0; i < data.size(); ++i) {
        assert(data[i].read->pore_type == PORETYPE_R9);
        flatten(sequence, data[i], flags, f[i]);
        jobs[i] = f[i].job;
    }
    std::vector<float> sc(data.size());
    const int rc = combiner().score(jobs.data(), (int)jobs.size(), sc.data());
    if (rc != NP_OK) { fail("profile_hmm_score (vector): np_hmm_score_host", rc); return -INFINITY; }
    float score = 0.0f;
    for (size_t i = 0; i < sc.size(); ++i) score += sc[i];
    return score;
}

// src/hmm/nanopolish_profile_hmm.cpp:32-56
float profile_hmm_score_set(const std::vector<HMMInputSequence>& sequences, const HMMInputData& data, const uint32_t flags)
{
    assert(!sequences.empty());
    assert(std::string(sequences[0].get_alphabet()->get_name()) == "nucleotide");
    assert(std::string(data.pore_model->pmalphabet->get_name()) == "nucleotide");
    std::vector<FlatJob> f(sequences.size());
    std::vector<np_hmm_job> jobs(sequences.size());
    HMMInputData alt = data;
    for (size_t s = 0; s < sequences.size(); ++s) {
        if (s > 0) {
            alt.pore_model = alt.read->get_model(alt.strand, sequences[s].get_alphabet()->get_name());
            assert(alt.pore_model != NULL);
        }
        flatten(sequences[s], s == 0 ? data : alt, flags, f[s]);
        jobs[s] = f[s].job;
    }
    // the members' forward scores through the combiner, then profile_hmm.cpp:38-54's own combination (the reference's add_logs)
    std::vector<float> sc(jobs.size());
    note_per_call_use();
    const int rc = combiner().score(jobs.data(), (int)jobs.size(), sc.data());
    if (rc != NP_OK) { fail("profile_hmm_score_set: np_hmm_score_host", rc); return -INFINITY; }
    const size_t num_models = sequences.size();
    const double num_model_penalty = log(num_models);
    double score = sc[0] - num_model_penalty;
    for (size_t q = 1; q < num_models; ++q) {
        const double alt_score = sc[q] - num_model_penalty;
        score = add_logs(score, alt_score);
    }
    return score;
}

// src/hmm/nanopolish_profile_hmm.cpp:58-65
std::vector<HMMAlignmentState> profile_hmm_align(const HMMInputSequence& sequence, const HMMInputData& data, const uint32_t flags)
{
    if (data.read->pore_type != PORETYPE_R9) return profile_hmm_align_r7(sequence, data, flags);
    FlatJob f;
    flatten(sequence, data, flags, f);
    const int64_t cap = (int64_t)f.ev.size() + f.job.n_kmers + 2;
    std::vector<np_hmm_state> st(cap);
    int64_t off[2] = {0, 0};
    const int rc = np_hmm_align_host(shim().get(), 1, &f.job, st.data(), cap, off);
    if (rc != NP_OK) { fail("profile_hmm_align: np_hmm_align_host", rc); return std::vector<HMMAlignmentState>(); }
    std::vector<HMMAlignmentState> out(off[1]);
    for (int64_t i = 0; i < off[1]; ++i) {
        out[i].event_idx = st[i].event_idx; out[i].kmer_idx = st[i].kmer_idx;
        out[i].l_posterior = -INFINITY; out[i].l_fm = st[i].l_fm; out[i].log_transition_probability = -INFINITY;
        out[i].state = st[i].state;
    }
    return out;
}

// src/nanopolish_raw_loader.cpp:17-60
SquiggleScalings estimate_scalings_using_mom(const std::string& sequence, const PoreModel& pore_model, const event_table& et)
{
    const size_t k = pore_model.k, n_kmers = sequence.size() - k + 1;
    std::vector<uint16_t> ranks(n_kmers);
    for (size_t i = 0; i < n_kmers; ++i) ranks[i] = (uint16_t)pore_model.pmalphabet->kmer_rank(sequence.c_str() + i, k);
    std::vector<double> lm(pore_model.states.size());
    for (size_t i = 0; i < lm.size(); ++i) lm[i] = pore_model.states[i].level_mean;
    std::vector<float> ev(et.n);
    for (size_t i = 0; i < et.n; ++i) ev[i] = et.event[i].mean;
    double shift, scale;
    np_estimate_scalings_mom(lm.data(), ranks.data(), (uint32_t)n_kmers, ev.data(), (uint32_t)et.n, &shift, &scale);
    SquiggleScalings out;
    out.set4(shift, scale, 0.0, 1.0);
    return out;
}

// src/nanopolish_raw_loader.cpp:77-379
std::vector<AlignedPair> adaptive_banded_simple_event_align(SquiggleRead& read, const PoreModel& pore_model, const std::string& sequence)
{
    const size_t strand_idx = 0, k = pore_model.k;
    const size_t n_events = read.events[strand_idx].size(), n_kmers = sequence.size() - k + 1;
    assert(read.scalings[strand_idx].drift == 0.0);
    std::vector<float> ev(n_events);
    for (size_t i = 0; i < n_events; ++i) ev[i] = read.events[strand_idx][i].mean;
    std::vector<uint16_t> ranks(n_kmers);
    for (size_t i = 0; i < n_kmers; ++i) ranks[i] = (uint16_t)pore_model.pmalphabet->kmer_rank(sequence.c_str() + i, k);
    np_align_job j;
    j.event_mean = ev.data(); j.n_events = (uint32_t)n_events; j.kmer_rank = ranks.data(); j.n_kmers = (uint32_t)n_kmers;
    j.model = shim().model_id(&pore_model);
    j.scale = read.scalings[strand_idx].scale; j.shift = read.scalings[strand_idx].shift; j.var = read.scalings[strand_idx].var;
    const int64_t cap = (int64_t)n_events + n_kmers + 2;
    std::vector<np_pair> pairs(cap);
    int64_t off[2] = {0, 0};
    const int rc = np_event_align_host(shim().get(), 1, &j, pairs.data(), cap, off);
    if (rc != NP_OK) { fail("adaptive_banded_simple_event_align: np_event_align_host", rc); return std::vector<AlignedPair>(); }
    std::vector<AlignedPair> out(off[1]);
    for (int64_t i = 0; i < off[1]; ++i) { out[i].ref_pos = pairs[i].ref_pos; out[i].read_pos = pairs[i].read_pos; }
    return out;
}

// Failed device calls since the process started (each reported on stderr once per distinct message): the caller's serial code checks
// this after a parallel region whose calls returned -INFINITY / empty vectors.
extern "C" long np_dropin_error_count(void) { return g_errors.load(); }
// diagnostics of the combiner: device rounds and the scoring calls they carried (calls / rounds = callers combined per launch)
extern "C" void np_dropin_combiner_stats(long* rounds, long* calls) { if (rounds) *rounds = combiner().rounds(); if (calls) *calls = combiner().requests(); }
extern "C" long np_dropin_combiner_flush_ns(void) { return combiner().flush_ns(); }        // wall time inside the rounds' np_hmm_score_host calls
