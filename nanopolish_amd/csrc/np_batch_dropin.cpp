// np_batch_dropin.cpp -- the throughput binding on the reference side: call-methylation's per-record work for whole
// BamProcessor batches, on the device, as a three-stage pipeline with its own host threads, over one or several GPUs.
//
// In the reference every record of a batch runs, under `#pragma omp parallel for` (src/common/nanopolish_bam_processor.cpp:99-106),
//     calculate_methylation_for_read_from_bam            src/nanopolish_call_methylation.cpp:163-177
//       SquiggleRead sr(read_name, read_db)               load_from_raw: detect_events, MoM scalings, event alignment, event map,
//                                                         recalibrate_model, QC gates (src/nanopolish_squiggle_read.cpp:141-336)
//       calculate_methylation_for_read(..., sr, ...)      src/basemods/nanopolish_basemods.cpp:238-419
// and the batch's writer then walks the result maps and clears them (write_methylation_results_for_batch, call_methylation.cpp:552-588).
// This file is compiled INSIDE a nanopolish build (it includes nanopolish's headers) and splits that loop in three stages:
//   phase 1 (host, the pipeline's PACKER thread + worker pool): what only the host can do -- the read's sequence and raw samples
//            (ReadDB / slow5 / fast5: the caller's NpBatchRead), the reference segment (faidx), the CIGAR -- packed into ONE pinned blob;
//   phase 2 (device): ONE upload of that blob, then the whole batch in seven enqueues through the C ABI (include/np_hmm.h)
//            np_cm_build_jobs_cigar_dev -> np_detect_events_dev -> np_mom_fill_dev -> np_event_align_dev ->
//            np_calibrate_resolve_dev -> np_cm_discard_degenerate_dev -> np_hmm_score_dev, then ONE read-back of the output blob;
//   phase 3 (host, the FINISHER thread + worker pool): one ScoredSite map per record from the scores, exactly the fields
//            basemods.cpp:384-413 fills.
// Round 3's form ran phases 1 and 3 on the CALLER's OpenMP team, one after the other around the wait for the device: at 8 192
// records the binding took 37 ms of the caller's time per batch and the harness's stand-in for the writer (count the sites, clear the
// maps: 1.5 M ScoredSites, two heap blocks each, serially) 79 ms more -- 70 k reads/s against 340 k for the device pass alone.  Now
// the three stages of three consecutive batches run at the same time on threads of the pipeline's own (np_pool.h), submit() only
// queues, collect() swaps finished maps into the caller's result, and recycle() takes written-out maps back to destroy them on the
// workers.  With several devices the batches are dealt round-robin to one context per GPU; results come back in submission order.
// oracle/Makefile builds the reference with this file in (`make -C oracle batch`), tests/test_gpu_batch_dropin.py feeds it
// the records of tests/golden/golden_reflevel.npz and expects the maps the unmodified reference produced (one context, and two
// contexts on one device); tests/bench_batch_dropin.py times it at BamProcessor-like batch sizes.  INTEGRATION.md section 2 shows the call site.
#include <malloc.h>
#include <sched.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include "np_batch_dropin.h"
#include "nanopolish_alphabet.h"
#include "nanopolish_eventalign.h"          // get_reference_region_ts
#include "nanopolish_pore_model_set.h"
#include "np_hmm.h"
#include "np_pool.h"
#include "np_shim_common.h"

using np_shim::shim;
using np_shim::check;
using np_shim::die;
using np_shim::Layout;
using np_shim::Blob;
using np_shim::Pool;

namespace {

int g_event_cap_divisor = 2;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// CPUs this process may use: the affinity mask, capped by the cgroup CPU quota (a container that SEES 256 hardware threads may be
// granted 16: a pool sized by the former would run 256 threads on 16 CPUs)
int usable_cpus()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    {
        std::ifstream f("/sys/fs/cgroup/cpu.max");                       // cgroup v2: "<quota> <period>" or "max <period>"
        std::string q; double period = 0.0;
        if (f >> q >> period && q != "max" && period > 0.0) n = std::min(n, std::max(1, (int)(atof(q.c_str()) / period + 0.5)));
    }
    {
        std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");      // cgroup v1
        double q = 0.0, per = 0.0;
        if (fq >> q && fp >> per && q > 0.0 && per > 0.0) n = std::min(n, std::max(1, (int)(q / per + 0.5)));
    }
    return std::max(1, n);
}

// 0..3 for A, C, G, T (DNAAlphabet's ranks), 4 for anything else
inline int base_code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }

bool is_plain_acgt(const std::string& s)
{
    for (size_t i = 0; i < s.size(); ++i) if (base_code(s[i]) > 3) return false;
    return true;
}

// gDNAAlphabet.kmer_rank(seq + j, k) for every k-mer of the read (Alphabet::kmer_rank, nanopolish_alphabet.h: the base-4 number of the
// k-mer's base ranks), as one rolling pass; a read with a character outside ACGT goes through the reference's own function
void nucleotide_kmer_ranks(const std::string& seq, uint32_t k, uint16_t* out)
{
    const size_t n = seq.size() >= k ? seq.size() - k + 1 : 0;
    if (!is_plain_acgt(seq)) {
        for (size_t j = 0; j < n; ++j) out[j] = (uint16_t)gDNAAlphabet.kmer_rank(seq.c_str() + j, k);
        return;
    }
    const uint32_t mask = (1u << (2 * k)) - 1u;
    uint32_t r = 0;
    for (size_t i = 0; i < seq.size(); ++i) {
        r = ((r << 2) | (uint32_t)base_code(seq[i])) & mask;
        if (i + 1 >= k) out[i + 1 - k] = (uint16_t)r;
    }
}

struct RefView { const char* p; size_t n; RefView() : p(NULL), n(0) {} };

// One DEVICE PASS: the records of one or several consecutive batches (round 5: a batch of BamProcessor's default 512 records is a 2.5 ms
// pass behind ~8 ms of latencies -- one wave per read walks 13 000 dependent band steps whatever the batch holds -- so the packer merges
// the batches that are already waiting, up to NP_BATCH_COALESCE records, into one upload, one run of the kernels and one read-back;
// results stay per batch, in submission order).  Three per device: one being packed, one on the device, one being turned into maps.
struct Pass {
    int dev;                                // index into Impl::devs
    Blob in, out;
    void *ev_h2d, *ev_cmp, *ev_d2h;
    std::map<int, std::shared_ptr<const std::string> > union_seq;   // tid -> the reference stretch the pass's records on that contig cover (shared with
    std::map<int, int> union_lo;                                    // the packer's stretch cache: alive for as long as any pass's views point into it)
    std::vector<int64_t> group_off;         // group slots of the pass's device records
    size_t o_scores, o_first, o_last, o_n_motif, o_n_groups, o_n_events, o_n_pairs, o_calibrated, out_bytes;   // offsets into `out`
    int unfinished;                         // member batches whose maps are not built yet (under Impl::m): 0 = free for the packer
    bool on_device;                         // enqueued and its read-back not seen complete yet (packer thread only)
    Pass() : dev(0), in(true), out(true), ev_h2d(NULL), ev_cmp(NULL), ev_d2h(NULL), out_bytes(0), unfinished(0), on_device(false) {}
};

// what one batch in flight needs on the host until it is collected
struct Slot {
    Pass* pass;                             // the device pass its records went into (NULL: none of them goes to the device)
    std::vector<NpBatchRead>* caller;       // the caller's vector: statuses go back into it at collect()
    std::vector<NpBatchRead> rec;           // its entries as they were at submit() (the buffers they point to stay the caller's)
    std::vector<std::string> own_seq;       // a record's own segment when no union serves it, or when it needed disambiguation
    std::vector<RefView> ref;               // every record's reference segment (a view into the pass's union_seq / own_seq)
    std::vector<int> ref_start, dev_index, status;     // dev_index: the record's index among the PASS's device records
    std::vector<std::map<int, ScoredSite> > built;      // phase 3's output, swapped into the caller's result by collect()
    std::vector<int> builder;               // the pool worker that built (allocated) each record's map
    size_t first;                           // index, in the caller's vector, of this piece's first record (round 6: a large batch travels in pieces)
    int n_dev;
    bool finished;                          // phase 3 done, not collected yet (under Impl::m)
    Slot() : pass(NULL), caller(NULL), first(0), n_dev(0), finished(false) {}
};

struct DevState {
    int ordinal;                            // the HIP device the context lives on (two contexts may share one)
    int slot_key;                           // context slot of the process-wide shim
    np_ctx* c;
    Blob scratch;                           // single per context: its compute stream runs one batch at a time
    void *s_h2d, *s_d2h;
    DevState() : ordinal(0), slot_key(0), c(NULL), scratch(false), s_h2d(NULL), s_d2h(NULL) {}
};

} // namespace

struct NpBatchPipeline::Impl {
    MethylationCallingParameters params;
    std::string kit;
    const faidx_t* fai; const bam_hdr_t* hdr;
    int region_start, region_end;
    std::vector<DevState*> devs;
    std::vector<Slot*> slots;               // NP_BATCH_SLOTS per device (default: three passes of NP_BATCH_COALESCE records in 512-record batches = 48); batch b uses slot b % slots.size()
    std::vector<Pass*> passes;              // 3 per context: passes[3 * d + j] belongs to devs[d]
    long n_passes;                          // device passes started so far (packer thread only)
    long coalesce_records;                  // a pass takes waiting batches while it holds fewer records than this (NP_BATCH_COALESCE, default 8192)
    long last_batch_records;                // size of the most recently submitted batch (under m): max_in_flight() scales with it
    // Round 6: a submitted batch of more than 2 x piece_records records is cut into PIECES of piece_records (NP_BATCH_PIECE, default 512: BamProcessor's own batch size; at most a third of the slots per batch), each in a
    // slot of its own, and collect() hands the batch back when its last piece is finished.  Everything between submit() and collect() -- pass
    // formation, map building by two finishers, pass buffers freed piece by piece -- then works at the granularity that measured best: through
    // one box's binding 8 192-record batches ran at 157 k reads/s whole, 169-192 k in pieces of 1 024 and 245 k in pieces of 512, against 312 k for 512-record batches (gpurun r06g, r06h), the same records in the same
    // 8 192-record device passes, because a whole-batch slot builds 1.5 M map nodes in one job and holds its pass buffers until the last one.
    long piece_records;
    bool maps_first, pack_first;            // NP_BATCH_MAPS_FIRST / NP_BATCH_PACK_FIRST (default 0 / 0: first come, first served): which loops go to the front of the pool's queue.
                                            // Measured (gpurun r06o): maps first 172-184 k reads/s at 512 records against 284-320 k -- the maps get quicker and the packer, which feeds
                                            // the device, starves
    std::deque<int> batch_pieces;           // pieces of every caller batch in flight, oldest first (under m)
    bool presized;                          // the buffers have been sized for a full merged pass (packer thread only)
    Pool* pool;
    std::thread packer;
    std::vector<std::thread> finisher;      // NP_BATCH_FINISHERS (default 2): one waits for batch k+1's read-back while another builds batch k's maps
    std::mutex m; std::condition_variable cv;
    long n_submitted, n_packed, n_claimed, n_finished, n_collected;      // batches that have passed each stage (under m); n_claimed: taken by a finisher
    bool stop;
    // Round 6: the reference stretches the packer fetched last, one per contig (packer thread only; cleared by configure()).  Consecutive passes of
    // a sorted BAM cover overlapping stretches, and faidx_fetch_seq -- serial, under the reference's lock, a line-by-line copy out of the FASTA --
    // was 10 of the 15 ms phase 1a took per 8 192-record pass: a pass whose span lies inside the cached stretch fetches nothing, one that reaches
    // past it fetches the union (while that stays under the 64 MB a pass may hold) and replaces the entry.
    struct RefStretch { int lo, hi; std::shared_ptr<const std::string> seq; };
    std::map<int, RefStretch> ref_cache;
    std::mutex tm; double t[8];
    // which pool worker allocated the map of a record handed out by collect(): recycle() sends every map back to ITS worker, so that a
    // heap block is freed by the thread that allocated it (glibc keeps an arena per thread: a free from another thread takes that
    // arena's lock -- sixteen workers freeing each other's blocks while sixteen build the next batch spent 147 ms on a batch whose
    // maps take 9 ms to build).  collect() and recycle() are the caller's: one thread.
    std::map<const bam1_t*, int> builder_of;
    bool track_builders;          // false: the synchronous pipeline (nobody recycles: nothing to remember)
    Impl() : fai(NULL), hdr(NULL), region_start(-1), region_end(-1), pool(NULL), n_submitted(0), n_packed(0), n_claimed(0), n_finished(0), n_collected(0), stop(false) { n_passes = 0; coalesce_records = 8192; last_batch_records = 0; piece_records = 512; maps_first = false; pack_first = false; presized = false; track_builders = true; for (int i = 0; i < 8; ++i) t[i] = 0.0; }
    void add_time(int i, double s) { std::lock_guard<std::mutex> g(tm); t[i] += s; }
    void open(const std::vector<int>& devices, bool shared_default, int host_threads);
    void pack(const std::vector<Slot*>& group, int dev);
    int device_passes(int ordinal);
    void finish(Slot& S);
    void packer_loop();
    void finisher_loop();
};

void NpBatchPipeline::Impl::open(const std::vector<int>& devices, bool shared_default, int host_threads)
{
    for (size_t d = 0; d < devices.size(); ++d) {
        DevState* D = new DevState();
        D->ordinal = devices[d];
        D->slot_key = (shared_default && d == 0) ? 0 : shim().take_slot(devices[d]);      // (shared_default: the first entry is the process-wide context)
        D->c = shim().ctx(D->slot_key);
        D->s_h2d = np_stream_create(D->c); D->s_d2h = np_stream_create(D->c);
        if (!D->s_h2d || !D->s_d2h) die(np_last_error(D->c));
        devs.push_back(D);
    }
    for (size_t i = 0; i < 3 * devs.size(); ++i) {
        Pass* P = new Pass();
        P->dev = (int)(i / 3);
        np_ctx* c = devs[P->dev]->c;
        P->ev_h2d = np_event_create(c); P->ev_cmp = np_event_create(c); P->ev_d2h = np_event_create(c);
        if (!P->ev_h2d || !P->ev_cmp || !P->ev_d2h) die(np_last_error(c));
        passes.push_back(P);
    }
    // batches in flight: enough small ones to fill three passes per device (max_in_flight() scales the number handed to the caller with
    // the batch size: large batches stay at three per device, as before)
    if (const char* v = getenv("NP_BATCH_COALESCE")) coalesce_records = std::max(1L, atol(v));
    // three passes of coalesce_records records each, in BamProcessor's 512-record batches: 48 slots at the default 8 192 (24 held 1.5
    // passes, so pass N+1 was only formed once the device had gone idle and every other pass ran half-size)
    int per_dev = (int)std::max(24L, std::min(64L, 3 * coalesce_records / 512));
    if (const char* v = getenv("NP_BATCH_SLOTS")) per_dev = std::max(3, std::min(64, atoi(v)));
    if (const char* v = getenv("NP_BATCH_PIECE")) piece_records = std::max(1L, atol(v));
    if (const char* v = getenv("NP_BATCH_MAPS_FIRST")) maps_first = atoi(v) != 0;
    if (const char* v = getenv("NP_BATCH_PACK_FIRST")) pack_first = atoi(v) != 0;
    if (!track_builders) { per_dev = 3; piece_records = 1L << 40; }      // the synchronous pipeline: one batch at a time, whole
    for (size_t i = 0; i < (size_t)per_dev * devs.size(); ++i) slots.push_back(new Slot());
    // Freed map memory goes back to the allocator, not to the kernel: with glibc's default trim threshold (128 KB) every batch's
    // 300 MB of released nodes is unmapped page by page and faulted in again by the next batch (seen as system time, and as a 7 ms
    // packing loop taking 58).  NP_KEEP_MALLOC_DEFAULTS=1 leaves the process's settings alone.
    if (!getenv("NP_KEEP_MALLOC_DEFAULTS")) { (void)mallopt(M_TRIM_THRESHOLD, 1 << 30); (void)mallopt(M_TOP_PAD, 64 << 20); }
    int nt = host_threads;
    if (nt <= 0) { const char* v = getenv("NP_HOST_THREADS"); nt = v ? atoi(v) : 0; }
    if (nt <= 0) { const int cpus = usable_cpus(); nt = cpus + cpus / 4; }      // (the workers stall on memory and on the allocator: 20 threads on 16 CPUs
                                                                                //  measured 7 % over 16, 24 no better -- profiles/r04_batch_binding.md)
    pool = new Pool(std::min(nt, 256));
    packer = std::thread(&Impl::packer_loop, this);
    int nf = 2;
    if (const char* v = getenv("NP_BATCH_FINISHERS")) nf = std::max(1, std::min(16, atoi(v)));
    for (int i = 0; i < nf; ++i) finisher.push_back(std::thread(&Impl::finisher_loop, this));
}

NpBatchPipeline::NpBatchPipeline(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                                 const bam_hdr_t* hdr, int region_start, int region_end) : p(new Impl())
{
    // One GPU: the process-wide context (NP_DEVICE) and, beside it, a second context on the same device -- two device passes in flight hide
    // the per-batch latencies (copies, ~12 launches, stream waits) that a 512-record batch (BamProcessor's default) cannot amortise:
    // 65 k -> 95 k reads/s at 512 records, 186 k -> 225 k at 2 048, the same at 8 192; four contexts are no better (profiles/r04_batch_binding.md).
    // NP_BATCH_CONTEXTS=1 keeps it to one (half the device scratch).
    const char* dev = getenv("NP_DEVICE"); const int device = dev ? atoi(dev) : 0;
    const char* nc = getenv("NP_BATCH_CONTEXTS"); const int contexts = nc ? std::max(1, std::min(4, atoi(nc))) : 2;
    p->open(std::vector<int>((size_t)contexts, device), true, 0);
    configure(calling_parameters, kit, fai, hdr, region_start, region_end);
}

NpBatchPipeline::NpBatchPipeline(synchronous_t, const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                                 const bam_hdr_t* hdr, int region_start, int region_end) : p(new Impl())
{
    const char* dev = getenv("NP_DEVICE"); const int device = dev ? atoi(dev) : 0;
    p->track_builders = false;
    p->open(std::vector<int>(1, device), true, 0);
    configure(calling_parameters, kit, fai, hdr, region_start, region_end);
}

NpBatchPipeline::NpBatchPipeline(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                                 const bam_hdr_t* hdr, int region_start, int region_end, const std::vector<int>& devices, int host_threads) : p(new Impl())
{
    if (devices.empty()) die("NpBatchPipeline: an empty device list");
    p->open(devices, false, host_threads);
    configure(calling_parameters, kit, fai, hdr, region_start, region_end);
}

NpBatchPipeline::~NpBatchPipeline()
{
    {
        std::unique_lock<std::mutex> g(p->m);
        while (p->n_finished < p->n_submitted) p->cv.wait(g);          // batches nobody collected still run to their end
        p->stop = true;
    }
    p->cv.notify_all();
    p->packer.join();
    for (size_t i = 0; i < p->finisher.size(); ++i) p->finisher[i].join();
    p->pool->drain();
    delete p->pool;
    for (size_t i = 0; i < p->passes.size(); ++i) {
        Pass* P = p->passes[i];
        np_ctx* c = p->devs[P->dev]->c;
        (void)np_sync(c, p->devs[P->dev]->s_h2d); (void)np_sync(c, NULL); (void)np_sync(c, p->devs[P->dev]->s_d2h);
        P->in.release(c); P->out.release(c);
        np_event_destroy(c, P->ev_h2d); np_event_destroy(c, P->ev_cmp); np_event_destroy(c, P->ev_d2h);
        delete P;
    }
    for (size_t i = 0; i < p->slots.size(); ++i) delete p->slots[i];
    for (size_t d = 0; d < p->devs.size(); ++d) {
        DevState* D = p->devs[d];
        D->scratch.release(D->c);
        np_stream_destroy(D->c, D->s_h2d); np_stream_destroy(D->c, D->s_d2h);
        shim().release_slot(D->slot_key);
        delete D;
    }
    delete p;
}

void NpBatchPipeline::configure(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                                const bam_hdr_t* hdr, int region_start, int region_end)
{
    if (in_flight() != 0) die("NpBatchPipeline::configure with batches in flight");
    p->params = calling_parameters; p->kit = kit; p->fai = fai; p->hdr = hdr; p->region_start = region_start; p->region_end = region_end;
    p->ref_cache.clear();                    // (another FASTA index may stand behind the same contig ids)
}

int NpBatchPipeline::in_flight() const { std::lock_guard<std::mutex> g(p->m); return (int)p->batch_pieces.size(); }
// pieces a batch of `records` travels in: of `piece` records each, but never more than a third of the slots (three such batches stay in flight)
static long pieces_of(long records, long piece, long n_slots) { return records > 2 * piece ? std::max(1L, std::min((records + piece - 1) / piece, n_slots / 3)) : 1; }
// Batches the caller may keep in flight: three device passes per device, each of up to NP_BATCH_COALESCE records -- three batches per device
// when a batch fills a pass on its own, more (up to the slots there are) when batches are small.  Before the first submit(): the upper bound.
int NpBatchPipeline::max_in_flight() const
{
    std::lock_guard<std::mutex> g(p->m);
    const long last = p->last_batch_records;
    if (last <= 0) return (int)p->slots.size();
    const long n_dev = (long)p->devs.size();
    const long per_dev = std::max(3L, (3 * p->coalesce_records + last - 1) / last);
    const long by_slots = std::max(1L, (long)p->slots.size() / pieces_of(last, p->piece_records, (long)p->slots.size()));
    return (int)std::min(by_slots, per_dev * n_dev);
}
// what max_in_flight() will answer once batches of `batch_records` records are being submitted: the number of record / result vectors a caller
// that knows its batch size has to rotate (the upper bound max_in_flight() gives before the first submit() is 96 vectors -- of 8 192 records each
// that is 786 000 records held for six that are ever in flight)
int NpBatchPipeline::max_in_flight_for(size_t batch_records) const
{
    std::lock_guard<std::mutex> g(p->m);
    const long last = std::max(1L, (long)batch_records), n_dev = (long)p->devs.size();
    const long per_dev = std::max(3L, (3 * p->coalesce_records + last - 1) / last);
    const long by_slots = std::max(1L, (long)p->slots.size() / pieces_of(last, p->piece_records, (long)p->slots.size()));
    return (int)std::min(by_slots, per_dev * n_dev);
}
int NpBatchPipeline::devices() const { return (int)p->devs.size(); }
void NpBatchPipeline::host_seconds(double out[8]) const { std::lock_guard<std::mutex> g(p->tm); for (int i = 0; i < 8; ++i) out[i] = p->t[i]; }

void NpBatchPipeline::submit(std::vector<NpBatchRead>& reads)
{
    const double t0 = now();
    {
        std::lock_guard<std::mutex> g(p->m);
        const long n = (long)reads.size(), np0 = pieces_of(n, p->piece_records, (long)p->slots.size()), len = np0 > 1 ? (n + np0 - 1) / np0 : n;
        const long np_ = np0 > 1 ? (n + len - 1) / len : 1;               // (no empty pieces)
        if (p->n_submitted - p->n_collected + np_ > (long)p->slots.size()) die("NpBatchPipeline::submit: max_in_flight() batches are in flight already (collect one first)");
        for (long q = 0; q < np_; ++q) {
            Slot& S = *p->slots[(p->n_submitted + q) % (long)p->slots.size()];
            S.caller = &reads;
            S.first = (size_t)(q * len);
            // (the records, sequences and samples they point to stay the caller's until collect())
            S.rec.assign(reads.begin() + S.first, reads.begin() + std::min(n, (q + 1) * len));
        }
        p->last_batch_records = n;
        p->n_submitted += np_;
        p->batch_pieces.push_back((int)np_);
    }
    p->cv.notify_all();
    p->add_time(7, now() - t0);
}

// passes enqueued on the HIP device `ordinal` (any of its contexts) whose read-back has not completed (packer thread; polls the events)
int NpBatchPipeline::Impl::device_passes(int ordinal)
{
    int busy = 0;
    for (size_t i = 0; i < passes.size(); ++i) {
        Pass& P = *passes[i];
        if (!P.on_device || devs[P.dev]->ordinal != ordinal) continue;
        if (np_event_query(devs[P.dev]->c, P.ev_d2h) == NP_OK) P.on_device = false; else busy += 1;
    }
    return busy;
}

// When does a pass start?  A wave walks one read's 13 000 dependent band steps whatever the batch holds: a pass of 512 records keeps a
// twentieth of the device busy for as long as a pass of 5 000.  So the packer lets waiting batches gather -- up to coalesce_records
// records per pass -- for as long as the GPU has a pass to work on, and takes whatever waits the moment a GPU has none (a caller that
// submits slowly is never made to wait for company; a caller that has stopped submitting because max_in_flight() batches are in flight
// is waiting for batches that are on a device already).  The context: the least loaded one of the least loaded GPU.
void NpBatchPipeline::Impl::packer_loop()
{
    for (;;) {
        std::vector<Slot*> group;
        int dev = 0;
        {
            std::unique_lock<std::mutex> g(m);
            for (;;) {
                if (stop) return;
                if (n_packed < n_submitted) {
                    long records = 0;
                    for (long b = n_packed; b < n_submitted; ++b) records += (long)slots[b % (long)slots.size()]->rec.size();
                    // least loaded GPU, then least loaded context on it
                    int best_gpu = -1, best_gpu_busy = 1 << 30, best_ctx_busy = 1 << 30;
                    for (size_t d = 0; d < devs.size(); ++d) {
                        const int gb = device_passes(devs[d]->ordinal);
                        int cb = 0;
                        for (int q = 0; q < 3; ++q) cb += passes[3 * d + q]->on_device ? 1 : 0;
                        if (gb < best_gpu_busy || (gb == best_gpu_busy && cb < best_ctx_busy)) { best_gpu = (int)d; best_gpu_busy = gb; best_ctx_busy = cb; }
                    }
                    if ((records >= coalesce_records || best_gpu_busy == 0) && best_ctx_busy < 3) { dev = best_gpu; break; }
                    cv.wait_for(g, std::chrono::microseconds(200));            // (a pass finishing on the device signals no condition variable: poll)
                    continue;
                }
                cv.wait(g);
            }
            long records = 0;
            for (long b = n_packed; b < n_submitted && (group.empty() || records < coalesce_records); ++b) {
                Slot* S = slots[b % (long)slots.size()];
                if (!group.empty() && records + (long)S->rec.size() > coalesce_records) break;
                group.push_back(S); records += (long)S->rec.size();
            }
        }
        pack(group, dev);
        { std::lock_guard<std::mutex> g(m); n_packed += (long)group.size(); }
        cv.notify_all();
    }
}

void NpBatchPipeline::Impl::finisher_loop()
{
    for (;;) {
        Slot* S;
        {
            std::unique_lock<std::mutex> g(m);
            while (!stop && n_claimed >= n_packed) cv.wait(g);
            if (stop) return;
            S = slots[n_claimed % (long)slots.size()];
            n_claimed += 1;
        }
        finish(*S);
        { std::lock_guard<std::mutex> g(m); S->finished = true; n_finished += 1; if (S->pass) S->pass->unfinished -= 1; }
        cv.notify_all();
    }
}

// ---- phases 1 and 2 of one device pass over the records of `group` (consecutive batches; the packer thread) ---------------------------
void NpBatchPipeline::Impl::pack(const std::vector<Slot*>& group, int dev)
{
    int n_all = 0;
    for (size_t gi = 0; gi < group.size(); ++gi) {
        Slot& S = *group[gi];
        const int nb = (int)S.rec.size();
        S.pass = NULL;
        S.own_seq.assign(nb, std::string()); S.ref.assign(nb, RefView()); S.ref_start.assign(nb, 0); S.dev_index.assign(nb, -1);
        S.status.assign(nb, NP_BATCH_OK);
        S.n_dev = 0;
        n_all += nb;
    }
    if (n_all == 0) return;

    // the strand's models as load_from_raw and calculate_methylation_for_read choose them for a DNA read (squiggle_read.cpp:197-218,
    // basemods.cpp:276-287): strand "template", k = 6 from the base model.  Reads the device pass is not built for (RNA: another kit,
    // k = 5, another detector) take the caller's host path; a kit without a motif model leaves every map empty, as the reference.
    const char* strand_name = "template";
    const uint32_t k = 6;
    const PoreModel* pm_nuc = PoreModelSet::has_model(kit, "nucleotide", strand_name, k) ? PoreModelSet::get_model(kit, "nucleotide", strand_name, k) : NULL;
    const bool have_meth = pm_nuc && PoreModelSet::has_model(kit, params.methylation_type, strand_name, k);
    const int alphabet = np_alphabet_id(params.methylation_type.c_str());
    const bool device_ok = pm_nuc && pm_nuc->k == k && alphabet >= 1 && alphabet <= 4;
    const int MINSEP = params.min_separation, FLANK = params.min_flank;

    // ---- phase 1a: which records go to the device, their reference segments and sizes ---------------------------------------
    double tm0 = now();
    struct Rec { Slot* S; int i; };
    std::vector<Rec> idx;                         // device order -> (batch, index in the batch)
    for (size_t gi = 0; gi < group.size(); ++gi) {
        Slot& S = *group[gi];
        const std::vector<NpBatchRead>& reads = S.rec;
        for (int i = 0; i < (int)reads.size(); ++i) {
            const bool fits = device_ok && !reads[i].rna && reads[i].record && reads[i].read_sequence && reads[i].read_sequence->size() >= k &&
                              (reads[i].raw_pa || reads[i].raw_adc) && reads[i].n_raw >= 64;
            if (!fits) { S.status[i] = NP_BATCH_HOST_PATH; continue; }
            if (!have_meth) continue;                                            // an empty map (basemods.cpp:280-287)
            S.dev_index[i] = (int)idx.size(); idx.push_back(Rec{&S, i}); S.n_dev += 1;
        }
    }
    const int n = (int)idx.size();
    if (n == 0) return;
    // the pass's buffers: one of the context's three that is off the device and has handed all its batches' maps over
    Pass* Pp = NULL;
    {
        std::unique_lock<std::mutex> g(m);
        for (;;) {
            for (int q = 0; q < 3 && !Pp; ++q) { Pass* C = passes[3 * dev + (int)((n_passes + q) % 3)]; if (!C->on_device && C->unfinished == 0) Pp = C; }
            if (Pp) break;
            if (stop) return;                         // (the destructor waits for every batch first; a failing run must not park the packer here)
            (void)device_passes(devs[dev]->ordinal);
            cv.wait_for(g, std::chrono::microseconds(200));
        }
        for (size_t gi = 0; gi < group.size(); ++gi) if (group[gi]->n_dev > 0) { group[gi]->pass = Pp; Pp->unfinished += 1; }
    }
    Pass& P = *Pp;
    n_passes += 1;
    DevState& D = *devs[P.dev];
    np_ctx* c = D.c;
    P.union_seq.clear(); P.union_lo.clear();
    auto rd = [&](int q) -> const NpBatchRead& { return idx[q].S->rec[idx[q].i]; };
    // The reference fetches every record's segment on its own, under a critical section (get_reference_region_ts,
    // src/alignment/nanopolish_eventalign.cpp:207-221: faidx_fetch_seq is not thread-safe) -- serial work per record.  The records
    // of a BamProcessor batch come from a sorted BAM: when the batch's records on one contig cover a compact stretch, ONE fetch of
    // their union serves them all (faidx clips a range to the contig, so a slice of the clipped union is what the clipped
    // per-record fetch returns) and every record's segment is a VIEW into it -- nothing is copied per record; a scattered batch
    // keeps the per-record fetches.
    std::map<int, std::pair<int, int> > span;           // tid -> [lowest pos, highest end] of the batch's records
    std::map<int, int64_t> covered;
    bool all_adc = true;
    for (int q = 0; q < n; ++q) {
        const bam1_t* record = rd(q).record;
        const int lo = record->core.pos, hi = bam_endpos(record);
        idx[q].S->ref_start[idx[q].i] = lo;
        std::map<int, std::pair<int, int> >::iterator it = span.find(record->core.tid);
        if (it == span.end()) span[record->core.tid] = std::make_pair(lo, hi);
        else { it->second.first = std::min(it->second.first, lo); it->second.second = std::max(it->second.second, hi); }
        covered[record->core.tid] += hi - lo + 1;
        all_adc = all_adc && rd(q).raw_adc != NULL;
    }
    for (std::map<int, std::pair<int, int> >::const_iterator it = span.begin(); it != span.end(); ++it) {
        const int64_t len = (int64_t)it->second.second - it->second.first + 1;
        if (len <= (64 << 20) && len <= 4 * covered[it->first] + (1 << 20)) {
            int lo = it->second.first, hi = it->second.second;
            std::map<int, RefStretch>::iterator ce = ref_cache.find(it->first);
            if (ce != ref_cache.end() && ce->second.lo <= lo && hi <= ce->second.hi) {          // inside the cached stretch: nothing to fetch
                P.union_seq[it->first] = ce->second.seq; P.union_lo[it->first] = ce->second.lo;
            } else {
                if (ce != ref_cache.end() && std::max(hi, ce->second.hi) - (int64_t)std::min(lo, ce->second.lo) + 1 <= (64 << 20) &&
                    lo <= ce->second.hi + 1 && ce->second.lo <= hi + 1) { lo = std::min(lo, ce->second.lo); hi = std::max(hi, ce->second.hi); }
                int fetched_len = 0;
                RefStretch st;
                st.lo = lo; st.hi = hi;
                st.seq = std::make_shared<const std::string>(get_reference_region_ts(fai, hdr->target_name[it->first], lo, hi, &fetched_len));
                size_t held = 0;
                for (std::map<int, RefStretch>::const_iterator q = ref_cache.begin(); q != ref_cache.end(); ++q) held += q->second.seq->size();
                if (held + st.seq->size() > ((size_t)512 << 20)) ref_cache.clear();              // (a bound on what the cache keeps alive)
                ref_cache[it->first] = st;
                P.union_seq[it->first] = st.seq; P.union_lo[it->first] = lo;
            }
        }
    }
    for (int q = 0; q < n; ++q) {                          // records no union serves: the reference's own per-record fetch (serial: faidx)
        Slot& S = *idx[q].S; const int i = idx[q].i;
        const bam1_t* record = rd(q).record;
        if (P.union_seq.find(record->core.tid) != P.union_seq.end()) continue;
        int fetched_len = 0;
        S.own_seq[i] = get_reference_region_ts(fai, hdr->target_name[record->core.tid], record->core.pos, bam_endpos(record), &fetched_len);   // :258-270
    }
    // Alphabet::disambiguate (upper-casing + IUPAC codes -> their first base) is the identity on an upper-case ACGT string, and it
    // builds one std::string per character: 0.3 ms of a host core per 5 kb read.  Only a segment that needs it gets it (as its own copy).
    pool->run(n, 32, [&](int q) {
        Slot& S = *idx[q].S; const int i = idx[q].i;
        const bam1_t* record = rd(q).record;
        std::map<int, std::shared_ptr<const std::string> >::const_iterator u = P.union_seq.find(record->core.tid);
        RefView v;
        if (u != P.union_seq.end()) {
            const int64_t off = (int64_t)record->core.pos - P.union_lo.find(record->core.tid)->second, want = (int64_t)bam_endpos(record) - record->core.pos + 1;
            const int64_t have = (int64_t)u->second->size() - off;
            if (have > 0) { v.p = u->second->data() + off; v.n = (size_t)std::min(want, have); }
        } else { v.p = S.own_seq[i].data(); v.n = S.own_seq[i].size(); }
        bool plain = true;
        for (size_t t = 0; t < v.n; ++t) if (base_code(v.p[t]) > 3) { plain = false; break; }
        if (!plain) {
            S.own_seq[i] = gDNAAlphabet.disambiguate(std::string(v.p, v.n));
            v.p = S.own_seq[i].data(); v.n = S.own_seq[i].size();
        }
        S.ref[i] = v;
    });
    std::vector<int64_t> raw_off(n + 1, 0), event_off(n + 1, 0), rank_off(n + 1, 0), cigar_off(n + 1, 0), jr_off(n + 1, 0),
                         pair_off(n + 1, 0), genome_off(n + 1, 0);
    std::vector<int64_t>& group_off = P.group_off;
    group_off.assign(n + 1, 0);
    for (int q = 0; q < n; ++q) {
        const bam1_t* record = rd(q).record;
        const int64_t n_raw = (int64_t)rd(q).n_raw, L = (int64_t)rd(q).read_sequence->size(), ln = (int64_t)idx[q].S->ref[idx[q].i].n;
        const int64_t ecap = n_raw / g_event_cap_divisor + 2, nk = L - k + 1, gcap = ln / (MINSEP + 1) + 2;
        raw_off[q + 1] = raw_off[q] + n_raw;
        event_off[q + 1] = event_off[q] + ecap;
        rank_off[q + 1] = rank_off[q] + nk;
        cigar_off[q + 1] = cigar_off[q] + record->core.n_cigar;
        genome_off[q + 1] = genome_off[q] + ln;
        group_off[q + 1] = group_off[q] + gcap;
        jr_off[q + 1] = jr_off[q] + 2 * (ln + (2 * FLANK + 1) * gcap);
        pair_off[q + 1] = pair_off[q] + ecap + nk + 2;
    }
    const int64_t n_slots = group_off[n], n_jobs = 2 * n_slots, n_ev = event_off[n], n_rk = rank_off[n];
    int64_t max_samples = 1, max_events = 1, max_bands = 1;
    for (int q = 0; q < n; ++q) {
        max_samples = std::max(max_samples, raw_off[q + 1] - raw_off[q]);
        max_events = std::max(max_events, event_off[q + 1] - event_off[q]);
        max_bands = std::max(max_bands, pair_off[q + 1] - pair_off[q]);
    }

    // ---- layouts ----------------------------------------------------------------------------------------------------------
    Layout li;
    const size_t i_raw = li.add((size_t)raw_off[n] * (all_adc ? sizeof(int16_t) : sizeof(float))), i_adc_offset = li.add((size_t)n * 4),
                 i_adc_unit = li.add((size_t)n * 4), i_ranks = li.add((size_t)n_rk * sizeof(uint16_t)),
                 i_reads_a = li.add((size_t)n * sizeof(np_read_dev)), i_reads_b = li.add((size_t)n * sizeof(np_read_dev)),
                 i_genome = li.add((size_t)genome_off[n]), i_raw_off = li.add((size_t)(n + 1) * 8), i_event_off = li.add((size_t)(n + 1) * 8),
                 i_cigar_off = li.add((size_t)(n + 1) * 8), i_group_off = li.add((size_t)(n + 1) * 8), i_jr_off = li.add((size_t)(n + 1) * 8),
                 i_pair_off = li.add((size_t)(n + 1) * 8), i_ref_begin = li.add((size_t)n * 8), i_ref_len = li.add((size_t)n * 4),
                 i_read_len = li.add((size_t)n * 4), i_cigar = li.add((size_t)cigar_off[n] * 4), i_rc = li.add((size_t)n);
    Layout lo;
    P.o_scores = lo.add((size_t)n_jobs * sizeof(float)); P.o_first = lo.add((size_t)n_slots * 4); P.o_last = lo.add((size_t)n_slots * 4);
    P.o_n_motif = lo.add((size_t)n_slots * 4); P.o_n_groups = lo.add((size_t)n * 4); P.o_n_events = lo.add((size_t)n * 4);
    P.o_n_pairs = lo.add((size_t)n * 4); P.o_calibrated = lo.add((size_t)n * 4);
    P.out_bytes = lo.size;
    Layout ls;       // the part of the scratch that must start a batch zeroed comes first
    const size_t s_pair_begin = ls.add((size_t)n * 4), s_deg = ls.add((size_t)n * 8), s_kpos = ls.add((size_t)n_jobs * 8),
                 s_epb = ls.add((size_t)n * 8), s_jobs = ls.add((size_t)n_jobs * sizeof(np_hmm_job_dev));
    const size_t zero_bytes = ls.size;
    const size_t s_raw_pa = ls.add(all_adc ? (size_t)raw_off[n] * sizeof(float) : 0);
    const size_t s_tstat = ls.add((size_t)(2 * raw_off[n] + 16) * sizeof(float)), s_ev_len = ls.add((size_t)n_ev * 4), s_ev_mean = ls.add((size_t)n_ev * 4),
                 s_ev_stdv = ls.add((size_t)n_ev * 4), s_ev_start = ls.add((size_t)n_ev * 4), s_map_start = ls.add((size_t)n_rk * 4),
                 s_pairs = ls.add((size_t)pair_off[n] * sizeof(np_pair)),
                 s_job_ranks = ls.add((size_t)jr_off[n] * sizeof(uint16_t));
    add_time(0, now() - tm0); tm0 = now();
    // The first MERGED pass sizes every buffer of the pipeline -- all passes, every context's scratch -- for a full pass (coalesce_records
    // records like these): the groups the packer forms vary in size, and growing pinned / device allocations step by step, pass buffer by
    // pass buffer, costs more than the passes themselves (seconds of hipHostMalloc spread over the first hundred batches).
    if (group.size() > 1 && !presized) {
        presized = true;
        const double full = n_all < coalesce_records ? std::min(32.0, (double)coalesce_records / (double)n_all) : 1.0;
        for (size_t i = 0; i < passes.size(); ++i) {
            Pass& Q = *passes[i];
            if (Q.on_device || Q.unfinished > 0) continue;     // (in use: it grows when its turn comes)
            np_ctx* qc = devs[Q.dev]->c;
            Q.in.reserve(qc, (size_t)((double)li.size * full) + 256); Q.out.reserve(qc, (size_t)((double)lo.size * full) + 256);
        }
        for (size_t d = 0; d < devs.size(); ++d) {
            if ((size_t)((double)ls.size * full) + 256 <= devs[d]->scratch.cap) continue;
            check(np_sync(devs[d]->c, NULL), "np_sync");
            devs[d]->scratch.reserve(devs[d]->c, (size_t)((double)ls.size * full) + 256);
        }
    }
    P.in.reserve(c, li.size + 256); P.out.reserve(c, lo.size + 256);
    if (ls.size + 256 > D.scratch.cap) {
        check(np_sync(c, NULL), "np_sync");                  // the batch in flight on this device still computes in the scratch that is about to be replaced
        D.scratch.reserve(c, ls.size + 256);
    }
    add_time(5, now() - tm0); tm0 = now();

    // ---- phase 1b: pack the pinned input blob -----------------------------------------------------------------------------
    char* H = P.in.h;
    float* h_raw = (float*)(H + i_raw); uint16_t* h_ranks = (uint16_t*)(H + i_ranks);
    np_read_dev* h_reads_a = (np_read_dev*)(H + i_reads_a); np_read_dev* h_reads_b = (np_read_dev*)(H + i_reads_b);
    char* h_genome = H + i_genome; int64_t* h_ref_begin = (int64_t*)(H + i_ref_begin); int32_t* h_ref_len = (int32_t*)(H + i_ref_len);
    int32_t* h_read_len = (int32_t*)(H + i_read_len); uint32_t* h_cigar = (uint32_t*)(H + i_cigar); uint8_t* h_rc = (uint8_t*)(H + i_rc);
    pool->run(n, 4, [&](int q) {
        const NpBatchRead& R = rd(q);
        const RefView& ref = idx[q].S->ref[idx[q].i];
        const bam1_t* record = R.record;
        const std::string& seq = *R.read_sequence;
        for (int t = 0; t < 2; ++t)
            np_fill_read_host(t ? &h_reads_b[q] : &h_reads_a[q], 0.0, 1.0, 1.0, event_off[q], (uint32_t)(event_off[q + 1] - event_off[q]), rank_off[q],
                              (uint32_t)(rank_off[q + 1] - rank_off[q]));
        ((float*)(H + i_adc_offset))[q] = R.adc_offset; ((float*)(H + i_adc_unit))[q] = R.adc_raw_unit;
        if (all_adc) memcpy((int16_t*)(H + i_raw) + raw_off[q], R.raw_adc, R.n_raw * sizeof(int16_t));
        else if (R.raw_pa) memcpy(h_raw + raw_off[q], R.raw_pa, R.n_raw * sizeof(float));
        else for (size_t t = 0; t < R.n_raw; ++t)            // the loader's conversion, fp32 (fast5_loader.cpp:96-103)
            h_raw[raw_off[q] + t] = ((float)R.raw_adc[t] + R.adc_offset) * R.adc_raw_unit;
        nucleotide_kmer_ranks(seq, k, h_ranks + rank_off[q]);
        memcpy(h_genome + genome_off[q], ref.p, ref.n);
        h_ref_begin[q] = genome_off[q]; h_ref_len[q] = (int32_t)ref.n;
        memcpy(h_cigar + cigar_off[q], bam_get_cigar(record), 4 * (size_t)record->core.n_cigar);
        h_read_len[q] = (int32_t)seq.size();
        h_rc[q] = bam_is_rev(record) ? 1 : 0;
    }, true, pack_first);
    memcpy(H + i_raw_off, raw_off.data(), (size_t)(n + 1) * 8); memcpy(H + i_event_off, event_off.data(), (size_t)(n + 1) * 8);
    memcpy(H + i_cigar_off, cigar_off.data(), (size_t)(n + 1) * 8); memcpy(H + i_group_off, group_off.data(), (size_t)(n + 1) * 8);
    memcpy(H + i_jr_off, jr_off.data(), (size_t)(n + 1) * 8); memcpy(H + i_pair_off, pair_off.data(), (size_t)(n + 1) * 8);

    add_time(1, now() - tm0); tm0 = now();
    // ---- phase 2: one upload, the batch on the device, one read-back -----------------------------------------------------------
    const int m_nuc = shim().model_id(pm_nuc, D.slot_key);
    const int m_meth = shim().model_id(PoreModelSet::get_model(kit, params.methylation_type, strand_name, k), D.slot_key);
    check(np_copy_to_device(c, D.s_h2d, P.in.d, P.in.h, li.size), "np_copy_to_device");
    check(np_event_record(c, P.ev_h2d, D.s_h2d), "np_event_record");
    check(np_stream_wait_event(c, NULL, P.ev_h2d), "np_stream_wait_event");
    check(np_memset_dev(c, NULL, P.out.d, 0, lo.size), "np_memset_dev");
    check(np_memset_dev(c, NULL, D.scratch.d, 0, zero_bytes), "np_memset_dev");
    {
        char* Dv = P.in.d; char* O = P.out.d; char* X = D.scratch.d;
        float* raw = all_adc ? (float*)(X + s_raw_pa) : (float*)(Dv + i_raw); uint16_t* ranks = (uint16_t*)(Dv + i_ranks);
        np_read_dev* reads_a = (np_read_dev*)(Dv + i_reads_a); np_read_dev* reads_b = (np_read_dev*)(Dv + i_reads_b);
        int64_t *d_raw_off = (int64_t*)(Dv + i_raw_off), *d_event_off = (int64_t*)(Dv + i_event_off), *d_cigar_off = (int64_t*)(Dv + i_cigar_off),
                *d_group_off = (int64_t*)(Dv + i_group_off), *d_jr_off = (int64_t*)(Dv + i_jr_off), *d_pair_off = (int64_t*)(Dv + i_pair_off),
                *ref_begin = (int64_t*)(Dv + i_ref_begin);
        int32_t *ref_len = (int32_t*)(Dv + i_ref_len), *read_len = (int32_t*)(Dv + i_read_len);
        uint32_t* cigar = (uint32_t*)(Dv + i_cigar); uint8_t* rc = (uint8_t*)(Dv + i_rc); char* genome = Dv + i_genome;
        float* scores = (float*)(O + P.o_scores);
        int32_t *first = (int32_t*)(O + P.o_first), *last = (int32_t*)(O + P.o_last), *n_motif = (int32_t*)(O + P.o_n_motif),
                *n_groups = (int32_t*)(O + P.o_n_groups), *n_events = (int32_t*)(O + P.o_n_events), *n_pairs = (int32_t*)(O + P.o_n_pairs),
                *calibrated = (int32_t*)(O + P.o_calibrated);
        int32_t *pair_begin = (int32_t*)(X + s_pair_begin), *deg = (int32_t*)(X + s_deg), *kpos = (int32_t*)(X + s_kpos),
                *map_start = (int32_t*)(X + s_map_start);
        double* epb = (double*)(X + s_epb);
        np_hmm_job_dev* jobs = (np_hmm_job_dev*)(X + s_jobs);
        float *tstat = (float*)(X + s_tstat), *ev_len = (float*)(X + s_ev_len), *ev_mean = (float*)(X + s_ev_mean), *ev_stdv = (float*)(X + s_ev_stdv);
        uint32_t* ev_start = (uint32_t*)(X + s_ev_start);
        np_pair* pairs = (np_pair*)(X + s_pairs);
        uint16_t* job_ranks = (uint16_t*)(X + s_job_ranks);
        np_detector_param prm;
        np_event_detection_params(&prm, 0);
        // the slot layout of this pass's work items: the kernels from the builder to the scorer visit live items only (np_set_job_layout)
        check(np_set_job_layout(c, n, d_group_off, n_groups, n_slots), "np_set_job_layout");
        check(np_cm_build_jobs_cigar_dev(c, NULL, n, genome, ref_begin, ref_len, cigar, d_cigar_off, cigar_off[n], read_len, rc, alphabet, k, MINSEP,
                                         FLANK, d_group_off, n_slots, d_jr_off, jobs, kpos, job_ranks, first, last, n_motif, n_groups, deg),
              "np_cm_build_jobs_cigar_dev");
        // counts in, events out: with the DNA windows the long reads never exist as pA values (np_detect_events_adc_dev; `raw` is its scratch for
        // the short and the serial-path reads)
        if (all_adc)
            check(np_detect_events_adc_dev(c, NULL, n, (const int16_t*)(Dv + i_raw), d_raw_off, max_samples, (const float*)(Dv + i_adc_offset),
                                           (const float*)(Dv + i_adc_unit), raw, &prm, tstat, d_event_off, max_events, ev_start, ev_len, ev_mean,
                                           ev_stdv, n_events), "np_detect_events_adc_dev");
        else
            check(np_detect_events_dev(c, NULL, n, raw, d_raw_off, max_samples, &prm, tstat, d_event_off, max_events, ev_start, ev_len, ev_mean,
                                       ev_stdv, n_events), "np_detect_events_dev");
        check(np_mom_fill_dev(c, NULL, n, reads_a, reads_b, ev_mean, n_events, ranks, m_nuc), "np_mom_fill_dev");
        check(np_event_align_dev(c, NULL, n, reads_a, ev_mean, ranks, m_nuc, max_bands, d_pair_off, pairs, pair_begin, n_pairs), "np_event_align_dev");
        check(np_calibrate_resolve_dev(c, NULL, n, reads_b, ev_mean, ranks, m_nuc, d_pair_off, pairs, pair_begin, n_pairs, map_start, NULL /* .stop: not read on this path */, epb,
                                       calibrated, n_jobs, jobs, kpos), "np_calibrate_resolve_dev");
        check(np_cm_discard_degenerate_dev(c, NULL, reads_b, map_start, deg, n_jobs, jobs), "np_cm_discard_degenerate_dev");
        check(np_hmm_score_dev(c, NULL, n_jobs, jobs, reads_b, ev_mean, job_ranks, m_meth, scores), "np_hmm_score_dev");
        check(np_set_job_layout(c, 0, NULL, NULL, 0), "np_set_job_layout");
    }
    check(np_event_record(c, P.ev_cmp, NULL), "np_event_record");
    check(np_stream_wait_event(c, D.s_d2h, P.ev_cmp), "np_stream_wait_event");
    check(np_copy_to_host(c, D.s_d2h, P.out.h, P.out.d, lo.size), "np_copy_to_host");
    check(np_event_record(c, P.ev_d2h, D.s_d2h), "np_event_record");
    P.on_device = true;
    add_time(2, now() - tm0);
}

// ---- phase 3 of one batch (the finisher thread): wait for the read-back, build the ScoredSite maps (basemods.cpp:384-413) -----------
void NpBatchPipeline::Impl::finish(Slot& S)
{
    std::vector<NpBatchRead>& reads = S.rec;
    const int n = (int)reads.size();
    S.built.clear(); S.built.resize(n); S.builder.assign(n, -1);
    if (n == 0) return;
    double tm0 = now();
    if (S.pass) check(np_event_sync(devs[S.pass->dev]->c, S.pass->ev_d2h), "np_event_sync");      // (the pass's other batches wait on the same event)
    add_time(3, now() - tm0); tm0 = now();
    if (!S.pass) return;
    const Pass& P = *S.pass;
    const char* O = P.out.h;
    const float* scores = (const float*)(O + P.o_scores);
    const int32_t *first = (const int32_t*)(O + P.o_first), *last = (const int32_t*)(O + P.o_last), *n_motif = (const int32_t*)(O + P.o_n_motif),
                  *n_groups = (const int32_t*)(O + P.o_n_groups), *n_events = (const int32_t*)(O + P.o_n_events),
                  *n_pairs = (const int32_t*)(O + P.o_n_pairs), *calibrated = (const int32_t*)(O + P.o_calibrated);
    const uint32_t k = 6;
    pool->run(n, 8, [&](int i) {
        const bam1_t* record = reads[i].record;
        S.builder[i] = pool->current_worker();
        if (S.status[i] == NP_BATCH_HOST_PATH) return;                    // decided in phase 1: the caller's per-record function fills its map
        std::map<int, ScoredSite>& site_score_map = S.built[i];
        const int q = S.dev_index[i];
        if (q < 0) return;                                                // no motif model for the kit: the map stays empty
        if (n_events[q] < 0 || n_groups[q] < 0) { S.status[i] = NP_BATCH_HOST_PATH; return; }   // NP_ED_INEXACT / NP_ED_OVERFLOW / capacity
        if (n_pairs[q] <= 0 || !calibrated[q]) { S.status[i] = NP_BATCH_NO_EVENTS; return; }
        const std::string contig = hdr->target_name[record->core.tid];
        const RefView ref_seq = S.ref[i];
        const int strand_idx = 0;
        for (int g = 0; g < n_groups[q]; ++g) {
            const int64_t slot = P.group_off[q] + g;
            const float unmethylated_score = scores[2 * slot], methylated_score = scores[2 * slot + 1];
            if (unmethylated_score != unmethylated_score || methylated_score != methylated_score) continue;   // a group the caller rules skip
            const int start_position = first[slot] + S.ref_start[i];
            const int end_position = last[slot] + S.ref_start[i];
            if ((region_start != -1 && start_position < region_start) || (region_end != -1 && end_position >= region_end)) continue;
            // groups come in ascending reference order: a new site goes to the END of the map, constructed in place (the reference's
            // find + copy-insert costs two tree descents, a temporary ScoredSite and a second copy of its strings); a position that
            // is already there -- the second strand of a 2D read in the reference's loop, never on this path -- takes the general route
            std::map<int, ScoredSite>::iterator iter;
            if (site_score_map.empty() || site_score_map.rbegin()->first < start_position) {
                iter = site_score_map.emplace_hint(site_score_map.end(), std::piecewise_construct, std::forward_as_tuple(start_position), std::forward_as_tuple());
                ScoredSite& ss = iter->second;
                ss.chromosome = contig;
                ss.start_position = start_position;
                ss.end_position = end_position;
                ss.n_motif = n_motif[slot];
                const size_t site_output_start = first[slot] - k + 1, site_output_end = last[slot] + k;
                if (site_output_start < ref_seq.n) ss.sequence.assign(ref_seq.p + site_output_start, std::min(site_output_end - site_output_start, ref_seq.n - site_output_start));
            } else {
                iter = site_score_map.find(start_position);
                if (iter == site_score_map.end()) {
                    ScoredSite ss;
                    ss.chromosome = contig;
                    ss.start_position = start_position;
                    ss.end_position = end_position;
                    ss.n_motif = n_motif[slot];
                    const size_t site_output_start = first[slot] - k + 1, site_output_end = last[slot] + k;
                    if (site_output_start < ref_seq.n) ss.sequence.assign(ref_seq.p + site_output_start, std::min(site_output_end - site_output_start, ref_seq.n - site_output_start));
                    iter = site_score_map.insert(std::make_pair(start_position, ss)).first;
                }
            }
            iter->second.ll_unmethylated[strand_idx] = unmethylated_score;
            iter->second.ll_methylated[strand_idx] = methylated_score;
            iter->second.strands_scored += 1;
        }
    }, false /* on the workers only: see Impl::builder_of */, maps_first);
    add_time(4, now() - tm0);
}

bool NpBatchPipeline::collect(MethylationCallingResult& result)
{
    int pieces;
    {
        std::lock_guard<std::mutex> g(p->m);
        if (p->batch_pieces.empty()) return false;
        pieces = p->batch_pieces.front();
    }
    for (int q = 0; q < pieces; ++q) {                              // the oldest batch's pieces, in order
        Slot* Sp;
        const double t0 = now();
        {
            std::unique_lock<std::mutex> g(p->m);
            Sp = p->slots[p->n_collected % (long)p->slots.size()];
            while (!Sp->finished) p->cv.wait(g);                    // (the two finishers may end out of order: collect() keeps submission order)
        }
        p->add_time(6, now() - t0);
        Slot& S = *Sp;
        std::vector<NpBatchRead>& reads = *S.caller;
        const int n = (int)S.rec.size();
        for (int i = 0; i < n; ++i) {
            reads[S.first + i].status = S.status[i];
            if (S.status[i] == NP_BATCH_HOST_PATH) continue;                 // the caller's per-record function fills (and creates) its map
            result[S.rec[i].record].swap(S.built[i]);                         // the (possibly empty) map of the record, basemods.cpp:253-256
            if (p->track_builders) p->builder_of[S.rec[i].record] = S.builder[i];
        }
        S.built.clear();
        { std::lock_guard<std::mutex> g(p->m); S.finished = false; p->n_collected += 1; }
        p->cv.notify_all();
    }
    { std::lock_guard<std::mutex> g(p->m); p->batch_pieces.pop_front(); }
    return true;
}

void NpBatchPipeline::recycle(MethylationCallingResult& result)
{
    typedef std::vector<std::map<int, ScoredSite> > Maps;
    const int W = p->pool->threads();
    std::vector<std::shared_ptr<Maps> > mine(W + 1);
    for (MethylationCallingResult::iterator it = result.begin(); it != result.end(); ++it) {
        // (the entry goes whether or not the map is empty: a stale one would route a later host-path map of a reused bam1_t* to the wrong worker)
        std::map<const bam1_t*, int>::iterator b = p->builder_of.find(it->first);
        const int w = b != p->builder_of.end() && b->second >= 0 && b->second < W ? b->second : W;       // W: built elsewhere (the caller's host path)
        if (b != p->builder_of.end()) p->builder_of.erase(b);
        if (it->second.empty()) continue;
        if (!mine[w]) mine[w] = std::make_shared<Maps>();
        mine[w]->push_back(std::map<int, ScoredSite>());
        mine[w]->back().swap(it->second);
    }
    for (int w = 0; w < W; ++w)
        if (mine[w]) { std::shared_ptr<Maps> g = mine[w]; p->pool->post_to(w, [g]() { g->clear(); }); }
    if (mine[W]) { std::shared_ptr<Maps> g = mine[W]; p->pool->post((int)g->size(), 32, [g](int i) { std::map<int, ScoredSite>().swap((*g)[i]); }, [g]() {}); }
}

extern "C" void np_batch_set_event_capacity_divisor(int divisor) { g_event_cap_divisor = divisor >= 2 ? divisor : 2; }

// The synchronous form: one submit + collect on a process-wide pipeline, so that buffers, streams and the registered models
// persist from batch to batch.
void np_calculate_methylation_for_batch(MethylationCallingResult& result, std::vector<NpBatchRead>& reads,
                                        const MethylationCallingParameters& params, const std::string& kit,
                                        const faidx_t* fai, const bam_hdr_t* hdr, int region_start, int region_end)
{
    static std::mutex lock;
    static NpBatchPipeline* pipe = NULL;
    std::lock_guard<std::mutex> g(lock);
    if (!pipe) pipe = new NpBatchPipeline(NpBatchPipeline::synchronous_t(), params, kit, fai, hdr, region_start, region_end);
    else pipe->configure(params, kit, fai, hdr, region_start, region_end);
    pipe->submit(reads);
    pipe->collect(result);
}
