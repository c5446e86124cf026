// np_batch_dropin.h -- the batched reference-side bindings (see np_batch_dropin.cpp): call-methylation's per-record work for whole
// BamProcessor batches, as a synchronous call and as a pipeline with its own host worker threads over one or several GPUs.
#pragma once
#include <string>
#include <vector>
#include "htslib/faidx.h"
#include "htslib/sam.h"
#include "nanopolish_basemods.h"

// One record of a BamProcessor batch (src/common/nanopolish_bam_processor.cpp:90-119) with what
// calculate_methylation_for_read_from_bam (src/nanopolish_call_methylation.cpp:163-177) loads for it through SquiggleRead:
struct NpBatchRead {
    const bam1_t* record = NULL;
    const std::string* read_sequence = NULL;   // SquiggleRead::read_sequence (ReadDB::get_read_sequence)
    const float* raw_pa = NULL;                // the read's raw table in pA, as load_from_raw hands it to detect_events
    size_t n_raw = 0;
    // alternatively the samples as the sequencer stored them, with the channel's conversion (src/io/nanopolish_fast5_loader.cpp:96-103:
    // pA = ((float)adc + offset) * raw_unit, raw_unit = range / digitisation, all in fp32): half the bytes to pack and to upload.
    // A batch whose records ALL carry raw_adc is converted on the device (np_adc_to_pa_dev, the same fp32 expression); otherwise a
    // record without raw_pa is converted on the host while packing.
    const int16_t* raw_adc = NULL;
    float adc_offset = 0.0f, adc_raw_unit = 1.0f;
    int rna = 0;                               // SquiggleRead::nucleotide_type == SRNT_RNA (squiggle_read.cpp:195): such a read uses the
                                               // r9.4_70bps 5-mer models and the RNA detector; the device pass is built for what
                                               // load_from_raw hard-codes for DNA (kit r9.4_450bps, "template", k = 6, :197-202), so an
                                               // RNA read comes back NP_BATCH_HOST_PATH
    int status = 0;                            // out: NP_BATCH_*
};
#define NP_BATCH_OK 0
#define NP_BATCH_NO_EVENTS 1     // the read has no usable event alignment (failed QC / calibration): an empty site map, as the reference
#define NP_BATCH_HOST_PATH 2     // not processed on the device (RNA read, a non-finite sample in the signal, or a
                                 // per-read capacity estimate was exceeded): the caller runs its per-record function on it

// Fills result[record] (one map per record, created even when empty, like basemods.cpp:253-256) for every read whose
// status comes back NP_BATCH_OK / NP_BATCH_NO_EVENTS.  kit: the pore-model kit of the DNA reads (load_from_raw: "r9.4_450bps").
// Synchronous: one submit + collect on a process-wide NpBatchPipeline (buffers persist from call to call).
void np_calculate_methylation_for_batch(MethylationCallingResult& result, std::vector<NpBatchRead>& reads,
                                        const MethylationCallingParameters& calling_parameters, const std::string& kit,
                                        const faidx_t* fai, const bam_hdr_t* hdr, int region_start, int region_end);

// The production feed: the same pass as a three-stage pipeline that does not borrow the caller's threads.
//
//     submit(batch k+1)  -> [pack: reference segments, k-mer ranks, samples into one pinned blob]      the pipeline's packer thread
//                        -> [device: one upload, the kernels, one read-back; three HIP streams]        + its worker pool
//                        -> [results: one std::map<int, ScoredSite> per record]                       the finisher thread + the pool
//     collect(batch k-1)    hands the finished maps over (a swap per record)
//
// submit() returns as soon as the batch is queued; collect() returns batches in submission order and blocks until the oldest one
// is finished.  With `devices` = {0, 1, ..., N-1} the pipeline owns one library context per listed GPU (the same device may be listed
// twice: two contexts on one GPU) and sends every device pass to the least loaded one.  A DEVICE PASS is one batch or -- round 5 --
// several consecutive ones: a pass costs the device the time of its longest read whatever it holds, so batches of BamProcessor's default
// 512 records that are already waiting are merged, up to NP_BATCH_COALESCE records (default 8 192) per pass; a pass starts as soon as
// that many records wait or a GPU has nothing to do, so a slow caller is never made to wait for company.  Results are per batch.
// `max_in_flight()` batches may be in flight: three per device when a batch fills a pass on its own, more when batches are small (up to
// NP_BATCH_SLOTS, default 3 x NP_BATCH_COALESCE / 512 = 48, per device); it can change after a submit() with the size of the batches -- ask again, as the loop below
// does; before the first submit() it returns the upper bound (the number of record / result vectors to rotate).  submit() with that
// many in flight is an error.  BamProcessor's loop (bam_processor.cpp:90-119) becomes
//     while (read a batch into recs[k % n]) { pipe.submit(recs[k % n]); if (pipe.in_flight() >= pipe.max_in_flight()) { pipe.collect(res); write(res); pipe.recycle(res); } ++k; }
//     while (pipe.collect(res)) { write(res); pipe.recycle(res); }
// (INTEGRATION.md section 2).  LIFETIME: the read vector, the records and every buffer its entries point to (sequence, samples) must
// stay alive and unchanged from submit() until the batch has been collected -- packing runs after submit() has returned.
class NpBatchPipeline {
public:
    NpBatchPipeline(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                    const bam_hdr_t* hdr, int region_start, int region_end);                     // one GPU: the process-wide context (NP_DEVICE)
    NpBatchPipeline(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                    const bam_hdr_t* hdr, int region_start, int region_end, const std::vector<int>& devices,
                    int host_threads = 0 /* worker threads; 0: the CPUs this process may use (affinity mask, cgroup quota), NP_HOST_THREADS */);
    ~NpBatchPipeline();
    void configure(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                   const bam_hdr_t* hdr, int region_start, int region_end);        // only with nothing in flight
    void submit(std::vector<NpBatchRead>& reads);
    bool collect(MethylationCallingResult& result);           // the oldest batch in flight; false if there is none
    // Takes the site maps of a batch the caller has written out back and destroys them on the pipeline's workers, off the caller's
    // thread (a batch of 8 192 reads holds 1.5 M ScoredSites, two heap blocks each: `results.clear()` on the writer's thread,
    // src/nanopolish_call_methylation.cpp:587, costs more than the device pass).  The outer map keeps its (now empty) entries.
    void recycle(MethylationCallingResult& result);
    int in_flight() const;
    int max_in_flight() const;
    int max_in_flight_for(size_t batch_records) const;        // max_in_flight() for batches of that size: how many record / result vectors to rotate
    int devices() const;
    // host wall-clock seconds spent in the phases since construction (diagnostics; tests/bench_batch_dropin.py prints them):
    // [0] phase 1a reference fetch + sizes, [1] phase 1b packing the pinned blob, [2] enqueueing copies and kernels,
    // [3] finisher waiting for the device, [4] phase 3 ScoredSite maps, [5] buffer growth (allocation), [6] collect() waiting for a
    // finished batch (the caller's thread), [7] submit() (the caller's thread)
    void host_seconds(double out[8]) const;
    struct Impl;
    // the pipeline behind np_calculate_methylation_for_batch: one batch in flight at a time, so ONE context (the process-wide one) and no
    // bookkeeping of which worker built which map (the caller of the synchronous form never recycles)
    struct synchronous_t {};
    NpBatchPipeline(synchronous_t, const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                    const bam_hdr_t* hdr, int region_start, int region_end);
private:
    Impl* p;
    NpBatchPipeline(const NpBatchPipeline&);
    NpBatchPipeline& operator=(const NpBatchPipeline&);
};

// Test knob: the per-read event capacity of the device detector is n_raw / divisor + 2 (default 2: boundaries of one detector are
// at least two samples apart, so the default can never be exceeded).  A larger divisor drives the overflow -> NP_BATCH_HOST_PATH
// route in tests/test_gpu_batch_dropin.py.
extern "C" void np_batch_set_event_capacity_divisor(int divisor);
