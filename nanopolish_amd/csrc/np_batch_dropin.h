// np_batch_dropin.h -- the batched reference-side bindings (see np_batch_dropin.cpp): call-methylation's per-record work for whole
// BamProcessor batches, as a synchronous call and as a double-buffered pipeline.
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

// The production feed: the same pass, double-buffered.  Two batches can be in flight: while the device works on batch k (one
// upload, the kernels, one read-back, on three HIP streams ordered by events) the caller reads batch k+1 from the BAM / signal
// files and submits it; collect() returns batches in submission order.  All device and pinned-host buffers persist and only
// grow.  BamProcessor's loop (bam_processor.cpp:90-119) becomes
//     while (read a batch into recs[k & 1]) { pipe.submit(recs[k & 1]); if (k > 0) pipe.collect(result_of(k - 1)); ++k; }
//     pipe.collect(result_of(k - 1));
// (INTEGRATION.md section 2).  The read vector, the records and the buffers its entries point to must stay alive and unchanged
// until the batch has been collected.
class NpBatchPipeline {
public:
    NpBatchPipeline(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                    const bam_hdr_t* hdr, int region_start, int region_end);
    ~NpBatchPipeline();
    void configure(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                   const bam_hdr_t* hdr, int region_start, int region_end);        // only with nothing in flight
    void submit(std::vector<NpBatchRead>& reads);              // at most two batches in flight (exits with a message otherwise)
    bool collect(MethylationCallingResult& result);           // the oldest batch in flight; false if there is none
    int in_flight() const;
    // host wall-clock seconds spent in the phases since construction (diagnostics; tests/bench_batch_dropin.py prints them):
    // [0] phase 1a reference fetch + sizes, [1] phase 1b packing the pinned blob, [2] enqueueing copies and kernels,
    // [3] collect: waiting for the device, [4] phase 3 ScoredSite maps, [5] buffer growth (allocation)
    void host_seconds(double out[6]) const;
    struct Impl;
private:
    Impl* p;
    NpBatchPipeline(const NpBatchPipeline&);
    NpBatchPipeline& operator=(const NpBatchPipeline&);
};

// Test knob: the per-read event capacity of the device detector is n_raw / divisor + 2 (default 2: boundaries of one detector are
// at least two samples apart, so the default can never be exceeded).  A larger divisor drives the overflow -> NP_BATCH_HOST_PATH
// route in tests/test_gpu_batch_dropin.py.
extern "C" void np_batch_set_event_capacity_divisor(int divisor);
