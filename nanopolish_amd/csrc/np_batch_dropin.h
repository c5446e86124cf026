// np_batch_dropin.h -- the batched reference-side binding of call-methylation's per-record work (see np_batch_dropin.cpp).
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
    int status = 0;                            // out: NP_BATCH_*
};
#define NP_BATCH_OK 0
#define NP_BATCH_NO_EVENTS 1     // the read has no usable event alignment (failed QC / calibration): an empty site map, as the reference
#define NP_BATCH_HOST_PATH 2     // not processed on the device (event detection not provably exact for this signal, or the
                                 // per-read capacity estimate was exceeded): the caller runs its per-record function on it

// Fills result[record] (one map per record, created even when empty, like basemods.cpp:253-256) for every read whose
// status comes back NP_BATCH_OK / NP_BATCH_NO_EVENTS.  kit: the pore-model kit of the reads (r9.4_450bps ...).
void np_calculate_methylation_for_batch(MethylationCallingResult& result, std::vector<NpBatchRead>& reads,
                                        const MethylationCallingParameters& calling_parameters, const std::string& kit,
                                        const faidx_t* fai, const bam_hdr_t* hdr, int region_start, int region_end);
