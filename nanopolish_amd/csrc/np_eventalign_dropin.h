// np_eventalign_dropin.h -- the batched reference-side binding of eventalign's per-record work (see np_eventalign_dropin.cpp).
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "htslib/faidx.h"
#include "htslib/sam.h"
#include "nanopolish_eventalign.h"
#include "nanopolish_squiggle_read.h"

// One record of a BamProcessor batch with what realign_read (src/alignment/nanopolish_eventalign.cpp:539-610) loads for it:
struct NpRealignRead {
    // in
    const bam1_t* record = NULL;
    std::string read_name;                     // bam_get_qname(record)
    const std::string* read_sequence = NULL;   // ReadDB::get_read_sequence
    const float* raw_pa = NULL;                // Fast5Data::rt.raw in pA
    size_t n_raw = 0;
    double sample_rate = 4000.0;               // Fast5Data::channel_params.sample_rate
    size_t read_idx = 0;
    int rna = 0;                               // SquiggleRead::nucleotide_type == SRNT_RNA: kit r9.4_70bps, alphabet u_to_t_rna, k = 5, the RNA
                                               // detector, events reversed (squiggle_read.cpp:206-213,260-263); a batch may mix both types
    // out
    std::shared_ptr<SquiggleRead> sr;          // the read as SquiggleRead(sequence, Fast5Data, 0) leaves it after load_from_raw: events,
                                               // scalings, base_to_event_map, events_per_base, base model (what the reference's writers read)
    std::vector<EventAlignment> alignment;     // align_read_to_ref for strand 0
    int status = 0;                            // NP_REALIGN_*
};
#define NP_REALIGN_OK 0
#define NP_REALIGN_NO_EVENTS 1      // the read failed the aligner / calibration / events-per-base gates: no events, no alignment (as the reference)
#define NP_REALIGN_HOST_PATH 2      // not processed on the device: the caller runs realign_read on it

// SquiggleRead::load_from_raw + align_read_to_ref (strand 0) for every record of the batch in one device pass:
// detect_events -> MoM scalings -> adaptive banded event alignment -> event map + recalibrate_model -> the segment chain of
// profile_hmm_align calls.  Fills sr / alignment / status of every read; the caller then runs its writer
// (emit_event_alignment_tsv / _sam, summarize_alignment) on them exactly as realign_read does.
// A record that only partly overlaps [region_start, region_end] (the reference trims its aligned pairs, :661-663), a spliced record
// or a read whose signal holds a non-finite sample come back NP_REALIGN_HOST_PATH.  Direct-RNA reads (rna = 1) run on the device like
// DNA reads, as a group of their own (k = 5 models, the RNA detector, reversed events): a batch may mix both.
void np_realign_reads_batch(std::vector<NpRealignRead>& reads, const faidx_t* fai, const bam_hdr_t* hdr, int region_start, int region_end);
