// np_variants_dropin.h -- the batched reference-side binding of the variant callers' scoring loops (see np_variants_dropin.cpp).
#pragma once
#include <string>
#include <vector>
#include "nanopolish_haplotype.h"
#include "nanopolish_variant.h"
#include "nanopolish_variant_db.h"

// One screening window: what one iteration of generate_candidate_single_base_edits' position loop hands to score_variant_thresholded
// (src/nanopolish_call_variants.cpp:300-352): the window's haplotype, its candidate variants, the reads' event subsequences.
struct NpVariantWindow {
    Haplotype base_haplotype;
    std::vector<Variant> variants;
    std::vector<HMMInputData> input;          // AlignmentDB::get_event_subsequences(contig, calling_start, calling_end)
    NpVariantWindow(const Haplotype& h) : base_haplotype(h) {}
};

// score_variant_thresholded (src/common/nanopolish_variant.cpp:765-799) for every variant of every window, ALL
// (read x haplotype x alphabet) forward passes of the call in one device batch.  out[w][v] = windows[w].variants[v] with its quality.
// The reference accumulates `variant_score - base_score` over the reads under `omp parallel for` with a racy early-out
// (`fabs(total_score) < score_threshold`); this binding computes every read's two scores and replays the loop in read order,
// i.e. it returns what the reference returns with ONE OpenMP thread (any other thread count gives the reference itself an
// order-dependent sum).  The base haplotype of a window is scored once, not once per variant.
std::vector<std::vector<Variant> > np_score_variants_thresholded(const std::vector<NpVariantWindow>& windows, uint32_t alignment_flags,
                                                                uint32_t score_threshold, const std::vector<std::string>& methylation_types);

// the reference's own signature (one variant, one window)
Variant np_score_variant_thresholded(const Variant& input_variant, Haplotype base_haplotype, const std::vector<HMMInputData>& input,
                                     const uint32_t alignment_flags, const uint32_t score_threshold,
                                     const std::vector<std::string>& methylation_types);

// score_variant_group (src/common/nanopolish_variant.cpp:182-262): every (read, haplotype) profile_hmm_score_set of the group in one
// device batch; the group's read scores are set exactly as the reference sets them (set_combination_read_score).
void np_score_variant_group(VariantGroup& variant_group, Haplotype base_haplotype, const std::vector<HMMInputData>& input,
                            const int max_haplotypes, const int ploidy, const bool genotype_all_input_variants,
                            const uint32_t alignment_flags, const std::vector<std::string>& methylation_types);

// The building block: profile_hmm_score_set (src/hmm/nanopolish_profile_hmm.cpp:32-56) for many (sequence set, read) pairs at once.
// sets[i] is scored against *data[i]; returns one score per pair (the float the reference returns, widened).
std::vector<double> np_profile_hmm_score_sets(const std::vector<const std::vector<HMMInputSequence>*>& sets,
                                              const std::vector<const HMMInputData*>& data, uint32_t flags);
