// np_variants_dropin.cpp -- the variant callers' scoring loops on the device, one batch per call instead of one launch per
// profile_hmm_score.
//
// In the reference `nanopolish variants` spends its time in two loops over (read x haplotype):
//     score_variant_thresholded      src/common/nanopolish_variant.cpp:765-799   screening: base vs variant haplotype, every read
//     score_variant_group            src/common/nanopolish_variant.cpp:182-262   genotyping: every haplotype of a group, every read
// each iteration one profile_hmm_score_set (src/hmm/nanopolish_profile_hmm.cpp:32-56) = one forward pass per methylation alphabet.
// Through the per-call shim (np_dropin.cpp) each of those passes is its own upload + launch + download.  This file is compiled
// INSIDE a nanopolish build like np_dropin.cpp and gathers a whole call's passes into one device batch through the C ABI:
//   * every distinct read of the call is uploaded ONCE (event means, scalings, transitions), every distinct (sequence, strand) ONCE
//     (k-mer ranks); a forward pass is a 32-byte work item (np_hmm_job_dev) pointing at both;
//   * one np_hmm_score_dev per pore model in the call (base model + one per methylation alphabet), scores read back once;
//   * the set combination (log-sum over the alphabets' scores, minus log #models) and the callers' accumulation run on the host with
//     the reference's own add_logs, in the reference's order.
// oracle/Makefile links it into `make -C oracle batch`; tests/test_gpu_variants_dropin.py compares it with the unmodified reference.
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <sstream>
#include "np_variants_dropin.h"
#include "nanopolish_profile_hmm.h"
#include "nanopolish_squiggle_read.h"
#include "np_hmm.h"
#include "np_shim_common.h"

extern double hmm_indel_bias_factor;   // src/hmm/nanopolish_profile_hmm_r9.cpp:19

using np_shim::shim;
using np_shim::check;
using np_shim::die;
using np_shim::Layout;
using np_shim::Blob;

namespace {

// persistent device buffers of this binding (one call at a time)
struct Buffers {
    std::mutex lock;
    Blob in, out;
    Buffers() : in(true), out(true) {}
};
Buffers& buffers() { static Buffers b; return b; }

} // namespace

std::vector<double> np_profile_hmm_score_sets(const std::vector<const std::vector<HMMInputSequence>*>& sets,
                                              const std::vector<const HMMInputData*>& data, uint32_t flags)
{
    const size_t n_sets = sets.size();
    std::vector<double> result(n_sets, 0.0);
    if (n_sets == 0) return result;
    np_ctx* c = shim().get();
    Buffers& B = buffers();
    std::lock_guard<std::mutex> g(B.lock);

    // ---- distinct reads: events, scalings, transitions ---------------------------------------------------------------------
    std::map<std::pair<const SquiggleRead*, int>, int> read_index;
    std::vector<std::pair<const SquiggleRead*, int> > read_list;
    std::vector<int> set_read(n_sets);
    // Preconditions are checked at run time (the reference's own are asserts, gone under NDEBUG): a set the device pass does not cover
    // must not come back as a silent NaN (np_hmm_score_dev marks such work items with a NaN score).
    std::vector<char> on_host(n_sets, 0);            // sets scored by the reference's per-call path instead (see below)
    for (size_t i = 0; i < n_sets; ++i) {
        const HMMInputData& d = *data[i];
        if (d.read->pore_type != PORETYPE_R9) die("np_profile_hmm_score_sets: only R9 reads are supported (profile_hmm.cpp:16-28 would take the R7 path)");
        if (!((d.rc && d.event_stride == -1) || (!d.rc && d.event_stride == 1)))      // r9.inl:275
            die("np_profile_hmm_score_sets: event stride does not match the strand (nanopolish_profile_hmm_r9.inl:275)");
        const std::pair<const SquiggleRead*, int> key(d.read, (int)d.strand);
        std::map<std::pair<const SquiggleRead*, int>, int>::iterator it = read_index.find(key);
        if (it == read_index.end()) { it = read_index.insert(std::make_pair(key, (int)read_list.size())).first; read_list.push_back(key); }
        set_read[i] = it->second;
    }
    const int n_reads = (int)read_list.size();
    std::vector<int64_t> event_off(n_reads + 1, 0);
    for (int r = 0; r < n_reads; ++r) event_off[r + 1] = event_off[r] + (int64_t)read_list[r].first->events[read_list[r].second].size();

    // ---- distinct (sequence, strand): k-mer ranks; work items per pore model ------------------------------------------------------
    struct Item { int set, seq; };
    std::map<const PoreModel*, std::vector<Item> > by_model;
    std::map<std::pair<const HMMInputSequence*, int>, int64_t> rank_index;       // -> offset into ranks
    std::vector<uint16_t> ranks;
    uint32_t k = 0;
    for (size_t i = 0; i < n_sets; ++i) {
        const std::vector<HMMInputSequence>& seqs = *sets[i];
        const HMMInputData& d = *data[i];
        if (seqs.empty()) die("np_profile_hmm_score_sets: an empty sequence set (profile_hmm.cpp:36)");
        if (std::string(seqs[0].get_alphabet()->get_name()) != "nucleotide" || std::string(d.pore_model->pmalphabet->get_name()) != "nucleotide")
            die("np_profile_hmm_score_sets: the first sequence of a set and the data's pore model must be over the nucleotide alphabet (profile_hmm.cpp:34-35)");
        // what the kernels do not cover goes to the reference's per-call function for the WHOLE set: more than NP_MAX_KMERS k-mers, a
        // sequence shorter than k, an event window longer than the clip-flank table
        const uint64_t window = (uint64_t)(d.event_stop_idx > d.event_start_idx ? d.event_stop_idx - d.event_start_idx : d.event_start_idx - d.event_stop_idx) + 1;
        for (size_t s = 0; s < seqs.size(); ++s) {
            const PoreModel* pm = s == 0 ? d.pore_model : d.read->get_model(d.strand, seqs[s].get_alphabet()->get_name());
            if (pm == NULL) die("np_profile_hmm_score_sets: the read has no pore model for a sequence's alphabet");
            if (seqs[s].length() < pm->k || seqs[s].length() - pm->k + 1 > NP_MAX_KMERS || window > NP_MAX_WINDOW_EVENTS) on_host[i] = 1;
        }
        if (on_host[i]) continue;
        for (size_t s = 0; s < seqs.size(); ++s) {
            const PoreModel* pm = s == 0 ? d.pore_model : d.read->get_model(d.strand, seqs[s].get_alphabet()->get_name());
            if (pm->states.size() != seqs[s].get_num_kmer_ranks(pm->k))               // r9.inl:305
                die("np_profile_hmm_score_sets: pore model and sequence alphabet disagree on the number of k-mers (nanopolish_profile_hmm_r9.inl:305)");
            k = pm->k;
            const std::pair<const HMMInputSequence*, int> key(&seqs[s], d.rc ? 1 : 0);
            if (rank_index.find(key) == rank_index.end()) {
                rank_index[key] = (int64_t)ranks.size();
                const uint32_t nk = seqs[s].length() - pm->k + 1;
                for (uint32_t q = 0; q < nk; ++q) ranks.push_back((uint16_t)seqs[s].get_kmer_rank(q, pm->k, d.rc));
            }
            Item it; it.set = (int)i; it.seq = (int)s;
            by_model[pm].push_back(it);
        }
    }
    int64_t n_jobs = 0;
    for (std::map<const PoreModel*, std::vector<Item> >::const_iterator it = by_model.begin(); it != by_model.end(); ++it) n_jobs += (int64_t)it->second.size();

    // ---- one pinned blob up ---------------------------------------------------------------------------------------------------
    Layout li;
    const size_t i_events = li.add((size_t)event_off[n_reads] * sizeof(float)), i_reads = li.add((size_t)n_reads * sizeof(np_read_dev)),
                 i_ranks = li.add(ranks.size() * sizeof(uint16_t)), i_jobs = li.add((size_t)n_jobs * sizeof(np_hmm_job_dev));
    B.in.reserve(c, li.size + 256); B.out.reserve(c, (size_t)n_jobs * sizeof(float) + 256);
    float* h_events = (float*)(B.in.h + i_events); np_read_dev* h_reads = (np_read_dev*)(B.in.h + i_reads);
    uint16_t* h_ranks = (uint16_t*)(B.in.h + i_ranks); np_hmm_job_dev* h_jobs = (np_hmm_job_dev*)(B.in.h + i_jobs);
    #pragma omp parallel for schedule(dynamic)
    for (int r = 0; r < n_reads; ++r) {
        const SquiggleRead* sr = read_list[r].first; const int strand = read_list[r].second;
        const SquiggleScalings& sc = sr->scalings[strand];
        if (sc.drift != 0.0) die("np_profile_hmm_score_sets: a read with a drift term (always 0 on the R9 path, squiggle_read.cpp:310)");   // (exit inside the parallel region: a precondition, not a device error)
        const size_t ne = sr->events[strand].size();
        for (size_t e = 0; e < ne; ++e) h_events[event_off[r] + e] = sr->events[strand][e].mean;
        np_fill_read_host(&h_reads[r], sc.shift, sc.scale, sc.var, event_off[r], (uint32_t)ne, 0, 1);
        np_calculate_transitions(sr->events_per_base[strand], hmm_indel_bias_factor, h_reads[r].trans);
    }
    memcpy(h_ranks, ranks.data(), ranks.size() * sizeof(uint16_t));
    std::vector<std::pair<int64_t, std::pair<const PoreModel*, int64_t> > > launches;    // (first job, (model, count))
    std::vector<std::vector<int64_t> > job_of(n_sets);
    for (size_t i = 0; i < n_sets; ++i) job_of[i].assign(sets[i]->size(), -1);
    int64_t j = 0;
    for (std::map<const PoreModel*, std::vector<Item> >::const_iterator it = by_model.begin(); it != by_model.end(); ++it) {
        launches.push_back(std::make_pair(j, std::make_pair(it->first, (int64_t)it->second.size())));
        for (size_t q = 0; q < it->second.size(); ++q, ++j) {
            const Item& im = it->second[q];
            const HMMInputData& d = *data[im.set];
            const HMMInputSequence& sq = (*sets[im.set])[im.seq];
            np_hmm_job_dev& jd = h_jobs[j];
            jd.rank_off = rank_index[std::make_pair(&sq, d.rc ? 1 : 0)];
            jd.n_kmers = sq.length() - it->first->k + 1;
            jd.read = (uint32_t)set_read[im.set];
            jd.e_start = d.event_start_idx; jd.e_stop = d.event_stop_idx; jd.stride = d.event_stride; jd.flags = flags;
            job_of[im.set][im.seq] = j;
        }
    }
    (void)k;

    // ---- the batch on the device: one forward launch set per pore model -------------------------------------------------------------
    if (n_jobs > 0) check(np_copy_to_device(c, NULL, B.in.d, B.in.h, li.size), "np_copy_to_device");
    for (size_t l = 0; l < launches.size(); ++l) {
        const int model = shim().model_id(launches[l].second.first);
        const int64_t j0 = launches[l].first, nj = launches[l].second.second;
        check(np_hmm_score_dev(c, NULL, nj, (const np_hmm_job_dev*)(B.in.d + i_jobs) + j0, (const np_read_dev*)(B.in.d + i_reads),
                               (const float*)(B.in.d + i_events), (const uint16_t*)(B.in.d + i_ranks), model, (float*)B.out.d + j0), "np_hmm_score_dev");
    }
    if (n_jobs > 0) check(np_copy_to_host(c, NULL, B.out.h, B.out.d, (size_t)n_jobs * sizeof(float)), "np_copy_to_host");
    check(np_sync(c, NULL), "np_sync");
    const float* sc = (const float*)B.out.h;

    // ---- profile_hmm_score_set's combination (profile_hmm.cpp:38-54), the reference's own add_logs ------------------------------------
    for (size_t i = 0; i < n_sets; ++i) {
        if (on_host[i]) {
            // the reference's own per-call function (in a drop-in build that is np_dropin.cpp's, which refuses sizes the library does
            // not cover with NP_ERR_UNSUPPORTED and a message -- loudly, never a NaN)
            result[i] = profile_hmm_score_set(*sets[i], *data[i], flags);
            continue;
        }
        const size_t num_models = sets[i]->size();
        for (size_t s = 0; s < num_models; ++s)
            if (sc[job_of[i][s]] != sc[job_of[i][s]]) die("np_profile_hmm_score_sets: the device returned no score for a work item (NaN)");
        const double num_model_penalty = log(num_models);
        double score = sc[job_of[i][0]] - num_model_penalty;
        for (size_t s = 1; s < num_models; ++s) {
            const double alt_score = sc[job_of[i][s]] - num_model_penalty;
            score = add_logs(score, alt_score);
        }
        result[i] = (float)score;                 // the function returns float
    }
    return result;
}

std::vector<std::vector<Variant> > np_score_variants_thresholded(const std::vector<NpVariantWindow>& windows, uint32_t alignment_flags,
                                                                uint32_t score_threshold, const std::vector<std::string>& methylation_types)
{
    // the haplotypes' sequence sets (variant.cpp:771-779): per window the base, then one per variant
    std::vector<std::vector<std::vector<HMMInputSequence> > > seqs(windows.size());
    std::vector<const std::vector<HMMInputSequence>*> sets;
    std::vector<const HMMInputData*> data;
    for (size_t w = 0; w < windows.size(); ++w) {
        const NpVariantWindow& W = windows[w];
        seqs[w].reserve(W.variants.size() + 1);
        seqs[w].push_back(generate_methylated_alternatives(W.base_haplotype.get_sequence(), methylation_types));
        for (size_t v = 0; v < W.variants.size(); ++v) {
            Haplotype variant_haplotype = W.base_haplotype;
            variant_haplotype.apply_variant(W.variants[v]);
            seqs[w].push_back(generate_methylated_alternatives(variant_haplotype.get_sequence(), methylation_types));
        }
    }
    for (size_t w = 0; w < windows.size(); ++w)
        for (size_t h = 0; h < seqs[w].size(); ++h)
            for (size_t j = 0; j < windows[w].input.size(); ++j) { sets.push_back(&seqs[w][h]); data.push_back(&windows[w].input[j]); }
    const std::vector<double> sc = np_profile_hmm_score_sets(sets, data, alignment_flags);

    std::vector<std::vector<Variant> > out(windows.size());
    size_t at = 0;
    for (size_t w = 0; w < windows.size(); ++w) {
        const size_t nr = windows[w].input.size();
        const double* base = sc.data() + at;
        for (size_t v = 0; v < windows[w].variants.size(); ++v) {
            const double* var = sc.data() + at + (v + 1) * nr;
            Variant out_variant = windows[w].variants[v];
            double total_score = 0.0f;
            for (size_t j = 0; j < nr; ++j)                                    // variant.cpp:782-795, in read order
                if (fabs(total_score) < score_threshold) total_score += (var[j] - base[j]);
            out_variant.quality = total_score;
            out[w].push_back(out_variant);
        }
        at += (windows[w].variants.size() + 1) * nr;
    }
    return out;
}

Variant np_score_variant_thresholded(const Variant& input_variant, Haplotype base_haplotype, const std::vector<HMMInputData>& input,
                                     const uint32_t alignment_flags, const uint32_t score_threshold,
                                     const std::vector<std::string>& methylation_types)
{
    std::vector<NpVariantWindow> w(1, NpVariantWindow(base_haplotype));
    w[0].variants.push_back(input_variant);
    w[0].input = input;
    return np_score_variants_thresholded(w, alignment_flags, score_threshold, methylation_types)[0][0];
}

void np_score_variant_group(VariantGroup& variant_group, Haplotype base_haplotype, const std::vector<HMMInputData>& input,
                            const int max_haplotypes, const int ploidy, const bool genotype_all_input_variants,
                            const uint32_t alignment_flags, const std::vector<std::string>& methylation_types)
{
    (void)ploidy; (void)genotype_all_input_variants;
    const size_t num_variants = variant_group.get_num_variants();
    // the variant combinations that fit max_haplotypes and their haplotypes, as variant.cpp:191-232
    size_t sum_num_haplotypes = 0, max_r = 1;
    while (max_r <= num_variants) {
        const size_t num_haplotypes_r = nChoosek(num_variants, max_r);
        if (num_haplotypes_r + sum_num_haplotypes < (size_t)max_haplotypes) sum_num_haplotypes += num_haplotypes_r;
        else break;
        max_r += 1;
    }
    max_r -= 1;
    if (max_r != num_variants)
        fprintf(stderr, "Number of variants in span (%lu) would exceed max-haplotypes. Variants may be missed. Consider running with a higher value of max-haplotypes!\n", num_variants);
    std::vector<std::pair<Haplotype, size_t> > haplotypes;
    for (size_t r = 0; r <= max_r; ++r) {
        Combinations combinations(num_variants, r);
        while (!combinations.done()) {
            VariantCombination vc(combinations.get());
            Haplotype current_haplotype = base_haplotype;
            const bool good_haplotype = current_haplotype.apply_variants(variant_group.get_variants(vc));
            if (good_haplotype) {
                const size_t vc_idx = variant_group.add_combination(vc);
                haplotypes.push_back(std::make_pair(current_haplotype, vc_idx));
            }
            combinations.next();
        }
    }
    std::vector<std::string> read_ids;
    for (size_t i = 0; i < input.size(); ++i) {
        std::stringstream ss;
        ss << input[i].read->read_name << ":" << input[i].strand;
        read_ids.push_back(ss.str());
        variant_group.set_read_strand(ss.str(), input[i].rc);
    }
    // every (read, haplotype) profile_hmm_score_set of variant.cpp:241-257 in one device batch
    std::vector<std::vector<HMMInputSequence> > seqs(haplotypes.size());
    for (size_t hi = 0; hi < haplotypes.size(); ++hi) seqs[hi] = generate_methylated_alternatives(haplotypes[hi].first.get_sequence(), methylation_types);
    std::vector<const std::vector<HMMInputSequence>*> sets;
    std::vector<const HMMInputData*> data;
    for (size_t ri = 0; ri < input.size(); ++ri)
        for (size_t hi = 0; hi < haplotypes.size(); ++hi) { sets.push_back(&seqs[hi]); data.push_back(&input[ri]); }
    const std::vector<double> sc = np_profile_hmm_score_sets(sets, data, alignment_flags);
    for (size_t ri = 0; ri < input.size(); ++ri)
        for (size_t hi = 0; hi < haplotypes.size(); ++hi)
            variant_group.set_combination_read_score(haplotypes[hi].second, read_ids[ri], sc[ri * haplotypes.size() + hi]);
}
