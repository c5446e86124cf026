// np_eventalign_dropin.cpp -- eventalign's per-record work for a whole BamProcessor batch on the device.
//
// In the reference every record runs realign_read (src/alignment/nanopolish_eventalign.cpp:539-610) under OpenMP:
//     SquiggleRead sr(read_name, read_db, flags)        load_from_raw: detect_events, MoM scalings, event alignment, event map,
//                                                       recalibrate_model, QC gates (src/nanopolish_squiggle_read.cpp:186-336)
//     align_read_to_ref(params)                         the chain of ~100-base profile_hmm_align segments (:612-826)
//     emit_event_alignment_tsv / _sam, summarize        the writers
// This file (compiled INSIDE a nanopolish build) replaces the first two for a batch: one upload, nine enqueues through the C ABI
// (np_detect_events_dev -> np_mom_fill_dev -> np_event_align_dev -> np_calibrate_resolve_dev -> np_eventalign_dev), one read-back,
// and then rebuilds on the host what the writers need -- a SquiggleRead with the fields load_from_raw sets, and the EventAlignment
// rows (:774-812) -- so that the reference's own writers run unchanged on the results.
// tests/test_gpu_eventalign_dropin.py compares the rebuilt reads field by field and the TSV text the reference's writer prints
// from them with the unmodified reference.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <omp.h>
#include "np_eventalign_dropin.h"
#include "nanopolish_alphabet.h"
#include "nanopolish_pore_model_set.h"
#include "np_hmm.h"
#include "np_shim_common.h"

using np_shim::shim;
using np_shim::check;
using np_shim::Layout;
using np_shim::Blob;

namespace {

struct Buffers {
    std::mutex lock;
    Blob in, out, scratch;
    Buffers() : in(true), out(true), scratch(false) {}
};
Buffers& buffers() { static Buffers b; return b; }

inline int base_code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
bool plain_acgt(const std::string& s)
{
    for (size_t i = 0; i < s.size(); ++i) if (base_code(s[i]) > 3) return false;
    return true;
}
bool spliced(const bam1_t* b)
{
    const uint32_t* cg = bam_get_cigar(b);
    for (uint32_t i = 0; i < b->core.n_cigar; ++i) if (bam_cigar_op(cg[i]) == BAM_CREF_SKIP) return true;
    return false;
}

} // namespace

// one device pass over the records of ONE nucleotide type (load_from_raw picks kit, alphabet, k and detector by it, squiggle_read.cpp:197-213)
static void realign_group(std::vector<NpRealignRead>& reads, const bool rna, const faidx_t* fai, const bam_hdr_t* hdr, int region_start, int region_end)
{
    const int n_all = (int)reads.size();
    np_ctx* c = shim().get();
    Buffers& B = buffers();
    // DNA: what load_from_raw hard-codes (squiggle_read.cpp:197-202); direct RNA: kit r9.4_70bps, alphabet u_to_t_rna, k = 5, the RNA
    // detector, U read as T, and the events reversed after the MoM scalings (:206-213, :260-263)
    const std::string kit = rna ? "r9.4_70bps" : "r9.4_450bps", alphabet = rna ? "u_to_t_rna" : "nucleotide", strand_str = "template";
    const uint32_t k = rna ? 5 : 6;
    if (!PoreModelSet::has_model(kit, alphabet, strand_str, k)) {          // (a build without the RNA models: the caller's own path)
        for (int i = 0; i < n_all; ++i) if ((reads[i].rna != 0) == rna) reads[i].status = NP_REALIGN_HOST_PATH;
        return;
    }
    const PoreModel* pm = PoreModelSet::get_model(kit, alphabet, strand_str, k);
    std::vector<std::string> seq_t(rna ? n_all : 0);                      // RNA: the read sequence with U -> T (:212)

    // ---- phase 1: which records go to the device; reference segments; sizes ----------------------------------------------------------
    std::vector<int> idx;
    std::vector<std::string> ref_seqs(n_all);
    for (int i = 0; i < n_all; ++i) {
        NpRealignRead& R = reads[i];
        if ((R.rna != 0) != rna) continue;                                // the other group's record
        R.alignment.clear(); R.sr.reset(); R.status = NP_REALIGN_OK;
        const bam1_t* b = R.record;
        if (rna && R.read_sequence) { seq_t[i] = *R.read_sequence; std::replace(seq_t[i].begin(), seq_t[i].end(), 'U', 'T'); }
        const std::string* rs = rna ? &seq_t[i] : R.read_sequence;
        bool fits = b && R.read_sequence && rs->length() > 20 && R.raw_pa && R.n_raw >= 64 &&      // (:141-146)
                    (b->core.flag & BAM_FUNMAP) == 0 && !spliced(b) && plain_acgt(*rs);
        if (fits && region_start != -1 && region_end != -1)            // trim_aligned_pairs_to_ref_region would cut the record: host path
            fits = b->core.pos >= region_start && bam_endpos(b) <= region_end;
        if (!fits) { R.status = NP_REALIGN_HOST_PATH; continue; }
        idx.push_back(i);
    }
    const int n = (int)idx.size();
    if (n == 0) return;
    #pragma omp parallel for schedule(dynamic)
    for (int q = 0; q < n; ++q) {
        const int i = idx[q];
        const bam1_t* b = reads[i].record;
        int fetched_len = 0;
        ref_seqs[i] = get_reference_region_ts(fai, hdr->target_name[b->core.tid], b->core.pos, bam_endpos(b), &fetched_len);     // :626-640
        if (!plain_acgt(ref_seqs[i])) {
            std::transform(ref_seqs[i].begin(), ref_seqs[i].end(), ref_seqs[i].begin(), ::toupper);
            ref_seqs[i] = pm->pmalphabet->disambiguate(ref_seqs[i]);
        }
    }
    std::vector<int64_t> raw_off(n + 1, 0), event_off(n + 1, 0), rank_off(n + 1, 0), cigar_off(n + 1, 0), pair_off(n + 1, 0), genome_off(n + 1, 0),
                         out_off(n + 1, 0);
    int64_t max_samples = 1, max_events = 1, max_bands = 1;
    for (int q = 0; q < n; ++q) {
        const NpRealignRead& R = reads[idx[q]];
        const int64_t n_raw = (int64_t)R.n_raw, nk = (int64_t)(rna ? seq_t[idx[q]].size() : R.read_sequence->size()) - k + 1, ecap = n_raw / 2 + 2;
        raw_off[q + 1] = raw_off[q] + n_raw; event_off[q + 1] = event_off[q] + ecap; rank_off[q + 1] = rank_off[q] + nk;
        cigar_off[q + 1] = cigar_off[q] + R.record->core.n_cigar; genome_off[q + 1] = genome_off[q] + (int64_t)ref_seqs[idx[q]].size();
        pair_off[q + 1] = pair_off[q] + ecap + nk + 2; out_off[q + 1] = out_off[q] + ecap + 1;
        max_samples = std::max(max_samples, n_raw); max_events = std::max(max_events, ecap); max_bands = std::max(max_bands, ecap + nk + 2);
    }
    const int64_t n_ev = event_off[n], n_rk = rank_off[n], n_rows = out_off[n];

    Layout li;
    const size_t i_raw = li.add((size_t)raw_off[n] * 4), i_ranks = li.add((size_t)n_rk * 2), i_reads_a = li.add((size_t)n * sizeof(np_read_dev)),
                 i_reads_b = li.add((size_t)n * sizeof(np_read_dev)), i_genome = li.add((size_t)genome_off[n]), i_raw_off = li.add((size_t)(n + 1) * 8),
                 i_event_off = li.add((size_t)(n + 1) * 8), i_cigar_off = li.add((size_t)(n + 1) * 8), i_pair_off = li.add((size_t)(n + 1) * 8),
                 i_out_off = li.add((size_t)(n + 1) * 8), i_ref_begin = li.add((size_t)n * 8), i_ref_len = li.add((size_t)n * 4),
                 i_read_len = li.add((size_t)n * 4), i_cigar = li.add((size_t)cigar_off[n] * 4), i_rc = li.add((size_t)n);
    Layout lo;     // read back: the events, the calibrated records, the event map, the gates, the rows
    const size_t o_ev_start = lo.add((size_t)n_ev * 4), o_ev_len = lo.add((size_t)n_ev * 4), o_ev_mean = lo.add((size_t)n_ev * 4),
                 o_ev_stdv = lo.add((size_t)n_ev * 4), o_reads_a = lo.add((size_t)n * sizeof(np_read_dev)), o_reads_b = lo.add((size_t)n * sizeof(np_read_dev)),
                 o_map_start = lo.add((size_t)n_rk * 4), o_map_stop = lo.add((size_t)n_rk * 4), o_epb = lo.add((size_t)n * 8),
                 o_n_events = lo.add((size_t)n * 4), o_n_pairs = lo.add((size_t)n * 4), o_calibrated = lo.add((size_t)n * 4),
                 o_out_ref = lo.add((size_t)n_rows * 4), o_out_event = lo.add((size_t)n_rows * 4), o_out_state = lo.add((size_t)n_rows),
                 o_n_out = lo.add((size_t)n * 4), o_status = lo.add((size_t)n * 4), o_n_calls = lo.add((size_t)n * 4);
    Layout ls;
    const size_t s_pair_begin = ls.add((size_t)n * 4), s_dummy = ls.add(256);
    const size_t zero_bytes = ls.size;
    const size_t s_tstat = ls.add((size_t)(2 * raw_off[n] + 16) * 4), s_pairs = ls.add((size_t)pair_off[n] * sizeof(np_pair));
    B.in.reserve(c, li.size + 256); B.out.reserve(c, lo.size + 256); B.scratch.reserve(c, ls.size + 256);

    // ---- pack ------------------------------------------------------------------------------------------------------------------
    char* H = B.in.h;
    float* h_raw = (float*)(H + i_raw); uint16_t* h_ranks = (uint16_t*)(H + i_ranks);
    np_read_dev* h_reads_a = (np_read_dev*)(H + i_reads_a); np_read_dev* h_reads_b = (np_read_dev*)(H + i_reads_b);
    #pragma omp parallel for schedule(dynamic)
    for (int q = 0; q < n; ++q) {
        const NpRealignRead& R = reads[idx[q]];
        const std::string& seq = rna ? seq_t[idx[q]] : *R.read_sequence;
        for (int t = 0; t < 2; ++t)
            np_fill_read_host(t ? &h_reads_b[q] : &h_reads_a[q], 0.0, 1.0, 1.0, event_off[q], (uint32_t)(event_off[q + 1] - event_off[q]), rank_off[q],
                              (uint32_t)(rank_off[q + 1] - rank_off[q]));
        memcpy(h_raw + raw_off[q], R.raw_pa, R.n_raw * 4);
        const uint32_t mask = (1u << (2 * k)) - 1u;
        uint32_t r = 0;
        for (size_t j = 0; j < seq.size(); ++j) {                        // Alphabet::kmer_rank of every k-mer, one rolling pass (ACGT only)
            r = ((r << 2) | (uint32_t)base_code(seq[j])) & mask;
            if (j + 1 >= k) h_ranks[rank_off[q] + j + 1 - k] = (uint16_t)r;
        }
        memcpy(H + i_genome + genome_off[q], ref_seqs[idx[q]].data(), ref_seqs[idx[q]].size());
        ((int64_t*)(H + i_ref_begin))[q] = genome_off[q]; ((int32_t*)(H + i_ref_len))[q] = (int32_t)ref_seqs[idx[q]].size();
        memcpy((uint32_t*)(H + i_cigar) + cigar_off[q], bam_get_cigar(R.record), 4 * (size_t)R.record->core.n_cigar);
        ((int32_t*)(H + i_read_len))[q] = (int32_t)seq.size();
        ((uint8_t*)(H + i_rc))[q] = bam_is_rev(R.record) ? 1 : 0;
    }
    memcpy(H + i_raw_off, raw_off.data(), (size_t)(n + 1) * 8); memcpy(H + i_event_off, event_off.data(), (size_t)(n + 1) * 8);
    memcpy(H + i_cigar_off, cigar_off.data(), (size_t)(n + 1) * 8); memcpy(H + i_pair_off, pair_off.data(), (size_t)(n + 1) * 8);
    memcpy(H + i_out_off, out_off.data(), (size_t)(n + 1) * 8);

    // ---- the batch on the device --------------------------------------------------------------------------------------------------
    const int m_nuc = shim().model_id(pm);
    char* D = B.in.d; char* O = B.out.d; char* X = B.scratch.d;
    check(np_copy_to_device(c, NULL, D, H, li.size), "np_copy_to_device");
    check(np_memset_dev(c, NULL, O, 0, lo.size), "np_memset_dev");
    check(np_memset_dev(c, NULL, X, 0, zero_bytes), "np_memset_dev");
    np_detector_param prm;
    np_event_detection_params(&prm, rna ? 1 : 0);
    np_read_dev* reads_a = (np_read_dev*)(D + i_reads_a); np_read_dev* reads_b = (np_read_dev*)(D + i_reads_b);
    float* ev_mean = (float*)(O + o_ev_mean);
    check(np_detect_events_dev(c, NULL, n, (float*)(D + i_raw), (int64_t*)(D + i_raw_off), max_samples, &prm, (float*)(X + s_tstat),
                               (int64_t*)(D + i_event_off), max_events, (uint32_t*)(O + o_ev_start), (float*)(O + o_ev_len), ev_mean,
                               (float*)(O + o_ev_stdv), (int32_t*)(O + o_n_events)), "np_detect_events_dev");
    check(np_mom_fill_dev(c, NULL, n, reads_a, reads_b, ev_mean, (int32_t*)(O + o_n_events), (uint16_t*)(D + i_ranks), m_nuc), "np_mom_fill_dev");
    if (rna)
        check(np_reverse_events_dev(c, NULL, n, (int64_t*)(D + i_event_off), (int32_t*)(O + o_n_events), (uint32_t*)(O + o_ev_start), (float*)(O + o_ev_len),
                                    ev_mean, (float*)(O + o_ev_stdv)), "np_reverse_events_dev");
    check(np_event_align_dev(c, NULL, n, reads_a, ev_mean, (uint16_t*)(D + i_ranks), m_nuc, max_bands, (int64_t*)(D + i_pair_off),
                             (np_pair*)(X + s_pairs), (int32_t*)(X + s_pair_begin), (int32_t*)(O + o_n_pairs)), "np_event_align_dev");
    check(np_calibrate_resolve_dev(c, NULL, n, reads_b, ev_mean, (uint16_t*)(D + i_ranks), m_nuc, (int64_t*)(D + i_pair_off), (np_pair*)(X + s_pairs),
                                   (int32_t*)(X + s_pair_begin), (int32_t*)(O + o_n_pairs), (int32_t*)(O + o_map_start), (int32_t*)(O + o_map_stop),
                                   (double*)(O + o_epb), (int32_t*)(O + o_calibrated), 0, (np_hmm_job_dev*)(X + s_dummy), (int32_t*)(X + s_dummy)),
          "np_calibrate_resolve_dev");
    check(np_eventalign_dev(c, NULL, n, reads_b, ev_mean, (int32_t*)(O + o_map_start), (int32_t*)(O + o_n_pairs), (double*)(O + o_epb),
                            (int32_t*)(O + o_calibrated), m_nuc, D + i_genome, (int64_t*)(D + i_ref_begin), (int32_t*)(D + i_ref_len),
                            (uint32_t*)(D + i_cigar), (int64_t*)(D + i_cigar_off), cigar_off[n], (int32_t*)(D + i_read_len), (uint8_t*)(D + i_rc), k,
                            (int64_t*)(D + i_out_off), (int32_t*)(O + o_out_ref), (int32_t*)(O + o_out_event), (uint8_t*)(O + o_out_state),
                            (int32_t*)(O + o_n_out), (int32_t*)(O + o_status), (int32_t*)(O + o_n_calls)), "np_eventalign_dev");
    // the records as the kernels left them (MoM scalings in a, calibrated scalings in b) travel back with the results
    check(np_copy_to_host(c, NULL, B.out.h, O, lo.size), "np_copy_to_host");
    check(np_copy_to_host(c, NULL, B.out.h + o_reads_a, reads_a, (size_t)n * sizeof(np_read_dev)), "np_copy_to_host");
    check(np_copy_to_host(c, NULL, B.out.h + o_reads_b, reads_b, (size_t)n * sizeof(np_read_dev)), "np_copy_to_host");
    check(np_sync(c, NULL), "np_sync");

    // ---- rebuild the SquiggleReads (squiggle_read.cpp:186-336) and the EventAlignment rows (eventalign.cpp:774-812) ---------------------
    const char* P = B.out.h;
    const uint32_t* ev_start = (const uint32_t*)(P + o_ev_start); const float* ev_len = (const float*)(P + o_ev_len);
    const float* evm = (const float*)(P + o_ev_mean); const float* evs = (const float*)(P + o_ev_stdv);
    const np_read_dev* ra = (const np_read_dev*)(P + o_reads_a); const np_read_dev* rb = (const np_read_dev*)(P + o_reads_b);
    const int32_t *map_start = (const int32_t*)(P + o_map_start), *map_stop = (const int32_t*)(P + o_map_stop), *n_events = (const int32_t*)(P + o_n_events),
                  *n_pairs = (const int32_t*)(P + o_n_pairs), *calibrated = (const int32_t*)(P + o_calibrated), *out_ref = (const int32_t*)(P + o_out_ref),
                  *out_event = (const int32_t*)(P + o_out_event), *n_out = (const int32_t*)(P + o_n_out), *status = (const int32_t*)(P + o_status);
    const uint8_t* out_state = (const uint8_t*)(P + o_out_state);
    const double* epb = (const double*)(P + o_epb);
    (void)ev_start;
    #pragma omp parallel for schedule(dynamic)
    for (int q = 0; q < n; ++q) {
        NpRealignRead& R = reads[idx[q]];
        if (n_events[q] < 0 || status[q] != NP_EA_OK) { R.status = NP_REALIGN_HOST_PATH; continue; }    // NP_ED_INEXACT / overflow / a record the chain refuses
        std::shared_ptr<SquiggleRead> sr(new SquiggleRead());
        sr->read_name = R.read_name; sr->read_sequence = rna ? seq_t[idx[q]] : *R.read_sequence;
        sr->nucleotide_type = rna ? SRNT_RNA : SRNT_DNA; sr->read_type = SRT_TEMPLATE; sr->pore_type = PORETYPE_R9;
        sr->base_model[0] = pm; sr->base_model[1] = NULL;
        sr->sample_rate = R.sample_rate; sr->channel_id = 0; sr->sample_start_time = 0; sr->read_id = 0;
        sr->events_per_base[0] = sr->events_per_base[1] = 0.0;
        const int ne = n_events[q];
        const bool aligned = n_pairs[q] > 0;
        // scalings: MoM (set4(shift, scale, 0, 1), raw_loader.cpp:52-58), replaced by recalibrate_model's set4 when it ran
        sr->scalings[0].set4(ra[q].shift, ra[q].scale, 0.0, 1.0);
        // (also when it ran and the result failed the MIN_CALIBRATION_VAR gate: the reference leaves those scalings in place and clears the
        //  events, squiggle_read.cpp:316-323.  The device writes a read's recalibrated scalings whenever the fit ran and reports the gate in
        //  calibrated[]; "ran and failed" is therefore calibrated == 0 with var > 2.5 -- the method-of-moments input carries var = 1.)
        const bool recal_ran = aligned && (calibrated[q] || rb[q].var > 2.5);
        if (recal_ran) sr->scalings[0].set4(rb[q].shift, rb[q].scale, 0.0, rb[q].var);
        const bool keep = aligned && calibrated[q] && !(epb[q] > 5.0);
        if (keep) {
            sr->events[0].resize(ne);
            double start_time = 0;
            for (int e = 0; e < ne; ++e) {                                              // squiggle_read.cpp:243-249, in DETECTION order
                const int64_t o = event_off[q] + (rna ? ne - 1 - e : e);                // (the device arrays of an RNA read are reversed already)
                const float length_in_seconds = ev_len[o] / sr->sample_rate;
                const SquiggleEvent se = { evm[o], evs[o], start_time, length_in_seconds, logf(evs[o]) };
                sr->events[0][e] = se;
                start_time += length_in_seconds;
            }
            if (rna) std::reverse(sr->events[0].begin(), sr->events[0].end());           // :260-263
        }
        if (aligned) {
            const int64_t nk = rank_off[q + 1] - rank_off[q];
            sr->base_to_event_map.resize(nk);
            for (int64_t j = 0; j < nk; ++j) {
                sr->base_to_event_map[j].indices[0].start = map_start[rank_off[q] + j];
                sr->base_to_event_map[j].indices[0].stop = map_stop[rank_off[q] + j];
            }
            sr->events_per_base[0] = epb[q];
        }
        R.sr = sr;
        if (!keep) { R.status = NP_REALIGN_NO_EVENTS; continue; }
        const bam1_t* b = R.record;
        const std::string ref_name(hdr->target_name[b->core.tid]);
        const std::string& ref_seq = ref_seqs[idx[q]];
        const int ref_offset = b->core.pos;
        const bool rc = bam_is_rev(b);
        R.alignment.resize(n_out[q]);
        for (int t = 0; t < n_out[q]; ++t) {
            const int64_t o = out_off[q] + t;
            EventAlignment& ea = R.alignment[t];
            ea.ref_name = ref_name;
            ea.ref_position = out_ref[o] + ref_offset;
            ea.ref_kmer = ref_seq.substr(ea.ref_position - ref_offset, k);
            ea.read_idx = R.read_idx; ea.strand_idx = 0; ea.event_idx = out_event[o]; ea.rc = rc;
            ea.hmm_state = (char)out_state[o];
            if (ea.hmm_state != 'B') ea.model_kmer = rc ? pm->pmalphabet->reverse_complement(ea.ref_kmer) : ea.ref_kmer;   // HMMInputSequence::get_kmer(i, k, rc)
            else ea.model_kmer = std::string(k, 'N');
        }
    }
}

void np_realign_reads_batch(std::vector<NpRealignRead>& reads, const faidx_t* fai, const bam_hdr_t* hdr, int region_start, int region_end)
{
    if (reads.empty()) return;
    Buffers& B = buffers();
    std::lock_guard<std::mutex> g(B.lock);
    bool any_dna = false, any_rna = false;
    for (size_t i = 0; i < reads.size(); ++i) { if (reads[i].rna) any_rna = true; else any_dna = true; }
    if (any_dna) realign_group(reads, false, fai, hdr, region_start, region_end);
    if (any_rna) realign_group(reads, true, fai, hdr, region_start, region_end);
}
