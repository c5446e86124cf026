// np_batch_dropin.cpp -- the throughput binding on the reference side: call-methylation's per-record work for whole
// BamProcessor batches, on the device, double-buffered.
//
// In the reference every record of a batch runs, under `#pragma omp parallel for` (src/common/nanopolish_bam_processor.cpp:99-106),
//     calculate_methylation_for_read_from_bam            src/nanopolish_call_methylation.cpp:163-177
//       SquiggleRead sr(read_name, read_db)               load_from_raw: detect_events, MoM scalings, event alignment, event map,
//                                                         recalibrate_model, QC gates (src/nanopolish_squiggle_read.cpp:141-336)
//       calculate_methylation_for_read(..., sr, ...)      src/basemods/nanopolish_basemods.cpp:238-419
// This file is compiled INSIDE a nanopolish build (it includes nanopolish's headers) and splits that loop in three:
//   phase 1 (host, OpenMP over the records): what only the host can do -- the read's sequence and raw samples (ReadDB / slow5 /
//            fast5: the caller's NpBatchRead), the reference segment (faidx), the CIGAR -- packed into ONE pinned blob;
//   phase 2 (device): ONE upload of that blob, then the whole batch in seven enqueues through the C ABI (include/np_hmm.h)
//            np_cm_build_jobs_cigar_dev -> np_detect_events_dev -> np_mom_fill_dev -> np_event_align_dev ->
//            np_calibrate_resolve_dev -> np_cm_discard_degenerate_dev -> np_hmm_score_dev, then ONE read-back of the output blob;
//   phase 3 (host): one ScoredSite map per record from the scores, exactly the fields basemods.cpp:384-413 fills.
// NpBatchPipeline keeps two input and two output blobs (device + pinned host, persistent, growing on demand) and three streams --
// upload, compute (the context's own), read-back -- ordered by events, so the upload and phase 1 of batch k+1 and the read-back and
// phase 3 of batch k-1 run beside the kernels of batch k.  Scratch (events, alignments, event maps, work items) is single: the
// compute stream runs one batch at a time.
// oracle/Makefile builds the reference with this file in (`make -C oracle batch`), tests/test_gpu_batch_dropin.py feeds it
// the records of tests/golden/golden_reflevel.npz and expects the maps the unmodified reference produced;
// tests/bench_batch_dropin.py times it at BamProcessor-like batch sizes.  INTEGRATION.md section 2 shows the call site.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <omp.h>
#include "np_batch_dropin.h"
#include "nanopolish_alphabet.h"
#include "nanopolish_eventalign.h"          // get_reference_region_ts
#include "nanopolish_pore_model_set.h"
#include "np_hmm.h"
#include "np_shim_common.h"

using np_shim::shim;
using np_shim::check;
using np_shim::die;
using np_shim::Layout;
using np_shim::Blob;

namespace {

int g_event_cap_divisor = 2;

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

// what one batch in flight needs on the host until it is collected
struct Slot {
    Blob in, out;
    void *ev_h2d, *ev_cmp, *ev_d2h;
    std::vector<NpBatchRead>* reads;
    std::vector<std::string> ref_seqs;
    std::vector<int> ref_start;
    std::vector<int64_t> group_off;
    std::vector<int> dev_index;             // batch index -> position among the records that went to the device, or -1
    int n_dev;
    // offsets into `out`
    size_t o_scores, o_first, o_last, o_n_motif, o_n_groups, o_n_events, o_n_pairs, o_calibrated, out_bytes;
    Slot() : in(true), out(true), ev_h2d(NULL), ev_cmp(NULL), ev_d2h(NULL), reads(NULL), n_dev(0) {}
};

} // namespace

struct NpBatchPipeline::Impl {
    np_ctx* c;
    MethylationCallingParameters params;
    std::string kit;
    const faidx_t* fai; const bam_hdr_t* hdr;
    int region_start, region_end;
    Slot slot[2];
    Blob scratch;
    void *s_h2d, *s_d2h;
    long n_submitted, n_collected;
    double t[6];
    Impl() : c(NULL), fai(NULL), hdr(NULL), region_start(-1), region_end(-1), scratch(false), s_h2d(NULL), s_d2h(NULL), n_submitted(0), n_collected(0) { for (int i = 0; i < 6; ++i) t[i] = 0.0; }
};

NpBatchPipeline::NpBatchPipeline(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                                 const bam_hdr_t* hdr, int region_start, int region_end) : p(new Impl())
{
    p->c = shim().get();
    p->s_h2d = np_stream_create(p->c); p->s_d2h = np_stream_create(p->c);
    if (!p->s_h2d || !p->s_d2h) die(np_last_error(p->c));
    for (int i = 0; i < 2; ++i) {
        Slot& s = p->slot[i];
        s.ev_h2d = np_event_create(p->c); s.ev_cmp = np_event_create(p->c); s.ev_d2h = np_event_create(p->c);
        if (!s.ev_h2d || !s.ev_cmp || !s.ev_d2h) die(np_last_error(p->c));
    }
    configure(calling_parameters, kit, fai, hdr, region_start, region_end);
}

NpBatchPipeline::~NpBatchPipeline()
{
    np_ctx* c = p->c;
    (void)np_sync(c, p->s_h2d); (void)np_sync(c, NULL); (void)np_sync(c, p->s_d2h);
    for (int i = 0; i < 2; ++i) {
        Slot& s = p->slot[i];
        s.in.release(c); s.out.release(c);
        np_event_destroy(c, s.ev_h2d); np_event_destroy(c, s.ev_cmp); np_event_destroy(c, s.ev_d2h);
    }
    p->scratch.release(c);
    np_stream_destroy(c, p->s_h2d); np_stream_destroy(c, p->s_d2h);
    delete p;
}

void NpBatchPipeline::configure(const MethylationCallingParameters& calling_parameters, const std::string& kit, const faidx_t* fai,
                                const bam_hdr_t* hdr, int region_start, int region_end)
{
    if (in_flight() != 0) die("NpBatchPipeline::configure with batches in flight");
    p->params = calling_parameters; p->kit = kit; p->fai = fai; p->hdr = hdr; p->region_start = region_start; p->region_end = region_end;
}

int NpBatchPipeline::in_flight() const { return (int)(p->n_submitted - p->n_collected); }
void NpBatchPipeline::host_seconds(double out[6]) const { for (int i = 0; i < 6; ++i) out[i] = p->t[i]; }

void NpBatchPipeline::submit(std::vector<NpBatchRead>& reads)
{
    if (in_flight() >= 2) die("NpBatchPipeline::submit: two batches are in flight already (collect one first)");
    np_ctx* c = p->c;
    Slot& S = p->slot[p->n_submitted & 1];
    p->n_submitted += 1;
    S.reads = &reads;
    const int n_all = (int)reads.size();
    S.ref_seqs.assign(n_all, std::string()); S.ref_start.assign(n_all, 0); S.dev_index.assign(n_all, -1);
    S.group_off.assign(1, 0); S.out_bytes = 0; S.n_dev = 0;
    if (n_all == 0) return;

    // the strand's models as load_from_raw and calculate_methylation_for_read choose them for a DNA read (squiggle_read.cpp:197-218,
    // basemods.cpp:276-287): strand "template", k = 6 from the base model.  Reads the device pass is not built for (RNA: another kit,
    // k = 5, another detector) take the caller's host path; a kit without a motif model leaves every map empty, as the reference.
    const char* strand_name = "template";
    const uint32_t k = 6;
    const PoreModel* pm_nuc = PoreModelSet::has_model(p->kit, "nucleotide", strand_name, k) ? PoreModelSet::get_model(p->kit, "nucleotide", strand_name, k) : NULL;
    const bool have_meth = pm_nuc && PoreModelSet::has_model(p->kit, p->params.methylation_type, strand_name, k);
    const int alphabet = np_alphabet_id(p->params.methylation_type.c_str());
    const bool device_ok = pm_nuc && pm_nuc->k == k && alphabet >= 1 && alphabet <= 4;
    const int MINSEP = p->params.min_separation, FLANK = p->params.min_flank;

    // ---- phase 1a: which records go to the device, their reference segments and sizes ---------------------------------------
    double tm = omp_get_wtime();
    std::vector<int> idx;                         // device order -> batch index
    for (int i = 0; i < n_all; ++i) {
        reads[i].status = NP_BATCH_OK;
        const bool fits = device_ok && !reads[i].rna && reads[i].record && reads[i].read_sequence && reads[i].read_sequence->size() >= k &&
                          (reads[i].raw_pa || reads[i].raw_adc) && reads[i].n_raw >= 64;
        if (!fits) { reads[i].status = NP_BATCH_HOST_PATH; continue; }
        if (!have_meth) continue;                                            // an empty map (basemods.cpp:280-287)
        S.dev_index[i] = (int)idx.size(); idx.push_back(i);
    }
    const int n = (int)idx.size();
    S.n_dev = n;
    if (n == 0) return;
    // The reference fetches every record's segment on its own, under a critical section (get_reference_region_ts,
    // src/alignment/nanopolish_eventalign.cpp:207-221: faidx_fetch_seq is not thread-safe) -- serial work per record.  The records
    // of a BamProcessor batch come from a sorted BAM: when the batch's records on one contig cover a compact stretch, ONE fetch of
    // their union serves them all (faidx clips a range to the contig, so a slice of the clipped union is what the clipped
    // per-record fetch returns); a scattered batch keeps the per-record fetches.
    std::map<int, std::pair<int, int> > span;           // tid -> [lowest pos, highest end] of the batch's records
    std::map<int, int64_t> covered;
    bool all_adc = true;
    for (int q = 0; q < n; ++q) {
        const bam1_t* record = reads[idx[q]].record;
        const int lo = record->core.pos, hi = bam_endpos(record);
        S.ref_start[idx[q]] = lo;
        std::map<int, std::pair<int, int> >::iterator it = span.find(record->core.tid);
        if (it == span.end()) span[record->core.tid] = std::make_pair(lo, hi);
        else { it->second.first = std::min(it->second.first, lo); it->second.second = std::max(it->second.second, hi); }
        covered[record->core.tid] += hi - lo + 1;
        all_adc = all_adc && reads[idx[q]].raw_adc != NULL;
    }
    std::map<int, std::string> union_seq;
    for (std::map<int, std::pair<int, int> >::const_iterator it = span.begin(); it != span.end(); ++it) {
        const int64_t len = (int64_t)it->second.second - it->second.first + 1;
        if (len <= (64 << 20) && len <= 4 * covered[it->first] + (1 << 20)) {
            int fetched_len = 0;
            union_seq[it->first] = get_reference_region_ts(p->fai, p->hdr->target_name[it->first], it->second.first, it->second.second, &fetched_len);
        }
    }
    #pragma omp parallel for schedule(dynamic, 16)
    for (int q = 0; q < n; ++q) {
        const int i = idx[q];
        const bam1_t* record = reads[i].record;
        std::map<int, std::string>::const_iterator u = union_seq.find(record->core.tid);
        if (u != union_seq.end()) {
            const int64_t off = (int64_t)record->core.pos - span[record->core.tid].first, want = (int64_t)bam_endpos(record) - record->core.pos + 1;
            const int64_t have = (int64_t)u->second.size() - off;
            S.ref_seqs[i] = have > 0 ? u->second.substr((size_t)off, (size_t)std::min(want, have)) : std::string();
        } else {
            int fetched_len = 0;
            S.ref_seqs[i] = get_reference_region_ts(p->fai, p->hdr->target_name[record->core.tid], record->core.pos, bam_endpos(record), &fetched_len);   // :258-270
        }
        // Alphabet::disambiguate (upper-casing + IUPAC codes -> their first base) is the identity on an upper-case ACGT string, and
        // it builds one std::string per character: 0.3 ms of a host core per 5 kb read.  Only a segment that needs it gets it.
        if (!is_plain_acgt(S.ref_seqs[i])) S.ref_seqs[i] = gDNAAlphabet.disambiguate(S.ref_seqs[i]);
    }
    std::vector<int64_t> raw_off(n + 1, 0), event_off(n + 1, 0), rank_off(n + 1, 0), cigar_off(n + 1, 0), jr_off(n + 1, 0),
                         pair_off(n + 1, 0), genome_off(n + 1, 0);
    std::vector<int64_t>& group_off = S.group_off;
    group_off.assign(n + 1, 0);
    for (int q = 0; q < n; ++q) {
        const int i = idx[q];
        const bam1_t* record = reads[i].record;
        const int64_t n_raw = (int64_t)reads[i].n_raw, L = (int64_t)reads[i].read_sequence->size(), ln = (int64_t)S.ref_seqs[i].size();
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
    S.o_scores = lo.add((size_t)n_jobs * sizeof(float)); S.o_first = lo.add((size_t)n_slots * 4); S.o_last = lo.add((size_t)n_slots * 4);
    S.o_n_motif = lo.add((size_t)n_slots * 4); S.o_n_groups = lo.add((size_t)n * 4); S.o_n_events = lo.add((size_t)n * 4);
    S.o_n_pairs = lo.add((size_t)n * 4); S.o_calibrated = lo.add((size_t)n * 4);
    S.out_bytes = lo.size;
    Layout ls;       // the part of the scratch that must start a batch zeroed comes first
    const size_t s_pair_begin = ls.add((size_t)n * 4), s_deg = ls.add((size_t)n * 8), s_kpos = ls.add((size_t)n_jobs * 8),
                 s_epb = ls.add((size_t)n * 8), s_jobs = ls.add((size_t)n_jobs * sizeof(np_hmm_job_dev));
    const size_t zero_bytes = ls.size;
    const size_t s_raw_pa = ls.add(all_adc ? (size_t)raw_off[n] * sizeof(float) : 0);
    const size_t s_tstat = ls.add((size_t)(2 * raw_off[n] + 16) * sizeof(float)), s_ev_len = ls.add((size_t)n_ev * 4), s_ev_mean = ls.add((size_t)n_ev * 4),
                 s_ev_stdv = ls.add((size_t)n_ev * 4), s_ev_start = ls.add((size_t)n_ev * 4), s_map_start = ls.add((size_t)n_rk * 4),
                 s_map_stop = ls.add((size_t)n_rk * 4), s_pairs = ls.add((size_t)pair_off[n] * sizeof(np_pair)),
                 s_job_ranks = ls.add((size_t)jr_off[n] * sizeof(uint16_t));
    p->t[0] += omp_get_wtime() - tm; tm = omp_get_wtime();
    S.in.reserve(c, li.size + 256); S.out.reserve(c, lo.size + 256);
    if (ls.size + 256 > p->scratch.cap) {
        check(np_sync(c, NULL), "np_sync");                  // the batch in flight still computes in the scratch that is about to be replaced
        p->scratch.reserve(c, ls.size + 256);
    }
    p->t[5] += omp_get_wtime() - tm; tm = omp_get_wtime();

    // ---- phase 1b: pack the pinned input blob -----------------------------------------------------------------------------
    char* H = S.in.h;
    float* h_raw = (float*)(H + i_raw); uint16_t* h_ranks = (uint16_t*)(H + i_ranks);
    np_read_dev* h_reads_a = (np_read_dev*)(H + i_reads_a); np_read_dev* h_reads_b = (np_read_dev*)(H + i_reads_b);
    char* h_genome = H + i_genome; int64_t* h_ref_begin = (int64_t*)(H + i_ref_begin); int32_t* h_ref_len = (int32_t*)(H + i_ref_len);
    int32_t* h_read_len = (int32_t*)(H + i_read_len); uint32_t* h_cigar = (uint32_t*)(H + i_cigar); uint8_t* h_rc = (uint8_t*)(H + i_rc);
    #pragma omp parallel for schedule(dynamic)
    for (int q = 0; q < n; ++q) {
        const int i = idx[q];
        const bam1_t* record = reads[i].record;
        const std::string& seq = *reads[i].read_sequence;
        for (int t = 0; t < 2; ++t)
            np_fill_read_host(t ? &h_reads_b[q] : &h_reads_a[q], 0.0, 1.0, 1.0, event_off[q], (uint32_t)(event_off[q + 1] - event_off[q]), rank_off[q],
                              (uint32_t)(rank_off[q + 1] - rank_off[q]));
        ((float*)(H + i_adc_offset))[q] = reads[i].adc_offset; ((float*)(H + i_adc_unit))[q] = reads[i].adc_raw_unit;
        if (all_adc) memcpy((int16_t*)(H + i_raw) + raw_off[q], reads[i].raw_adc, reads[i].n_raw * sizeof(int16_t));
        else if (reads[i].raw_pa) memcpy(h_raw + raw_off[q], reads[i].raw_pa, reads[i].n_raw * sizeof(float));
        else for (size_t t = 0; t < reads[i].n_raw; ++t)            // the loader's conversion, fp32 (fast5_loader.cpp:96-103)
            h_raw[raw_off[q] + t] = ((float)reads[i].raw_adc[t] + reads[i].adc_offset) * reads[i].adc_raw_unit;
        nucleotide_kmer_ranks(seq, k, h_ranks + rank_off[q]);
        memcpy(h_genome + genome_off[q], S.ref_seqs[i].data(), S.ref_seqs[i].size());
        h_ref_begin[q] = genome_off[q]; h_ref_len[q] = (int32_t)S.ref_seqs[i].size();
        memcpy(h_cigar + cigar_off[q], bam_get_cigar(record), 4 * (size_t)record->core.n_cigar);
        h_read_len[q] = (int32_t)seq.size();
        h_rc[q] = bam_is_rev(record) ? 1 : 0;
    }
    memcpy(H + i_raw_off, raw_off.data(), (size_t)(n + 1) * 8); memcpy(H + i_event_off, event_off.data(), (size_t)(n + 1) * 8);
    memcpy(H + i_cigar_off, cigar_off.data(), (size_t)(n + 1) * 8); memcpy(H + i_group_off, group_off.data(), (size_t)(n + 1) * 8);
    memcpy(H + i_jr_off, jr_off.data(), (size_t)(n + 1) * 8); memcpy(H + i_pair_off, pair_off.data(), (size_t)(n + 1) * 8);

    p->t[1] += omp_get_wtime() - tm; tm = omp_get_wtime();
    // ---- phase 2: one upload, the batch on the device, one read-back -----------------------------------------------------------
    const int m_nuc = shim().model_id(pm_nuc);
    const int m_meth = shim().model_id(PoreModelSet::get_model(p->kit, p->params.methylation_type, strand_name, k));
    check(np_copy_to_device(c, p->s_h2d, S.in.d, S.in.h, li.size), "np_copy_to_device");
    check(np_event_record(c, S.ev_h2d, p->s_h2d), "np_event_record");
    check(np_stream_wait_event(c, NULL, S.ev_h2d), "np_stream_wait_event");
    check(np_memset_dev(c, NULL, S.out.d, 0, lo.size), "np_memset_dev");
    check(np_memset_dev(c, NULL, p->scratch.d, 0, zero_bytes), "np_memset_dev");
    {
        char* D = S.in.d; char* O = S.out.d; char* X = p->scratch.d;
        float* raw = all_adc ? (float*)(p->scratch.d + s_raw_pa) : (float*)(D + i_raw); uint16_t* ranks = (uint16_t*)(D + i_ranks);
        np_read_dev* reads_a = (np_read_dev*)(D + i_reads_a); np_read_dev* reads_b = (np_read_dev*)(D + i_reads_b);
        int64_t *d_raw_off = (int64_t*)(D + i_raw_off), *d_event_off = (int64_t*)(D + i_event_off), *d_cigar_off = (int64_t*)(D + i_cigar_off),
                *d_group_off = (int64_t*)(D + i_group_off), *d_jr_off = (int64_t*)(D + i_jr_off), *d_pair_off = (int64_t*)(D + i_pair_off),
                *ref_begin = (int64_t*)(D + i_ref_begin);
        int32_t *ref_len = (int32_t*)(D + i_ref_len), *read_len = (int32_t*)(D + i_read_len);
        uint32_t* cigar = (uint32_t*)(D + i_cigar); uint8_t* rc = (uint8_t*)(D + i_rc); char* genome = D + i_genome;
        float* scores = (float*)(O + S.o_scores);
        int32_t *first = (int32_t*)(O + S.o_first), *last = (int32_t*)(O + S.o_last), *n_motif = (int32_t*)(O + S.o_n_motif),
                *n_groups = (int32_t*)(O + S.o_n_groups), *n_events = (int32_t*)(O + S.o_n_events), *n_pairs = (int32_t*)(O + S.o_n_pairs),
                *calibrated = (int32_t*)(O + S.o_calibrated);
        int32_t *pair_begin = (int32_t*)(X + s_pair_begin), *deg = (int32_t*)(X + s_deg), *kpos = (int32_t*)(X + s_kpos),
                *map_start = (int32_t*)(X + s_map_start), *map_stop = (int32_t*)(X + s_map_stop);
        double* epb = (double*)(X + s_epb);
        np_hmm_job_dev* jobs = (np_hmm_job_dev*)(X + s_jobs);
        float *tstat = (float*)(X + s_tstat), *ev_len = (float*)(X + s_ev_len), *ev_mean = (float*)(X + s_ev_mean), *ev_stdv = (float*)(X + s_ev_stdv);
        uint32_t* ev_start = (uint32_t*)(X + s_ev_start);
        np_pair* pairs = (np_pair*)(X + s_pairs);
        uint16_t* job_ranks = (uint16_t*)(X + s_job_ranks);
        np_detector_param prm;
        np_event_detection_params(&prm, 0);
        check(np_cm_build_jobs_cigar_dev(c, NULL, n, genome, ref_begin, ref_len, cigar, d_cigar_off, cigar_off[n], read_len, rc, alphabet, k, MINSEP,
                                         FLANK, d_group_off, n_slots, d_jr_off, jobs, kpos, job_ranks, first, last, n_motif, n_groups, deg),
              "np_cm_build_jobs_cigar_dev");
        if (all_adc)
            check(np_adc_to_pa_dev(c, NULL, n, (const int16_t*)(D + i_raw), d_raw_off, max_samples, (const float*)(D + i_adc_offset),
                                   (const float*)(D + i_adc_unit), raw), "np_adc_to_pa_dev");
        check(np_detect_events_dev(c, NULL, n, raw, d_raw_off, max_samples, &prm, tstat, d_event_off, max_events, ev_start, ev_len, ev_mean,
                                   ev_stdv, n_events), "np_detect_events_dev");
        check(np_mom_fill_dev(c, NULL, n, reads_a, reads_b, ev_mean, n_events, ranks, m_nuc), "np_mom_fill_dev");
        check(np_event_align_dev(c, NULL, n, reads_a, ev_mean, ranks, m_nuc, max_bands, d_pair_off, pairs, pair_begin, n_pairs), "np_event_align_dev");
        check(np_calibrate_resolve_dev(c, NULL, n, reads_b, ev_mean, ranks, m_nuc, d_pair_off, pairs, pair_begin, n_pairs, map_start, map_stop, epb,
                                       calibrated, n_jobs, jobs, kpos), "np_calibrate_resolve_dev");
        check(np_cm_discard_degenerate_dev(c, NULL, reads_b, map_start, deg, n_jobs, jobs), "np_cm_discard_degenerate_dev");
        check(np_hmm_score_dev(c, NULL, n_jobs, jobs, reads_b, ev_mean, job_ranks, m_meth, scores), "np_hmm_score_dev");
    }
    check(np_event_record(c, S.ev_cmp, NULL), "np_event_record");
    check(np_stream_wait_event(c, p->s_d2h, S.ev_cmp), "np_stream_wait_event");
    check(np_copy_to_host(c, p->s_d2h, S.out.h, S.out.d, lo.size), "np_copy_to_host");
    check(np_event_record(c, S.ev_d2h, p->s_d2h), "np_event_record");
    p->t[2] += omp_get_wtime() - tm;
}

bool NpBatchPipeline::collect(MethylationCallingResult& result)
{
    if (in_flight() <= 0) return false;
    np_ctx* c = p->c;
    Slot& S = p->slot[p->n_collected & 1];
    p->n_collected += 1;
    std::vector<NpBatchRead>& reads = *S.reads;
    const int n = (int)reads.size();
    if (n == 0) return true;
    double tm = omp_get_wtime();
    if (S.n_dev > 0) check(np_event_sync(c, S.ev_d2h), "np_event_sync");
    p->t[3] += omp_get_wtime() - tm; tm = omp_get_wtime();
    const char* O = S.out.h;
    const float* scores = (const float*)(O + S.o_scores);
    const int32_t *first = (const int32_t*)(O + S.o_first), *last = (const int32_t*)(O + S.o_last), *n_motif = (const int32_t*)(O + S.o_n_motif),
                  *n_groups = (const int32_t*)(O + S.o_n_groups), *n_events = (const int32_t*)(O + S.o_n_events),
                  *n_pairs = (const int32_t*)(O + S.o_n_pairs), *calibrated = (const int32_t*)(O + S.o_calibrated);
    const uint32_t k = 6;

    // ---- phase 3: ScoredSite maps (basemods.cpp:384-413) ------------------------------------------------------------------
    // the per-record maps are created serially (result is one std::map), then filled in parallel: records are independent
    std::vector<std::map<int, ScoredSite>*> maps(n, (std::map<int, ScoredSite>*)NULL);
    for (int i = 0; i < n; ++i) {
        if (reads[i].status == NP_BATCH_HOST_PATH) continue;            // decided in phase 1: the caller's per-record function fills its map
        maps[i] = &result[reads[i].record];                              // the (possibly empty) map of the record, basemods.cpp:253-256
    }
    #pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; ++i) {
        const bam1_t* record = reads[i].record;
        if (!maps[i]) continue;
        std::map<int, ScoredSite>& site_score_map = *maps[i];
        const int q = S.dev_index[i];
        if (q < 0) continue;                                             // no motif model for the kit: the map stays empty
        if (n_events[q] < 0 || n_groups[q] < 0) { reads[i].status = NP_BATCH_HOST_PATH; continue; }   // NP_ED_INEXACT / NP_ED_OVERFLOW / capacity
        if (n_pairs[q] <= 0 || !calibrated[q]) { reads[i].status = NP_BATCH_NO_EVENTS; continue; }
        const std::string contig = p->hdr->target_name[record->core.tid];
        const std::string& ref_seq = S.ref_seqs[i];
        const int strand_idx = 0;
        for (int g = 0; g < n_groups[q]; ++g) {
            const int64_t slot = S.group_off[q] + g;
            const float unmethylated_score = scores[2 * slot], methylated_score = scores[2 * slot + 1];
            if (unmethylated_score != unmethylated_score || methylated_score != methylated_score) continue;   // a group the caller rules skip
            const int start_position = first[slot] + S.ref_start[i];
            const int end_position = last[slot] + S.ref_start[i];
            if ((p->region_start != -1 && start_position < p->region_start) || (p->region_end != -1 && end_position >= p->region_end)) continue;
            std::map<int, ScoredSite>::iterator iter = site_score_map.find(start_position);
            if (iter == site_score_map.end()) {
                ScoredSite ss;
                ss.chromosome = contig;
                ss.start_position = start_position;
                ss.end_position = end_position;
                ss.n_motif = n_motif[slot];
                const size_t site_output_start = first[slot] - k + 1, site_output_end = last[slot] + k;
                ss.sequence = ref_seq.substr(site_output_start, site_output_end - site_output_start);
                iter = site_score_map.insert(std::make_pair(start_position, ss)).first;
            }
            iter->second.ll_unmethylated[strand_idx] = unmethylated_score;
            iter->second.ll_methylated[strand_idx] = methylated_score;
            iter->second.strands_scored += 1;
        }
    }
    p->t[4] += omp_get_wtime() - tm;
    return true;
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
    if (!pipe) pipe = new NpBatchPipeline(params, kit, fai, hdr, region_start, region_end);
    else pipe->configure(params, kit, fai, hdr, region_start, region_end);
    pipe->submit(reads);
    pipe->collect(result);
}
