// np_host.cpp -- host-side (CPU, no device) parts of the C ABI: alphabets, k-mer ranks, methylation-aware
// string transforms, transitions, MoM scaling estimate, motif grouping.  These mirror what the reference does
// on the host *around* the two kernels, so that a caller (or nanopolish_amd/csrc/np_dropin.cpp) can flatten
// HMMInputSequence/HMMInputData into np_hmm_job without linking any reference code.
//
// Reference: src/common/nanopolish_alphabet.{h,cpp}, src/hmm/nanopolish_hmm_input_sequence.h,
// src/hmm/nanopolish_profile_hmm_r9.inl:17-76, src/nanopolish_raw_loader.cpp:17-60,99-108,
// src/basemods/nanopolish_basemods.cpp:298-320.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/np_hmm.h"
#include "np_logf.h"
#include "np_log.h"

namespace {

struct Alpha {
    const char* name;
    const char* bases;        // rank order
    const char* comp;         // complement by rank
    int n_sites;
    int site_len;
    const char* site[2];
    const char* site_m[2];
    const char* site_mc[2];
    uint8_t rank[256];
};

Alpha make(const char* name, const char* bases, const char* comp, int n_sites, int site_len,
           const char* s0, const char* m0, const char* c0, const char* s1 = nullptr, const char* m1 = nullptr, const char* c1 = nullptr)
{
    Alpha a{};
    a.name = name; a.bases = bases; a.comp = comp; a.n_sites = n_sites; a.site_len = site_len;
    a.site[0] = s0; a.site_m[0] = m0; a.site_mc[0] = c0; a.site[1] = s1; a.site_m[1] = m1; a.site_mc[1] = c1;
    memset(a.rank, 0, sizeof(a.rank));                      // every other byte ranks 0 (the _rank[256] tables)
    for (int i = 0; bases[i]; ++i) a.rank[(uint8_t)bases[i]] = (uint8_t)i;
    return a;
}

const Alpha& alpha(int id)
{
    static const Alpha A[6] = {
        make("nucleotide", "ACGT", "TGCA", 0, 0, nullptr, nullptr, nullptr),
        make("cpg", "ACGMT", "TGCGA", 1, 2, "CG", "MG", "GM"),
        make("gpc", "ACGMT", "TGCGA", 1, 2, "GC", "GM", "MG"),
        make("dam", "ACGMT", "TGCTA", 1, 4, "GATC", "GMTC", "CTMG"),
        make("dcm", "ACGMT", "TGCGA", 2, 5, "CCAGG", "CMAGG", "GGTMC", "CCTGG", "CMTGG", "GGAMC"),
        make("u_to_t_rna", "ACGT", "TGCA", 0, 0, nullptr, nullptr, nullptr),
    };
    return A[id];
}

struct Match { int offset = 0, length = 0; bool covers_m = false; };

// match_to_site semantics (nanopolish_alphabet.h:27-56): (1) at i == 0 the whole string may be an infix of the
// site; otherwise (2) the suffix starting at i is compared with the site's prefix, truncated at the string end.
Match site_match(const char* s, size_t n, size_t i, const char* site, size_t rl)
{
    Match m;
    bool infix = false;
    if (i == 0 && n <= rl) {
        for (size_t o = 0; o + n <= rl && !infix; ++o)
            if (memcmp(site + o, s, n) == 0) { infix = true; m.offset = (int)o; m.length = (int)n; }
    }
    if (!infix) {
        const size_t cl = std::min(rl, n - i);
        if (memcmp(s + i, site, cl) == 0) { m.offset = 0; m.length = (int)cl; }
    }
    for (int j = 0; j < m.length; ++j) m.covers_m |= (s[i + j] == 'M');
    return m;
}

} // namespace

extern "C" {

void np_default_params(np_params* p)
{
    p->hmm_indel_bias_factor = 1.0;
    p->min_average_log_emission = -5.0;
    p->max_gap_threshold = 50;
    p->reserved = 0;
}

int np_alphabet_id(const char* name)
{
    for (int i = 0; i < 6; ++i) if (strcmp(alpha(i).name, name) == 0) return i;
    return NP_ERR_INVALID;
}

uint32_t np_alphabet_size(int a) { return (uint32_t)strlen(alpha(a).bases); }

uint32_t np_kmer_rank(int a, const char* kmer, uint32_t k)
{
    const Alpha& A = alpha(a);
    const uint32_t sz = (uint32_t)strlen(A.bases);
    uint32_t r = 0;
    for (uint32_t i = 0; i < k; ++i) r = r * sz + A.rank[(uint8_t)kmer[i]];   // == sum rank(str[k-i-1]) * size^i
    return r;
}

int np_reverse_complement(int a, const char* in, size_t n, char* out)
{
    const Alpha& A = alpha(a);
    size_t i = 0;
    ptrdiff_t j = (ptrdiff_t)n - 1;
    while (i < n) {
        int hit = -1;
        Match m;
        for (int s = 0; s < A.n_sites; ++s) {
            m = site_match(in, n, i, A.site_m[s], A.site_len);
            if (m.length > 0 && m.covers_m) { hit = s; break; }
        }
        if (hit >= 0) {
            for (int q = m.offset; q < m.offset + m.length; ++q) { out[j--] = A.site_mc[hit][q]; ++i; }
        } else {
            out[j--] = A.comp[A.rank[(uint8_t)in[i++]]];
        }
    }
    out[n] = 0;
    return NP_OK;
}

int np_methylate(int a, const char* in, size_t n, char* out)
{
    const Alpha& A = alpha(a);
    memcpy(out, in, n); out[n] = 0;
    for (size_t i = 0; i < n;) {
        size_t step = 1;
        for (int s = 0; s < A.n_sites; ++s) {
            const Match m = site_match(in, n, i, A.site[s], A.site_len);
            if (m.length == A.site_len) { memcpy(out + i, A.site_m[s], A.site_len); step = m.length; break; }
        }
        i += step;
    }
    return NP_OK;
}

int np_unmethylate(int a, const char* in, size_t n, char* out)
{
    const Alpha& A = alpha(a);
    memcpy(out, in, n); out[n] = 0;
    for (size_t i = 0; i < n;) {
        size_t step = 1;
        for (int s = 0; s < A.n_sites; ++s) {
            const Match m = site_match(in, n, i, A.site_m[s], A.site_len);
            if (m.length > 0) { memcpy(out + i, A.site[s] + m.offset, m.length); step = m.length; break; }
        }
        i += step;
    }
    return NP_OK;
}

int np_is_motif_match(int a, const char* str, size_t n, size_t i)
{
    const Alpha& A = alpha(a);
    for (int s = 0; s < A.n_sites; ++s)
        if (site_match(str, n, i, A.site[s], A.site_len).length == A.site_len) return 1;
    return 0;
}

int np_sequence_kmer_ranks(int a, const char* seq, const char* rc_seq, size_t n, uint32_t k, int do_rc, uint16_t* out)
{
    if (n < k) return NP_ERR_INVALID;
    std::string tmp;
    if (do_rc && !rc_seq) { tmp.resize(n + 1); np_reverse_complement(a, seq, n, &tmp[0]); rc_seq = tmp.c_str(); }
    const size_t nk = n - k + 1;
    for (size_t i = 0; i < nk; ++i)
        out[i] = (uint16_t)(!do_rc ? np_kmer_rank(a, seq + i, k) : np_kmer_rank(a, rc_seq + (n - i - k), k));
    return NP_OK;
}

void np_calculate_transitions(double events_per_base, double indel_bias, float out[10])
{
    np_transitions(events_per_base, indel_bias, out);
}

void np_estimate_scalings_mom(const double* level_mean, const uint16_t* ranks, uint32_t n_kmers,
                              const float* event_mean, uint32_t n_events, double* shift_out, double* scale_out)
{
    double event_level_sum = 0.0;
    for (uint32_t i = 0; i < n_events; ++i) event_level_sum += event_mean[i];
    double kmer_level_sum = 0.0, kmer_level_sq_sum = 0.0;
    for (uint32_t i = 0; i < n_kmers; ++i) {
        const double l = level_mean[ranks[i]];
        kmer_level_sum += l;
        kmer_level_sq_sum += pow(l, 2.0);
    }
    const double shift = event_level_sum / n_events - kmer_level_sum / n_kmers;
    double event_level_sq_sum = 0.0;
    for (uint32_t i = 0; i < n_events; ++i) event_level_sq_sum += pow(event_mean[i] - shift, 2.0);
    *shift_out = shift;
    *scale_out = (event_level_sq_sum / n_events) / (kmer_level_sq_sum / n_kmers);
}

int np_scan_motif_groups(int a, const char* ref, size_t n, int min_separation,
                         int32_t* first_site, int32_t* last_site, int32_t* n_motif, int cap)
{
    std::vector<int> sites;
    for (size_t i = 0; i + 1 < n; ++i) if (np_is_motif_match(a, ref, n, i)) sites.push_back((int)i);
    int ng = 0;
    for (size_t c = 0; c < sites.size();) {
        size_t e = c + 1;
        while (e < sites.size() && sites[e] - sites[e - 1] <= min_separation) ++e;
        if (ng < cap) { first_site[ng] = sites[c]; last_site[ng] = sites[e - 1]; n_motif[ng] = (int)(e - c); }
        ++ng;
        c = e;
    }
    return ng;
}

// Work items of calculate_methylation_for_read (src/basemods/nanopolish_basemods.cpp:298-378) for a read whose
// base-to-reference alignment is the identity (ref position p <-> read position p on the reference strand), which is
// how bench.py and the tests lay synthetic reads out.  Everything that does not need the event alignment is done
// here: motif scan, grouping, window, the boundary rule of EventAlignmentRecord (alignment_db.cpp:63-72) +
// _find_by_ref_bounds (:688-731), methylate / reverse-complement, k-mer ranks.  The event bounds themselves are
// resolved on the device from the k-mer positions written to kpos (np_resolve_jobs_dev).
// Returns the number of jobs written (<= cap_jobs), or NP_ERR_NOMEM if a capacity is too small.
int np_cm_build_jobs_identity(int alphabet, const char* ref_seq, size_t n, int read_rc, uint32_t k,
                              int min_separation, int min_flank, int cap_jobs, int64_t cap_ranks,
                              int32_t* first_site, int32_t* last_site, int32_t* n_motif,
                              int32_t* kpos /*2 per job*/, int32_t* job_n_kmers,
                              uint16_t* ranks_unmeth, uint16_t* ranks_meth, int64_t* rank_off /*cap_jobs+1*/)
{
    std::vector<int32_t> f(n + 1), l(n + 1), c(n + 1);
    const int ng = np_scan_motif_groups(alphabet, ref_seq, n, min_separation, f.data(), l.data(), c.data(), (int)n + 1);
    int nj = 0;
    int64_t w = 0;
    rank_off[0] = 0;
    std::string sub, rc_sub, m_sub, rc_m_sub;
    for (int g = 0; g < ng; ++g) {
        const int sub_start = f[g] - min_flank, sub_end = l[g] + min_flank, span = l[g] - f[g];
        if (sub_start <= min_separation || span > 200) continue;                       // basemods.cpp:334
        // aligned_events holds ref positions p with k <= p and p + k < n (alignment_db.cpp:65-71); both bounds must be
        // found exactly for the identity alignment (alignment_db.cpp:697-708)
        if (sub_start < (int)k || (size_t)sub_end + k >= n) continue;
        const size_t len = (size_t)(sub_end - sub_start + 1);
        const uint32_t nk = (uint32_t)(len - k + 1);
        if (nj >= cap_jobs || w + nk > cap_ranks) return NP_ERR_NOMEM;
        sub.assign(ref_seq + sub_start, len);
        rc_sub.resize(len + 1); m_sub.resize(len + 1); rc_m_sub.resize(len + 1);
        np_reverse_complement(alphabet, sub.c_str(), len, &rc_sub[0]);                 // basemods.cpp:339
        np_methylate(alphabet, sub.c_str(), len, &m_sub[0]);                           // :377
        np_reverse_complement(alphabet, m_sub.c_str(), len, &rc_m_sub[0]);             // :378
        np_sequence_kmer_ranks(alphabet, sub.c_str(), rc_sub.c_str(), len, k, read_rc, ranks_unmeth + w);
        np_sequence_kmer_ranks(alphabet, m_sub.c_str(), rc_m_sub.c_str(), len, k, read_rc, ranks_meth + w);
        first_site[nj] = f[g]; last_site[nj] = l[g]; n_motif[nj] = c[g];
        // read-strand k-mer positions of the window ends (flip_k_strand for reverse-strand reads, squiggle_read.h:229-233)
        kpos[2 * nj] = read_rc ? (int32_t)(n - sub_start - k) : sub_start;
        kpos[2 * nj + 1] = read_rc ? (int32_t)(n - sub_end - k) : sub_end;
        job_n_kmers[nj] = (int32_t)nk;
        w += nk;
        rank_off[++nj] = w;
    }
    return nj;
}

// ---------------------------------------------------------------------------------------------------------------------
// CIGAR-driven work items: the same as above for a read whose base-to-reference alignment
// comes from a BAM record's CIGAR.
// ---------------------------------------------------------------------------------------------------------------------
// get_aligned_segments (src/alignment/nanopolish_anchor.cpp:20-95) for a non-spliced record: M/=/X emit a pair and advance
// both, D advances the reference, I and S advance the read, H neither; N (a second segment) is what SequenceAlignmentRecord
// rejects (alignment_db.cpp:43-47) and P / B hit the reference's "Unhandled cigar operation" assert: NP_ERR_INVALID here.  ref positions start at ref_pos0.
int np_cigar_aligned_bases(const uint32_t* cigar, int n_cigar, int ref_pos0, int32_t* ref_pos, int32_t* read_pos, int cap)
{
    int n = 0, rp = ref_pos0, qp = 0;
    for (int ci = 0; ci < n_cigar; ++ci) {
        const int len = (int)(cigar[ci] >> 4), op = (int)(cigar[ci] & 0xf);
        const bool aligned = op == 0 || op == 7 || op == 8;
        if (op == 3 || op == 6 || op > 8) return NP_ERR_INVALID;     // N: second segment; P and beyond: the reference asserts
        const int ref_inc = (aligned || op == 2) ? 1 : 0, read_inc = (aligned || op == 1 || op == 4) ? 1 : 0;
        if (aligned) {
            for (int j = 0; j < len; ++j) { if (n < cap) { ref_pos[n] = rp + j; read_pos[n] = qp + j; } ++n; }
        }
        rp += ref_inc * len; qp += read_inc * len;
    }
    return n;
}

namespace {
// EventAlignmentRecord's filter (alignment_db.cpp:63-72) + AlignmentDB::_find_iter_by_ref_bounds (:688-711) on the aligned
// BASES: the aligned events are these pairs with read_pos replaced by an event index, so the bounding pairs are the same.
// Returns false when unbounded; else the read positions (reference strand) of the two bounding pairs.
bool cigar_find_bounds(const std::vector<int32_t>& rp, const std::vector<int32_t>& qp, int ref_start, int ref_stop, int& q1, int& q2)
{
    const auto b = rp.begin(), e = rp.end();
    const auto it1 = std::lower_bound(b, e, ref_start), it2 = std::lower_bound(b, e, ref_stop);
    if (it1 == e || it2 == e) return false;
    const bool left_bounded = *it1 <= ref_start || it1 != b;      // (start_iter - 1)->ref_pos <= ref_start always holds
    // right_bounded: stop_iter->ref_pos >= ref_stop holds for every lower_bound result
    if (!left_bounded) return false;
    q1 = qp[it1 - b]; q2 = qp[it2 - b];
    return true;
}
}

// Work items of calculate_methylation_for_read for a CIGAR-aligned read.
//   ref_seq[0..n)   the reference segment the reference fetches: contig[pos .. bam_endpos] inclusive, clipped to the contig
//                   (basemods.cpp:259-270), already disambiguated; window coordinates are relative to pos
//   cigar / read_len / read_rc   the record's CIGAR, SquiggleRead::read_sequence.length(), bam_is_rev
//   deg_kpos[2]     read-strand k-mer positions of the first and last aligned event: when their closest events coincide the
//                   reference discards the record (alignment_db.cpp:83-86); -1, -1 when the record has no aligned events
// Other outputs as np_cm_build_jobs_identity.  Returns the number of jobs, NP_ERR_NOMEM or NP_ERR_INVALID.
int np_cm_build_jobs_cigar(int alphabet, const char* ref_seq, size_t n, const uint32_t* cigar, int n_cigar,
                           int read_len, int read_rc, uint32_t k, int min_separation, int min_flank,
                           int cap_jobs, int64_t cap_ranks, int32_t* first_site, int32_t* last_site, int32_t* n_motif,
                           int32_t* kpos, int32_t* job_n_kmers, uint16_t* ranks_unmeth, uint16_t* ranks_meth,
                           int64_t* rank_off, int32_t* deg_kpos)
{
    const int n_al = np_cigar_aligned_bases(cigar, n_cigar, 0, nullptr, nullptr, 0);
    if (n_al < 0) return n_al;
    std::vector<int32_t> arp(n_al), aqp(n_al), rp, qp;
    np_cigar_aligned_bases(cigar, n_cigar, 0, arp.data(), aqp.data(), n_al);
    for (int i = 0; i < n_al; ++i)
        if (aqp[i] >= (int)k && aqp[i] + (int)k < read_len) { rp.push_back(arp[i]); qp.push_back(aqp[i]); }
    auto flip = [&](int q) { return read_rc ? read_len - q - (int)k : q; };     // flip_k_strand, squiggle_read.h:229-233
    deg_kpos[0] = rp.empty() ? -1 : flip(qp.front());
    deg_kpos[1] = rp.empty() ? -1 : flip(qp.back());

    std::vector<int32_t> f(n + 1), l(n + 1), c(n + 1);
    const int ng = np_scan_motif_groups(alphabet, ref_seq, n, min_separation, f.data(), l.data(), c.data(), (int)n + 1);
    int nj = 0;
    int64_t w = 0;
    rank_off[0] = 0;
    std::string sub, rc_sub, m_sub, rc_m_sub;
    for (int g = 0; g < ng; ++g) {
        const int sub_start = f[g] - min_flank, sub_end = l[g] + min_flank, span = l[g] - f[g];
        if (sub_start <= min_separation || span > 200) continue;                       // basemods.cpp:334
        int q1 = 0, q2 = 0;
        if (!cigar_find_bounds(rp, qp, sub_start, sub_end, q1, q2)) continue;          // :346-358, `bounded`
        if ((size_t)sub_end >= n) continue;          // cannot happen for a bounded window: the segment covers every aligned base
        const size_t len = (size_t)(sub_end - sub_start + 1);
        const uint32_t nk = (uint32_t)(len - k + 1);
        if (nj >= cap_jobs || w + nk > cap_ranks) return NP_ERR_NOMEM;
        sub.assign(ref_seq + sub_start, len);
        rc_sub.resize(len + 1); m_sub.resize(len + 1); rc_m_sub.resize(len + 1);
        np_reverse_complement(alphabet, sub.c_str(), len, &rc_sub[0]);
        np_methylate(alphabet, sub.c_str(), len, &m_sub[0]);
        np_reverse_complement(alphabet, m_sub.c_str(), len, &rc_m_sub[0]);
        np_sequence_kmer_ranks(alphabet, sub.c_str(), rc_sub.c_str(), len, k, read_rc, ranks_unmeth + w);
        np_sequence_kmer_ranks(alphabet, m_sub.c_str(), rc_m_sub.c_str(), len, k, read_rc, ranks_meth + w);
        first_site[nj] = f[g]; last_site[nj] = l[g]; n_motif[nj] = c[g];
        kpos[2 * nj] = flip(q1);
        kpos[2 * nj + 1] = flip(q2);
        job_n_kmers[nj] = (int32_t)nk;
        w += nk;
        rank_off[++nj] = w;
    }
    return nj;
}

void np_fill_read_host(np_read_dev* r, double shift, double scale, double var,
                       int64_t event_off, uint32_t n_events, int64_t rank_off, uint32_t n_kmers)
{
    memset(r, 0, sizeof(*r));
    r->scale = scale; r->shift = shift; r->var = var; r->log_var = log(var);   // set4/set6, squiggle_read.cpp:38-65
    // aligner transition constants, raw_loader.cpp:99-108 (double, host libm as in the reference)
    const double events_per_kmer = (double)n_events / n_kmers;
    const double p_stay = 1 - (1 / (events_per_kmer + 1));
    const double epsilon = 1e-10;
    r->lp_skip = log(epsilon);
    r->lp_stay = log(p_stay);
    r->lp_step = log(1.0 - exp(r->lp_skip) - exp(r->lp_stay));
    r->lp_trim = log(0.01);
    r->event_off = event_off; r->rank_off = rank_off; r->n_events = n_events; r->n_kmers = n_kmers;
}

// The aligner's per-read constants (raw_loader.cpp:99-108) exactly as np_mom_fill_dev computes them on the device: glibc's
// log / exp restated in np_log.h instead of libm calls.  Exposed so that the restatement can be checked against the
// host's libm without a GPU (tests/test_host_logic.py).
void np_aligner_constants(uint32_t n_events, uint32_t n_kmers, double out[4])
{
    const double events_per_kmer = (double)n_events / (double)n_kmers;
    const double p_stay = 1 - (1 / (events_per_kmer + 1));
    const double epsilon = 1e-10;
    out[0] = np_log_glibc(epsilon);
    out[1] = np_log_glibc(p_stay);
    out[2] = np_log_glibc(1.0 - np_exp_glibc(out[0]) - np_exp_glibc(out[1]));
    out[3] = np_log_glibc(0.01);
}
// The device computes the aligner's constants, set4's log(var) and calculate_transitions with restatements of glibc 2.35's
// log / exp / logf (np_log.h, np_logf.h; the x86-64 FMA variants this image's libm selects).  A host whose libm is another
// version or lacks FMA computes what the REFERENCE would compute there -- which may differ in the last bit from the
// restatement.  This check compares them on `n` pseudo-random arguments in the ranges the path uses; 0 mismatches means
// device-side constants are bit-identical to this host's libm.
int np_selftest_libm(uint64_t n, uint64_t seed, uint64_t* n_mismatch)
{
    if (!n_mismatch) return NP_ERR_INVALID;
    uint64_t bad = 0, s = seed ? seed : 1;
    for (uint64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);            // [0, 1)
        const double x_log = ldexp(1.0 + u, (int)((s >> 3) % 24) - 12);              // log: 2^-12 .. 2^12 (probabilities, variances, ratios)
        const double x_exp = 30.0 * u;                                               // exp(-x): 0 .. 30 nats
        const float xf = (float)ldexp(1.0 + u, (int)((s >> 5) % 16) - 10);           // logf: transition probabilities
        bad += np_log_glibc(x_log) != log(x_log);
        bad += np_exp_glibc(-x_exp) != exp(-x_exp);
        bad += np_logf_glibc(xf) != logf(xf);
    }
    *n_mismatch = bad;
    return NP_OK;
}

void np_restated_log_exp(const double* x, size_t n, double* out_log, double* out_exp)
{
    for (size_t i = 0; i < n; ++i) { out_log[i] = np_log_glibc(x[i]); out_exp[i] = np_exp_glibc(-x[i]); }
}

} // extern "C"
