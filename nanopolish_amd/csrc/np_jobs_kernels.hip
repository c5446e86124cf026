// np_jobs_kernels.hip -- call-methylation work-item generation on the device for reads that
// are identity-aligned to their reference strand (the bench/test layout; a CIGAR-driven caller supplies kpos itself):
//   motif scan + grouping            calculate_methylation_for_read, src/basemods/nanopolish_basemods.cpp:298-320
//   window rule / boundary rules     :328-345, EventAlignmentRecord bounds src/alignment/nanopolish_alignment_db.cpp:65-71,697-708
//   methylated / unmethylated k-mers Alphabet::methylate / reverse_complement (src/common/nanopolish_alphabet.h:59-253) and
//                                    HMMInputSequence::get_kmer_rank (src/hmm/nanopolish_hmm_input_sequence.h:76-91)
// for the methylation alphabets whose recognition site is a dinucleotide (cpg: CG -> MG, gpc: GC -> GM).  It is the device
// twin of np_cm_build_jobs_identity (np_host.cpp), against which tests/test_gpu_jobs.py compares it item by item.
#include "np_kernels.h"

namespace {

struct site2 { char a, b, ma, mb, ca, cb; };          // site, methylated site, methylated complement (as written forward)

__device__ __forceinline__ site2 site_of(int alphabet)
{
    // nanopolish_alphabet.cpp:67-125: cpg {"CG","MG","GM"}, gpc {"GC","GM","MG"}
    return alphabet == 2 ? site2{'G', 'C', 'G', 'M', 'M', 'G'} : site2{'C', 'G', 'M', 'G', 'G', 'M'};
}
__device__ __forceinline__ int digit(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'M' ? 3 : c == 'T' ? 4 : 0; }   // "ACGMT"
__device__ __forceinline__ char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'M' ? 'G' : c == 'T' ? 'A' : 'T'; }   // "TGCGA"

// character q of the window [w0, w0 + len) of ref, after Alphabet::methylate of the WINDOW (a site cut by the window's
// end stays unmethylated)
__device__ __forceinline__ char meth_char(const char* __restrict__ ref, int w0, int len, int q, const site2& s)
{
    const char c = ref[w0 + q];
    if (c == s.a && q + 1 < len && ref[w0 + q + 1] == s.b) return s.ma;
    if (q > 0 && c == s.b && ref[w0 + q - 1] == s.a) return s.mb;
    return c;
}
// character j of reverse_complement(methylate(window)): a methylated site is replaced by its methylated complement
__device__ __forceinline__ char rc_meth_char(const char* __restrict__ ref, int w0, int len, int j, const site2& s)
{
    const int q = len - 1 - j;
    const char c = ref[w0 + q];
    if (c == s.a && q + 1 < len && ref[w0 + q + 1] == s.b) return s.ca;       // first base of a methylated site
    if (q > 0 && c == s.b && ref[w0 + q - 1] == s.a) return s.cb;            // second base
    return comp(c);
}

// pass 1, one lane per read: sequential motif scan and grouping, the skip rules, slot and k-mer offsets
__global__ void __launch_bounds__(64) np_cm_groups_kernel(int n_reads, const char* __restrict__ seq, const int64_t* __restrict__ seq_off,
                                                           int alphabet, int k, int min_separation, int min_flank,
                                                           const int64_t* __restrict__ group_off, const int64_t* __restrict__ rank_off_cap,
                                                           int32_t* __restrict__ first_site, int32_t* __restrict__ last_site,
                                                           int32_t* __restrict__ n_motif, int64_t* __restrict__ group_rank_off,
                                                           int32_t* __restrict__ n_groups)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_reads) return;
    const char* ref = seq + seq_off[r];
    const int n = (int)(seq_off[r + 1] - seq_off[r]);
    const site2 s = site_of(alphabet);
    const int64_t g0 = group_off[r];
    const int cap = (int)(group_off[r + 1] - g0);
    const int64_t rank_cap = rank_off_cap[r + 1] - rank_off_cap[r];
    int ng = 0, first = -1, last = -1, cnt = 0;
    int64_t w = 0;
    bool overflow = false;
    auto close_group = [&]() {
        if (cnt == 0) return;
        const int sub_start = first - min_flank, sub_end = last + min_flank, span = last - first;
        const bool skip = sub_start <= min_separation || span > 200 ||                       // basemods.cpp:334
                          sub_start < k || sub_end + k >= n;                                 // alignment_db.cpp:65-71,697-708
        if (!skip) {
            const int nk = sub_end - sub_start + 1 - k + 1;
            if (ng >= cap || w + 2 * (int64_t)nk > rank_cap) { overflow = true; }
            else {
                first_site[g0 + ng] = first; last_site[g0 + ng] = last; n_motif[g0 + ng] = cnt;
                group_rank_off[g0 + ng] = rank_off_cap[r] + w;
                w += 2 * (int64_t)nk;
                ng++;
            }
        }
        cnt = 0;
    };
    for (int i = 0; i + 1 < n; ++i) {
        if (ref[i] == s.a && ref[i + 1] == s.b) {                                            // is_motif_match, whole site
            if (cnt > 0 && i - last > min_separation) close_group();
            if (cnt == 0) first = i;
            last = i; cnt++;
        }
    }
    close_group();
    n_groups[r] = overflow ? -1 : ng;
}

// pass 2, one block per read: every group's two work items and their k-mer ranks
__global__ void __launch_bounds__(256) np_cm_items_kernel(int n_reads, const char* __restrict__ seq, const int64_t* __restrict__ seq_off,
                                                          const uint8_t* __restrict__ read_rc, int alphabet, int k, int min_flank,
                                                          const int64_t* __restrict__ group_off, const int32_t* __restrict__ first_site,
                                                          const int32_t* __restrict__ last_site, const int64_t* __restrict__ group_rank_off,
                                                          const int32_t* __restrict__ n_groups, np_hmm_job_dev* __restrict__ jobs,
                                                          int32_t* __restrict__ kpos, uint16_t* __restrict__ job_ranks)
{
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const char* ref = seq + seq_off[r];
    const int n = (int)(seq_off[r + 1] - seq_off[r]);
    const bool rc = read_rc[r] != 0;
    const site2 s = site_of(alphabet);
    const int64_t g0 = group_off[r];
    const int cap = (int)(group_off[r + 1] - g0);
    const int ng = n_groups[r] > 0 ? n_groups[r] : 0;
    // unused slots: items the scoring kernel drops (score NaN)
    for (int g = ng + threadIdx.x; g < cap; g += 256) {
        for (int v = 0; v < 2; ++v) {
            np_hmm_job_dev jb; jb.rank_off = 0; jb.n_kmers = 0; jb.read = (uint32_t)r; jb.e_start = jb.e_stop = 0; jb.stride = 1; jb.flags = NP_JOB_SKIP;
            jobs[2 * (g0 + g) + v] = jb;
            kpos[2 * (2 * (g0 + g) + v)] = 0; kpos[2 * (2 * (g0 + g) + v) + 1] = 0;
        }
    }
    for (int g = 0; g < ng; ++g) {
        const int sub_start = first_site[g0 + g] - min_flank, sub_end = last_site[g0 + g] + min_flank;
        const int len = sub_end - sub_start + 1, nk = len - k + 1;
        const int64_t ro = group_rank_off[g0 + g];
        if (threadIdx.x < 2) {
            const int v = threadIdx.x;                                 // 0: unmethylated, 1: methylated
            np_hmm_job_dev jb;
            jb.rank_off = ro + (int64_t)v * nk; jb.n_kmers = (uint32_t)nk; jb.read = (uint32_t)r; jb.e_start = jb.e_stop = 0; jb.stride = 1;
            jb.flags = NP_HAF_ALLOW_PRE_CLIP | NP_HAF_ALLOW_POST_CLIP;                        // basemods.cpp:363
            jobs[2 * (g0 + g) + v] = jb;
            // read-strand k-mer positions of the window ends (flip_k_strand for reverse-strand reads, squiggle_read.h:229-233)
            kpos[2 * (2 * (g0 + g) + v)] = rc ? n - sub_start - k : sub_start;
            kpos[2 * (2 * (g0 + g) + v) + 1] = rc ? n - sub_end - k : sub_end;
        }
        for (int i = threadIdx.x; i < nk; i += 256) {
            // HMMInputSequence::get_kmer_rank(i, k, do_rc): the forward k-mer at i, or the reverse-complement string's k-mer at
            // len - i - k
            uint32_t ru = 0, rm = 0;
            for (int t = 0; t < k; ++t) {
                char cu, cm;
                if (!rc) { cu = ref[sub_start + i + t]; cm = meth_char(ref, sub_start, len, i + t, s); }
                else { const int j = len - i - k + t; cu = comp(ref[sub_start + len - 1 - j]); cm = rc_meth_char(ref, sub_start, len, j, s); }
                ru = ru * 5u + (uint32_t)digit(cu);
                rm = rm * 5u + (uint32_t)digit(cm);
            }
            job_ranks[ro + i] = (uint16_t)ru;
            job_ranks[ro + nk + i] = (uint16_t)rm;
        }
    }
}

} // namespace

hipError_t np_launch_cm_build_jobs(int n_reads, const char* seq, const int64_t* seq_off, const uint8_t* read_rc, int alphabet, int k,
                                   int min_separation, int min_flank, const int64_t* group_off, const int64_t* rank_off_cap,
                                   np_hmm_job_dev* jobs, int32_t* kpos, uint16_t* job_ranks, int32_t* first_site, int32_t* last_site,
                                   int32_t* n_motif, int64_t* group_rank_off, int32_t* n_groups, hipStream_t s)
{
    if (n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(np_cm_groups_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, s, n_reads, seq, seq_off, alphabet, k, min_separation,
                       min_flank, group_off, rank_off_cap, first_site, last_site, n_motif, group_rank_off, n_groups);
    hipLaunchKernelGGL(np_cm_items_kernel, dim3(n_reads), dim3(256), 0, s, n_reads, seq, seq_off, read_rc, alphabet, k, min_flank, group_off,
                       first_site, last_site, group_rank_off, n_groups, jobs, kpos, job_ranks);
    return hipGetLastError();
}
