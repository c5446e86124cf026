// np_jobs_kernels.hip -- call-methylation work-item generation on the device, for reads that
// are identity-aligned to their reference strand (the bench layout) and for reads aligned by a BAM record's CIGAR:
//   CIGAR walk                       get_aligned_segments, src/alignment/nanopolish_anchor.cpp:20-95
//   aligned-event filter, bounds     EventAlignmentRecord, AlignmentDB::_find_iter_by_ref_bounds,
//                                    src/alignment/nanopolish_alignment_db.cpp:63-72,688-711
//   motif scan + grouping            calculate_methylation_for_read, src/basemods/nanopolish_basemods.cpp:298-320
//   window rule / boundary rules     :328-345, EventAlignmentRecord bounds src/alignment/nanopolish_alignment_db.cpp:65-71,697-708
//   methylated / unmethylated k-mers Alphabet::methylate / reverse_complement (src/common/nanopolish_alphabet.h:59-253) and
//                                    HMMInputSequence::get_kmer_rank (src/hmm/nanopolish_hmm_input_sequence.h:76-91)
// for the four methylation alphabets of the r9.4_450bps kit: cpg (CG -> MG), gpc (GC -> GM), dam (GATC -> GMTC) and dcm
// (CCAGG / CCTGG -> CMAGG / CMTGG).  It is the device
// twin of np_cm_build_jobs_identity / np_cm_build_jobs_cigar (np_host.cpp), against which tests/test_gpu_jobs.py compares it
// item by item.
// CIGAR mode never materialises the aligned pairs: a per-read exclusive scan of the operations' reference / read advances
// (np_cigar_index_kernel) turns every lower_bound of the reference into one binary search over the operations.
#include "np_kernels.h"
#include "np_cigar.h"
#include "np_motif.h"

namespace {

struct site2 { char a, b, ma, mb, ca, cb; };          // site, methylated site, methylated complement (as written forward)

__device__ __forceinline__ site2 site_of(int alphabet)
{
    // nanopolish_alphabet.cpp:67-125: cpg {"CG","MG","GM"}, gpc {"GC","GM","MG"}
    return alphabet == 2 ? site2{'G', 'C', 'G', 'M', 'M', 'G'} : site2{'C', 'G', 'M', 'G', 'G', 'M'};
}

__device__ __forceinline__ int digit(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'M' ? 3 : c == 'T' ? 4 : 0; }   // "ACGMT"
__device__ __forceinline__ char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'M' ? 'G' : c == 'T' ? 'A' : 'T'; }   // "TGCGA"

// character q of methylate(window), and character of reverse_complement(methylate(window)) that comes from window position q
__device__ __forceinline__ void meth_chars_general(const char* __restrict__ ref, int w0, int len, int q, const sites_t& S, char& cm, char& crc)
{
    const char c = ref[w0 + q];
    cm = c; crc = comp(c);
    for (int d = 0; d < S.len; ++d) {
        const int i = site_at(ref, w0, len, q - d, S);
        if (i >= 0) { cm = S.m[i][d]; crc = S.mc[i][d]; return; }
    }
}

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


// per read: (first filtered read_pos, last filtered read_pos) of the aligned events, or (-1, -1); [2] = 1 if the CIGAR is usable
struct cig_read_t { int32_t first_q, last_q, ok, pad; };

// EventAlignmentRecord keeps the aligned pairs with k <= read_pos and read_pos + k < read_len (alignment_db.cpp:63-72);
// _find_iter_by_ref_bounds (:688-711) takes lower_bound(ref_start) / lower_bound(ref_stop) on them.  Both coordinates grow
// along the pairs, so the first kept pair with ref_pos >= x is the later of "first pair with ref_pos >= x" and "first kept
// pair".  Returns false when unbounded, else the reference-strand read positions of the bounding pairs.
__device__ __forceinline__ bool cigar_find_bounds(const cig_view& c, const cig_read_t& R, int ref_start, int ref_stop, int& q1, int& q2)
{
    if (R.first_q < 0) return false;
    int qa, ra, qb, rb;
    if (!first_aligned_ref_ge(c, ref_start, qa, ra) || !first_aligned_ref_ge(c, ref_stop, qb, rb)) return false;
    bool is_first = false;
    if (qa <= R.first_q) { is_first = true; if (qa < R.first_q) { int t; first_aligned_read_ge(c, R.first_q, t, ra); qa = R.first_q; } }
    if (qb < R.first_q) qb = R.first_q;
    if (qa > R.last_q || qb > R.last_q) return false;                                   // lower_bound == end()
    if (is_first && ra > ref_start) return false;                                      // not left-bounded (:701-702)
    q1 = qa; q2 = qb;
    return true;
}

// pass 0 (CIGAR mode), one wave per read: exclusive scan of the operations' advances, then the kept range of pairs
__global__ void __launch_bounds__(64) np_cigar_index_kernel(int n_reads, const uint32_t* __restrict__ cigar, const int64_t* __restrict__ cigar_off,
                                                            const int32_t* __restrict__ read_len, int k, int32_t* __restrict__ op_ref,
                                                            int32_t* __restrict__ op_read, cig_read_t* __restrict__ out)
{
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const int lane = threadIdx.x;
    const uint32_t* cg = cigar + cigar_off[r];
    const int n = (int)(cigar_off[r + 1] - cigar_off[r]);
    int32_t* oref = op_ref + cigar_off[r] + r;            // n + 1 entries per read
    int32_t* oread = op_read + cigar_off[r] + r;
    int base_ref = 0, base_read = 0;
    bool bad = false;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        int dr = 0, dq = 0;
        if (i < n) {
            const uint32_t w = cg[i];
            const int len = (int)(w >> 4);
            const uint32_t op = w & 0xf;
            if (op == 3 || op == 6 || op > 8) bad = true;                                // spliced / pad / unknown: rejected by the reference
            dr = (op == 0 || op == 7 || op == 8 || op == 2) ? len : 0;
            dq = (op == 0 || op == 7 || op == 8 || op == 1 || op == 4) ? len : 0;
        }
        int sr = dr, sq = dq;                                                            // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int tr = __shfl_up(sr, o, 64), tq = __shfl_up(sq, o, 64);
            if (lane >= o) { sr += tr; sq += tq; }
        }
        if (i < n) { oref[i] = base_ref + sr - dr; oread[i] = base_read + sq - dq; }
        base_ref += __shfl(sr, 63, 64); base_read += __shfl(sq, 63, 64);
    }
    if (lane == 0) { oref[n] = base_ref; oread[n] = base_read; }
    bad = __any(bad);
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        cig_read_t R; R.first_q = -1; R.last_q = -1; R.ok = bad ? 0 : 1; R.pad = 0;
        if (!bad) {
            const cig_view c{cg, oref, oread, n};
            int q = 0, rr = 0, ql = 0;
            const int hi_q = read_len[r] - k - 1;
            if (first_aligned_read_ge(c, k, q, rr) && q <= hi_q && last_aligned_read_le(c, hi_q, ql) && ql >= q) { R.first_q = q; R.last_q = ql; }
        }
        out[r] = R;
    }
}

// pass 1, one wavefront per read: the motif test runs on 64 positions at a time (coalesced byte loads, one ballot), the
// sequential grouping only visits the matches (wave-uniform state, every lane computes the same values; lane 0 stores): the
// skip rules, slot and k-mer offsets.  (One LANE per read, a byte at a time, took 5.8 ms per 100 000 reads -- as long as the
// slowest lane's 5000 dependent iterations whatever the batch size.)
__global__ void __launch_bounds__(64) np_cm_groups_kernel(int n_reads, const char* __restrict__ seq, const int64_t* __restrict__ seq_off,
                                                           const int32_t* __restrict__ seq_len, const uint8_t* __restrict__ read_rc,
                                                           const uint32_t* __restrict__ cigar, const int64_t* __restrict__ cigar_off,
                                                           const int32_t* __restrict__ read_len, const int32_t* __restrict__ op_ref,
                                                           const int32_t* __restrict__ op_read, const cig_read_t* __restrict__ cig_reads,
                                                           int32_t* __restrict__ group_kpos, int32_t* __restrict__ deg_kpos,
                                                           int alphabet, int k, int min_separation, int min_flank,
                                                           const int64_t* __restrict__ group_off, const int64_t* __restrict__ rank_off_cap,
                                                           int32_t* __restrict__ first_site, int32_t* __restrict__ last_site,
                                                           int32_t* __restrict__ n_motif, int64_t* __restrict__ group_rank_off,
                                                           int32_t* __restrict__ n_groups)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= n_reads) return;
    const bool writer = lane == 0;
    const char* ref = seq + seq_off[r];
    const int n = seq_len ? seq_len[r] : (int)(seq_off[r + 1] - seq_off[r]);
    const site2 s = site_of(alphabet);
    const int64_t g0 = group_off[r];
    const int cap = (int)(group_off[r + 1] - g0);
    const int64_t rank_cap = rank_off_cap[r + 1] - rank_off_cap[r];
    // CIGAR mode: the read's operations, their scanned offsets and the kept range of aligned pairs
    const bool by_cigar = cigar != nullptr;
    cig_view cv{nullptr, nullptr, nullptr, 0};
    cig_read_t cr{-1, -1, 0, 0};
    bool rc = false;
    int rl = 0;
    if (by_cigar) {
        cv.cigar = cigar + cigar_off[r]; cv.n = (int)(cigar_off[r + 1] - cigar_off[r]);
        cv.op_ref = op_ref + cigar_off[r] + r; cv.op_read = op_read + cigar_off[r] + r;
        cr = cig_reads[r]; rc = read_rc[r] != 0; rl = read_len[r];
        // flip_k_strand for reverse-strand reads (squiggle_read.h:229-233); the degenerate-record test needs these two
        if (writer) {
            deg_kpos[2 * r] = cr.first_q < 0 ? -1 : (rc ? rl - cr.first_q - k : cr.first_q);
            deg_kpos[2 * r + 1] = cr.first_q < 0 ? -1 : (rc ? rl - cr.last_q - k : cr.last_q);
        }
    }
    // Round 5: the grouping itself runs on the lanes.  (Rounds 2-4 walked the matches one by one with wave-uniform state: ~45 scalar
    // instructions per match, 16 000 per read, and a CU issues one scalar instruction per cycle for all its waves -- 2.8 ms per 100 000
    // reads, bound by exactly that, profiles/r05_pmc.json.)  Per chunk of 64 positions: a match STARTS a group when the previous match
    // (in the chunk, or the open group's last) is more than min_separation behind it; the matches before the chunk's first start extend
    // the group left open by the previous chunks; every start but the last closes its group inside the chunk (its matches: up to the
    // next start), the last one's stays open.
    // Closed groups are QUEUED (position order: the carried group, then the chunk's own by lane) and taken 64 at a time, one per lane:
    // the window rules, then the slot index / k-mer offset of a kept group as the running count / sum plus a prefix over the kept groups
    // before it (as the reference emits them, basemods.cpp:306-336).  Applying the rules chunk by chunk -- two or three groups, i.e.
    // lanes, at a time, 85 times per read -- was 85 rounds of the CIGAR searches' dependent loads with nothing else to run: 2.25 ms per
    // 100 000 reads, all of it that latency.
    constexpr int QCAP = 256;                                    // (a chunk closes at most 33 groups; the queue is emptied from 64 on)
    __shared__ int q_first[QCAP], q_last[QCAP], q_cnt[QCAP];
    int q_head = 0, nq = 0;                                      // wave-uniform
    int ng = 0;
    int64_t w = 0;
    bool overflow = false;
    int o_first = -1, o_last = -1, o_cnt = 0;                    // the open group (wave-uniform)
    const unsigned long long below = (1ull << lane) - 1ull;
    // the first m <= 64 queued groups, one per lane
    auto emit = [&](const int m) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const bool mine = lane < m;
        const int slot = (q_head + lane) & (QCAP - 1);
        const int first = mine ? q_first[slot] : 0, last = mine ? q_last[slot] : 0, cnt = mine ? q_cnt[slot] : 0;
        q_head = (q_head + m) & (QCAP - 1); nq -= m;
        const int sub_start = first - min_flank, sub_end = last + min_flank, span = last - first;
        bool skip = !mine || sub_start <= min_separation || span > 200;                      // basemods.cpp:334
        int q1 = 0, q2 = 0;
        if (!skip) {
            if (by_cigar) skip = !cigar_find_bounds(cv, cr, sub_start, sub_end, q1, q2) || sub_end >= n;
            else skip = sub_start < k || sub_end + k >= n;                                   // alignment_db.cpp:65-71,697-708
        }
        const bool keep = !skip;
        const int nk2 = keep ? 2 * (sub_end - sub_start + 1 - k + 1) : 0;
        const unsigned long long km = __ballot(keep);
        if (km == 0ull) return;
        int scan = nk2;                                                                      // inclusive scan over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(scan, o, 64); if (lane >= o) scan += t; }
        const int idx = ng + __popcll(km & below);
        const int64_t off = w + (scan - nk2);
        const bool fits = idx < cap && off + nk2 <= rank_cap;
        if (keep && fits) {
            first_site[g0 + idx] = first; last_site[g0 + idx] = last; n_motif[g0 + idx] = cnt;
            group_rank_off[g0 + idx] = rank_off_cap[r] + off;
            if (by_cigar) {
                group_kpos[2 * (g0 + idx)] = rc ? rl - q1 - k : q1;
                group_kpos[2 * (g0 + idx) + 1] = rc ? rl - q2 - k : q2;
            }
        }
        if (__ballot(keep && !fits) != 0ull) overflow = true;                               // the read comes back n_groups = -1: its slots are void
        ng += __popcll(km);
        w += (int64_t)__shfl(scan, 63, 64);
    };
    const sites_t S = sites_of(alphabet);
    // (two-base motifs: the next chunk's bytes are requested before this chunk is grouped)
    char nb0 = 0, nb1 = 0;
    if (S.len == 2) { nb0 = lane + 1 < n ? ref[lane] : 0; nb1 = lane + 1 < n ? ref[lane + 1] : 0; }
    for (int base = 0; base + 1 < n; base += 64) {
        const int pos = base + lane;
        bool hit;
        if (S.len == 2) {
            const char b0 = nb0, b1 = nb1;
            const int np_ = pos + 64;
            nb0 = np_ + 1 < n ? ref[np_] : 0; nb1 = np_ + 1 < n ? ref[np_ + 1] : 0;
            hit = pos + 1 < n && b0 == s.a && b1 == s.b;                                      // is_motif_match, whole site
        } else hit = pos + 1 < n && site_at(ref, 0, n, pos, S) >= 0;
        const unsigned long long bits = __ballot(hit);
        if (bits == 0ull) continue;
        const unsigned long long pb = bits & below;
        const bool has_prev = pb != 0ull || o_cnt > 0;
        const int prev = pb ? base + 63 - __clzll((long long)pb) : o_last;
        const bool start = hit && (!has_prev || pos - prev > min_separation);
        const unsigned long long sm = __ballot(start);
        const int fs = sm ? __builtin_ctzll(sm) : 64;
        const unsigned long long ext = fs == 64 ? bits : bits & ((1ull << fs) - 1ull);       // matches that still belong to the open group
        if (ext) { o_last = base + 63 - __clzll((long long)ext); o_cnt += __popcll(ext); }
        if (sm == 0ull) continue;
        const int ls = 63 - __clzll((long long)sm);                                           // the last start: its group stays open
        // this lane's closed group: a start other than the last -> its own segment; the last start's lane -> the group carried in (if any)
        const unsigned long long later = sm & ~((below << 1) | 1ull);                        // starts above this lane
        const int ns = later ? __builtin_ctzll(later) : 64;
        const unsigned long long seg = ns == 64 ? bits & ~below : bits & ~below & ((1ull << ns) - 1ull);
        const bool own = start && lane != ls;
        const bool carried = lane == ls && o_cnt > 0;
        const int n_carried = o_cnt > 0 ? 1 : 0;
        const unsigned long long om = sm & ~(1ull << ls);                                     // the lanes with a closed group of their own
        if (own || carried) {
            const int slot = (q_head + nq + (carried ? 0 : n_carried + __popcll(om & below))) & (QCAP - 1);
            q_first[slot] = own ? pos : o_first;
            q_last[slot] = own ? base + 63 - __clzll((long long)(seg | 1ull)) : o_last;
            q_cnt[slot] = own ? __popcll(seg) : o_cnt;
        }
        nq += n_carried + __popcll(om);
        if (nq >= 64) emit(64);
        // the last start opens the new group
        const unsigned long long tail = bits & ~((1ull << ls) - 1ull);
        o_first = base + ls; o_last = base + 63 - __clzll((long long)tail); o_cnt = __popcll(tail);
    }
    if (o_cnt > 0) {
        if (lane == 0) { const int slot = (q_head + nq) & (QCAP - 1); q_first[slot] = o_first; q_last[slot] = o_last; q_cnt[slot] = o_cnt; }
        nq += 1;
    }
    while (nq > 0) emit(nq < 64 ? nq : 64);
    if (writer) n_groups[r] = overflow ? -1 : ng;
}

// pass 2, one block per read: every group's two work items and their k-mer ranks.
// The k-mers of all groups of the read form one flat index space (group_rank_off is its prefix sum, two rank arrays per
// group), so every thread of the block has a k-mer to rank in every iteration; the group of a flat index is found by a
// binary search over the read's (L1-resident) offsets.  A k-mer and its methylated twin need the 8 bases around it once.
__global__ void __launch_bounds__(256) np_cm_items_kernel(int n_reads, const char* __restrict__ seq, const int64_t* __restrict__ seq_off,
                                                          const int32_t* __restrict__ seq_len, const int32_t* __restrict__ group_kpos,
                                                          const uint8_t* __restrict__ read_rc, int alphabet, int k, int min_flank,
                                                          const int64_t* __restrict__ group_off, const int32_t* __restrict__ first_site,
                                                          const int32_t* __restrict__ last_site, const int64_t* __restrict__ group_rank_off,
                                                          const int32_t* __restrict__ n_groups, np_hmm_job_dev* __restrict__ jobs,
                                                          int32_t* __restrict__ kpos, uint16_t* __restrict__ job_ranks, int write_unused)
{
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const char* ref = seq + seq_off[r];
    const int n = seq_len ? seq_len[r] : (int)(seq_off[r + 1] - seq_off[r]);
    const bool rc = read_rc[r] != 0;
    const site2 s = site_of(alphabet);
    const int64_t g0 = group_off[r];
    const int cap = (int)(group_off[r + 1] - g0);
    const int ng = n_groups[r] > 0 ? n_groups[r] : 0;
    // the work-item records: two per group slot; unused slots are items the scoring kernel drops (score NaN) -- or, when the caller has
    // declared the slot layout (np_set_job_layout: every consumer then visits live items only), are not touched at all
    for (int g = threadIdx.x; g < (write_unused ? cap : ng); g += 256) {
        np_hmm_job_dev jb; jb.rank_off = 0; jb.n_kmers = 0; jb.read = (uint32_t)r; jb.e_start = jb.e_stop = 0; jb.stride = 1; jb.flags = NP_JOB_SKIP;
        int k0 = 0, k1 = 0, nk = 0;
        if (g < ng) {
            const int sub_start = first_site[g0 + g] - min_flank, sub_end = last_site[g0 + g] + min_flank;
            nk = sub_end - sub_start + 1 - k + 1;
            jb.rank_off = group_rank_off[g0 + g]; jb.n_kmers = (uint32_t)nk;
            jb.flags = NP_HAF_ALLOW_PRE_CLIP | NP_HAF_ALLOW_POST_CLIP;                        // basemods.cpp:363
            // read-strand k-mer positions of the window ends (flip_k_strand for reverse-strand reads, squiggle_read.h:229-233)
            k0 = group_kpos ? group_kpos[2 * (g0 + g)] : (rc ? n - sub_start - k : sub_start);
            k1 = group_kpos ? group_kpos[2 * (g0 + g) + 1] : (rc ? n - sub_end - k : sub_end);
        }
        jobs[2 * (g0 + g)] = jb;                                       // unmethylated
        jb.rank_off += nk;
        jobs[2 * (g0 + g) + 1] = jb;                                   // methylated
        int4 kp; kp.x = k0; kp.y = k1; kp.z = k0; kp.w = k1;
        *(int4*)(kpos + 4 * (g0 + g)) = kp;
    }
    if (ng == 0) return;
    // Groups in tiles of NP_ITEM_TILE: the tile's offsets and windows go to LDS, so the binary search and the three per-group
    // values of every k-mer are LDS reads instead of dependent global loads (the kernel was bound by those round trips).
#define NP_ITEM_TILE 512
    __shared__ int t_first[NP_ITEM_TILE + 1], t_sub[NP_ITEM_TILE], t_len[NP_ITEM_TILE];
    // digit(c) and digit(comp(c)) of every byte value: one LDS read instead of two chains of ten compares and selects per base
    // (the kernel was bound by those: ~250 vector instructions per k-mer)
    __shared__ uint8_t t_dig[256], t_dcomp[256];
    t_dig[threadIdx.x] = (uint8_t)digit((char)threadIdx.x); t_dcomp[threadIdx.x] = (uint8_t)digit(comp((char)threadIdx.x));
    const uint32_t dma = (uint32_t)digit(s.ma), dmb = (uint32_t)digit(s.mb), dca = (uint32_t)digit(s.ca), dcb = (uint32_t)digit(s.cb);
    for (int tile = 0; tile < ng; tile += NP_ITEM_TILE) {
    const int tn = ng - tile < NP_ITEM_TILE ? ng - tile : NP_ITEM_TILE;
    const int64_t base = group_rank_off[g0 + tile];
    __syncthreads();
    for (int g = threadIdx.x; g < tn; g += 256) {
        const int sub_start = first_site[g0 + tile + g] - min_flank, sub_end = last_site[g0 + tile + g] + min_flank;
        t_first[g] = (int)((group_rank_off[g0 + tile + g] - base) >> 1);   // flat index of the group's first k-mer (two rank arrays per group)
        t_sub[g] = sub_start; t_len[g] = sub_end - sub_start + 1;
    }
    __syncthreads();
    const int total = t_first[tn - 1] + (t_len[tn - 1] - k + 1);
    for (int t = threadIdx.x; t < total; t += 256) {
        int lo = 0, hi = tn - 1;                                       // the last group whose first k-mer is <= t
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (t_first[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const int g = lo;
        const int i = t - t_first[g];
        const int64_t ro = base + 2 * (int64_t)t_first[g];
        const int sub_start = t_sub[g];
        const int len = t_len[g], nk = len - k + 1;
        // HMMInputSequence::get_kmer_rank(i, k, do_rc): the forward k-mer at i, or the reverse-complement string's k-mer at
        // len - i - k, i.e. window characters i+k-1 down to i complemented.  Either way the characters are window positions
        // i .. i+k-1, and methylation looks one position to each side INSIDE the window (Alphabet::methylate of the window).
        uint32_t ru = 0, rm = 0;
        if (alphabet > 2) {
            // dam / dcm: sites of 4 and 5 bases (general form; rare alphabets, no fast path)
            const sites_t S = sites_of(alphabet);
            uint32_t pw = 1;
            for (int q = i; q < i + k; ++q) {
                char cm, crc;
                meth_chars_general(ref, sub_start, len, q, S, cm, crc);
                const char c = ref[sub_start + q];
                if (!rc) { ru = ru * 5u + (uint32_t)digit(c); rm = rm * 5u + (uint32_t)digit(cm); }
                else { ru += pw * (uint32_t)digit(comp(c)); rm += pw * (uint32_t)digit(crc); pw *= 5u; }
            }
        } else {
            // the k-mer's digits and its methylated twin's: a base is replaced when it is the first base of a whole site (next base
            // is the site's second) or the second base of one (previous base is its first), inside the window.
            // (round 5: the k = 6 case is unrolled at compile time and the eight bytes it looks at -- the base before the k-mer, the
            //  k-mer, the base after -- come from three aligned dword loads and two byte alignments instead of eight byte loads: the kernel
            //  spent 67 % of its wave-cycles waiting on them, profiles/r05_pmc.json.  The dwords reach up to 3 bytes before and 10 after
            //  the k-mer's first base: a k-mer closer than that to the end of the read's sequence takes the byte loads.)
            auto kmer = [&](auto KC) {
                constexpr int KK = decltype(KC)::value;                // 0: k at run time
                const int kk = KK ? KK : k;
                char w_[KK ? KK + 1 : 1];
                char prev, cur;
                const char* p0 = ref + sub_start + i - 1;               // (sub_start > min_separation >= 0: never before the sequence)
                if (KK == 6 && sub_start + i + 10 < n) {
                    const uintptr_t a = (uintptr_t)p0;
                    const uint32_t* __restrict__ aw = (const uint32_t*)(a & ~(uintptr_t)3);
                    const uint32_t d0 = aw[0], d1 = aw[1], d2 = aw[2], sh = (uint32_t)(a & 3u);
                    const uint32_t lo_ = __builtin_amdgcn_alignbyte(d1, d0, sh), hi_ = __builtin_amdgcn_alignbyte(d2, d1, sh);
                    prev = i > 0 ? (char)(lo_ & 0xffu) : (char)0;
                    cur = (char)((lo_ >> 8) & 0xffu);
                    w_[1 % (KK + 1)] = (char)((lo_ >> 16) & 0xffu); w_[2 % (KK + 1)] = (char)(lo_ >> 24);
                    w_[3 % (KK + 1)] = (char)(hi_ & 0xffu); w_[4 % (KK + 1)] = (char)((hi_ >> 8) & 0xffu); w_[5 % (KK + 1)] = (char)((hi_ >> 16) & 0xffu);
                    w_[6 % (KK + 1)] = i + 6 < len ? (char)(hi_ >> 24) : (char)0;
                } else {
                    prev = i > 0 ? p0[0] : (char)0;
                    cur = p0[1];
                    if (KK) {
#pragma unroll
                        for (int q = 0; q < KK; ++q) w_[q + 1] = i + q + 1 < len ? ref[sub_start + i + q + 1] : 0;
                    }
                }
                bool pa = prev == s.a, ca_ = cur == s.a, cb_ = cur == s.b;
                char c0 = cur;
                uint32_t pw = 1;                                       // reverse strand: window position q contributes digit * 5^(q - i)
#pragma unroll
                for (int q = 0; q < kk; ++q) {
                    const char nxt = KK ? w_[q + 1] : (i + q + 1 < len ? ref[sub_start + i + q + 1] : 0);
                    const bool na = nxt == s.a, nb = nxt == s.b;
                    const bool site1 = ca_ && nb, site2 = cb_ && pa;
                    if (!rc) {
                        const uint32_t d = t_dig[(uint8_t)c0];
                        ru = ru * 5u + d;
                        rm = rm * 5u + (site1 ? dma : (site2 ? dmb : d));
                    } else {
                        const uint32_t d = t_dcomp[(uint8_t)c0];
                        ru += pw * d;
                        rm += pw * (site1 ? dca : (site2 ? dcb : d));
                        pw *= 5u;
                    }
                    pa = ca_; ca_ = na; cb_ = nb; c0 = nxt;
                }
            };
            if (k == 6) kmer(std::integral_constant<int, 6>{}); else kmer(std::integral_constant<int, 0>{});
        }
        job_ranks[ro + i] = (uint16_t)ru;
        job_ranks[ro + nk + i] = (uint16_t)rm;
    }
    }
#undef NP_ITEM_TILE
}

} // namespace

hipError_t np_launch_cm_build_jobs(int n_reads, const char* seq, const int64_t* seq_off, const uint8_t* read_rc, int alphabet, int k,
                                   int min_separation, int min_flank, const int64_t* group_off, const int64_t* rank_off_cap,
                                   np_hmm_job_dev* jobs, int32_t* kpos, uint16_t* job_ranks, int32_t* first_site, int32_t* last_site,
                                   int32_t* n_motif, int64_t* group_rank_off, int32_t* n_groups, int write_unused, hipStream_t s)
{
    if (n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(np_cm_groups_kernel, dim3(n_reads), dim3(64), 0, s, n_reads, seq, seq_off, nullptr, read_rc, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, alphabet, k, min_separation,
                       min_flank, group_off, rank_off_cap, first_site, last_site, n_motif, group_rank_off, n_groups);
    hipLaunchKernelGGL(np_cm_items_kernel, dim3(n_reads), dim3(256), 0, s, n_reads, seq, seq_off, nullptr, nullptr, read_rc, alphabet, k, min_flank,
                       group_off, first_site, last_site, group_rank_off, n_groups, jobs, kpos, job_ranks, write_unused);
    return hipGetLastError();
}

// CIGAR mode.  genome: the contig(s) resident on the device; ref_begin[r] / ref_len[r]: the segment the reference fetches
// for read r (contig[pos .. bam_endpos], clipped).  scratch: op_ref / op_read (cigar_off[n_reads] + n_reads int32 each),
// cig_reads (16 B per read), group_kpos (2 int32 per group slot).
hipError_t np_launch_cm_build_jobs_cigar(int n_reads, const char* genome, const int64_t* ref_begin, const int32_t* ref_len,
                                         const uint32_t* cigar, const int64_t* cigar_off, const int32_t* read_len, const uint8_t* read_rc,
                                         int alphabet, int k, int min_separation, int min_flank, const int64_t* group_off,
                                         const int64_t* rank_off_cap, np_hmm_job_dev* jobs, int32_t* kpos, uint16_t* job_ranks,
                                         int32_t* first_site, int32_t* last_site, int32_t* n_motif, int64_t* group_rank_off,
                                         int32_t* n_groups, int32_t* deg_kpos, int32_t* op_ref, int32_t* op_read, void* cig_reads,
                                         int32_t* group_kpos, int write_unused, hipStream_t s)
{
    if (n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(np_cigar_index_kernel, dim3(n_reads), dim3(64), 0, s, n_reads, cigar, cigar_off, read_len, k, op_ref, op_read,
                       (cig_read_t*)cig_reads);
    hipLaunchKernelGGL(np_cm_groups_kernel, dim3(n_reads), dim3(64), 0, s, n_reads, genome, ref_begin, ref_len, read_rc, cigar,
                       cigar_off, read_len, op_ref, op_read, (const cig_read_t*)cig_reads, group_kpos, deg_kpos, alphabet, k, min_separation,
                       min_flank, group_off, rank_off_cap, first_site, last_site, n_motif, group_rank_off, n_groups);
    hipLaunchKernelGGL(np_cm_items_kernel, dim3(n_reads), dim3(256), 0, s, n_reads, genome, ref_begin, ref_len, group_kpos, read_rc, alphabet, k,
                       min_flank, group_off, first_site, last_site, group_rank_off, n_groups, jobs, kpos, job_ranks, write_unused);
    return hipGetLastError();
}

hipError_t np_launch_cigar_index(int n_reads, const uint32_t* cigar, const int64_t* cigar_off, const int32_t* read_len, int k, int32_t* op_ref,
                                 int32_t* op_read, void* cig_reads, hipStream_t s)
{
    if (n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(np_cigar_index_kernel, dim3(n_reads), dim3(64), 0, s, n_reads, cigar, cigar_off, read_len, k, op_ref, op_read,
                       (cig_read_t*)cig_reads);
    return hipGetLastError();
}
