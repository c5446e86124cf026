// np_glue_kernels.hip -- the read-level glue between the aligner and the HMM, kept on the device so the
// fused call-methylation pass needs no host round trip:
//   * np_build_map_kernel : base_to_event_map[].start + events_per_base (src/nanopolish_squiggle_read.cpp:273-301)
//                           and the per-read HMM transitions (src/hmm/nanopolish_profile_hmm_r9.inl:17-76)
//   * np_resolve_kernel   : get_closest_event_to (src/nanopolish_squiggle_read.cpp:161-186) for both window
//                           bounds of every work item + the skip rules of calculate_methylation_for_read
//                           (src/basemods/nanopolish_basemods.cpp:352-358) and the events-per-base QC
//                           (src/nanopolish_squiggle_read.cpp:332)
//   * np_classify_kernel  : bins HMM work items into the (lanes, blocks-per-lane) size classes of np_hmm_kernels.hip
#include "np_kernels.h"
#include "np_motif.h"
#include "np_logf.h"
#include "np_log.h"

namespace {

// wave_shr:1 / wave_shl:1 of a 32-bit value (lane j <- lane j-1 / j+1; the lane without a source keeps `fill`)
__device__ __forceinline__ int wave_shr1_i(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_shl1_i(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }

// One wave per read, lanes = pairs.  The path kernel A emits is monotone and moves one k-mer and/or one event per step, so
//   * every k-mer between the path's first and last has a FIRST pair and a LAST pair, and those two lanes are the only writers of
//     start[k] / stop[k] -- no atomics, and (round 5) no initialisation pass either: the first-pair lane writes -1 itself when the
//     k-mer has no recording pair; only k-mers outside the path's range (none, for an alignment that passed QC) are filled with -1;
//   * a pair "records" when its event differs from the previous pair's (:283).  Of the pairs of one k-mer only the first can fail to
//     record (it shares its event with the previous k-mer); every later pair of that k-mer advanced the event.  Hence
//       start[k] = event of the k-mer's first pair if that records, else of its second pair if it has one, else -1;
//       stop[k]  = event of the k-mer's last pair, unless that pair is also its first and does not record (then -1).
// A lane's neighbours p[i-1], p[i+1] come from the neighbouring lanes (DPP), the chunk's edges from the previous chunk's last lane
// and one extra load: 4 + 1 loads per 256 pairs (round 4: 16, plus two stores per k-mer of -1).
template <bool STOP>
__global__ void __launch_bounds__(64) np_build_map_kernel(int n_reads, np_read_dev* reads, const int64_t* pair_off,
                                                          const np_pair* pairs, const int32_t* pair_begin,
                                                          const int32_t* n_pairs, int32_t* map_start, int32_t* map_stop,
                                                          double* events_per_base, double indel_bias)
{
    const int ri = blockIdx.x;
    if (ri >= n_reads) return;
    const int lane = threadIdx.x;
    np_read_dev* rd = reads + ri;
    const int K = (int)rd->n_kmers;
    int32_t* ms = map_start + rd->rank_off;
    int32_t* mp = STOP ? map_stop + rd->rank_off : nullptr;
    const int np_ = n_pairs[ri];
    if (np_ <= 0) {
        // failed alignment: events cleared, events_per_base = 0 (squiggle_read.cpp:324-329); IndexPair(): start = stop = -1
        for (int k = lane; k < K; k += 64) { ms[k] = -1; if (STOP) mp[k] = -1; }
        if (lane == 0) { events_per_base[ri] = 0.0; np_transitions(0.0, indel_bias, rd->trans); }
        return;
    }
    const np_pair* p = pairs + pair_off[ri] + pair_begin[ri];
    const np_pair first = p[0], last = p[np_ - 1];
    // k-mers the path does not reach (the aligner's QC demands k-mer 0 ... K-1, so these loops are empty for a read that aligned)
    for (int k = lane; k < first.ref_pos && k < K; k += 64) { ms[k] = -1; if (STOP) mp[k] = -1; }
    for (int k = last.ref_pos + 1 + lane; k < K; k += 64) { ms[k] = -1; if (STOP) mp[k] = -1; }
    int carry_ref = -1, carry_read = -1;                       // p[i0 - 1]; prev_event_idx = -1 initially (:281)
    np_pair c_[4]; np_pair peek;
    auto load = [&](int i0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u + lane; c_[u] = i < np_ ? p[i] : np_pair{-2, -2}; }
        peek = i0 + 256 < np_ ? p[i0 + 256] : np_pair{-2, -2};   // (uniform address: one request)
    };
    load(0);
    for (int i0 = 0; i0 < np_; i0 += 256) {
        np_pair c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = c_[u];
        const np_pair pk = peek;
        if (i0 + 256 < np_) load(i0 + 256);                      // the next chunk's loads fly while this one is scattered
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + 64 * u >= np_) break;                        // wave-uniform
            // previous pair: lane - 1, lane 0 from the previous sub-chunk's lane 63 (or the carry)
            const int pr_ref = u == 0 ? carry_ref : __builtin_amdgcn_readlane(c[u == 0 ? 0 : u - 1].ref_pos, 63);
            const int pr_read = u == 0 ? carry_read : __builtin_amdgcn_readlane(c[u == 0 ? 0 : u - 1].read_pos, 63);
            const int a_ref = wave_shr1_i(c[u].ref_pos, pr_ref), a_read = wave_shr1_i(c[u].read_pos, pr_read);
            // next pair: lane + 1, lane 63 from the next sub-chunk's lane 0 (or the peeked pair); -2 past the end
            const int nx_ref0 = u == 3 ? pk.ref_pos : __builtin_amdgcn_readlane(c[u == 3 ? 3 : u + 1].ref_pos, 0);
            const int nx_read0 = u == 3 ? pk.read_pos : __builtin_amdgcn_readlane(c[u == 3 ? 3 : u + 1].read_pos, 0);
            const int n_ref = wave_shl1_i(c[u].ref_pos, nx_ref0), n_read = wave_shl1_i(c[u].read_pos, nx_read0);
            const int i = i0 + 64 * u + lane;
            if (i < np_) {
                const int k = c[u].ref_pos, e = c[u].read_pos;
                const bool first_of_k = a_ref != k, last_of_k = n_ref != k;
                const bool records = e != a_read;
                if (first_of_k) ms[k] = records ? e : (last_of_k ? -1 : n_read);
                if (STOP && last_of_k) mp[k] = (records || !first_of_k) ? e : -1;
            }
        }
        carry_ref = __builtin_amdgcn_readlane(c[3].ref_pos, 63); carry_read = __builtin_amdgcn_readlane(c[3].read_pos, 63);
    }
    if (lane == 0) {
        const size_t min_event = (size_t)first.read_pos, max_event = (size_t)last.read_pos;        // path is monotone
        const double epb = (double)(max_event - min_event) / (double)(size_t)K;                    // :301
        events_per_base[ri] = epb;
        np_transitions(epb, indel_bias, rd->trans);
    }
}

// ---------------------------------------------------------------------------------------------------------
// f1: recalibrate_model(read, base_model, strand, alignment, scale_var = true, scale_drift = false)
//     (src/nanopolish_methyltrain.cpp:204-306) on the 'M' entries of get_eventalignment_for_1d_basecalls
//     (src/nanopolish_squiggle_read.cpp:339-389).  One wavefront per read.
// Each k-mer that has events contributes exactly one candidate 'M' entry -- its first event -- and it is 'M' iff
// its rank differs from the rank of the previous k-mer that has events (all events of one k-mer share its rank, so
// later events of the run are 'E').  The five normal-equation sums and the residual sum are accumulated in the
// reference's order (ascending k-mer): lanes form the terms 64 at a time, stage them in LDS, and one lane per sum adds
// its row front to back.
// The 2x2 solve restates Eigen's FullPivLU (see oracle/np_oracle.c:eigen_fullpivlu_solve_2x2).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ int64_t uniform_i64(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void fullpivlu_solve_2x2(double a00, double a01, double a10, double a11, double b0, double b1,
                                                    double& x0, double& x1)
{
    double m00 = a00, m01 = a01, m10 = a10, m11 = a11;
    int pr = 0, pc = 0; double big = fabs(m00);                    // first maximum in column-major visiting order
    if (fabs(m10) > big) { big = fabs(m10); pr = 1; pc = 0; }
    if (fabs(m01) > big) { big = fabs(m01); pr = 0; pc = 1; }
    if (fabs(m11) > big) { big = fabs(m11); pr = 1; pc = 1; }
    x0 = x1 = 0.0;
    if (big == 0.0) return;
    if (pr == 1) { double t = m00; m00 = m10; m10 = t; t = m01; m01 = m11; m11 = t; }
    if (pc == 1) { double t = m00; m00 = m01; m01 = t; t = m10; m10 = m11; m11 = t; }
    m10 /= m00;
    m11 -= m10 * m01;
    const double maxpivot = fabs(m00) > fabs(m11) ? fabs(m00) : fabs(m11);
    const double thr = maxpivot * (2.220446049250313e-16 * 2);
    const int rank = (fabs(m00) > thr) + (fabs(m11) > thr);
    double c0 = pr == 1 ? b1 : b0, c1 = pr == 1 ? b0 : b1;
    c1 -= m10 * c0;
    double y1 = 0.0;
    if (rank == 2) { y1 = c1 / m11; c0 -= y1 * m01; }
    const double y0 = c0 / m00;
    if (pc == 1) { x0 = y1; x1 = y0; } else { x0 = y0; x1 = y1; }
}

// Round 5.  Two things bounded round 4's kernel (one read per 64-thread workgroup, 7.7 ms per 100 000 reads, 76 % of its wave-cycles
// waiting): (1) the model look-ups -- two doubles out of a 32-byte state per k-mer, 64 random 128-byte lines per wave instruction from a
// table that does not fit the L1: ~170 GB of L2 -> L1 traffic per pass over the batch; (2) the ordered sums -- one fp64 addition per
// term and sum, in k-mer order, whatever the hardware: a serial phase of 64 steps per chunk with 5 of 64 lanes busy.
// Now a workgroup of W waves keeps the model in LDS -- (level_mean, 1 / level_stdv^2) for the first pass, (level_mean, level_stdv^2)
// for the second, 64 KB for the 4 096 states of a 6-mer base model, the division done once per state and workgroup instead of once per
// term -- and a wave carries R reads: it forms the terms of a 64-k-mer chunk of EACH of its reads (lanes = k-mers, one LDS tile of
// 5 x 65 doubles per read; the row stride keeps the serial readers on distinct banks), then lane (r, c) adds the 64 terms of sum c of
// read r -- one serial phase per R reads.  The next chunk's map entries and ranks are requested before the current chunk is consumed.
// A read shorter than its group's longest contributes zero terms past its end (a zero term leaves a non-negative-zero sum unchanged),
// so the groups are made of reads of similar length (the aligner's longest-first order) only for efficiency.  Same terms, same order,
// same roundings as before: shift / scale / var / log_var bit-identical (tests/test_gpu_parity.py::test_calibrated_pass...).
// TABLE = false: a base model of more than NP_RC_STATES states (none of the kits') reads the states from memory as round 4 did.
#define NP_RC_STATES 4096
template <int W, int R, bool TABLE>
__global__ void __launch_bounds__(64 * W) np_recalibrate_kernel(int n_reads, np_read_dev* reads, const float* event_mean,
                                                                const uint16_t* ranks, const np_state_dev* model, int n_states,
                                                                const int32_t* n_pairs, const int32_t* map_start,
                                                                int32_t* calibrated, const uint32_t* order)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ double terms[W][R][5][65];
    __shared__ double2 table[TABLE ? NP_RC_STATES : 1];         // .x = level_mean, .y = 1 / level_stdv^2 (pass 0) or level_stdv^2 (pass 1)
    int ri[R], K[R]; bool live[R];
    const int32_t* ms[R]; const uint16_t* rk[R]; const float* ev[R];
    int maxK = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int slot = (blockIdx.x * W + wave) * R + r;
        // (readfirstlane: the per-read values are wave-uniform; told so, the compiler keeps them and the loop control in scalar registers)
        ri[r] = slot < n_reads ? __builtin_amdgcn_readfirstlane(order ? (int)order[slot] : slot) : -1;
        live[r] = false; K[r] = 0; ms[r] = nullptr; rk[r] = nullptr; ev[r] = nullptr;
        if (ri[r] >= 0) {
            if (__builtin_amdgcn_readfirstlane(n_pairs[ri[r]]) <= 0) { if (lane == 0) calibrated[ri[r]] = 0; }
            else {
                const np_read_dev* rd = reads + ri[r];
                live[r] = true; K[r] = __builtin_amdgcn_readfirstlane((int)rd->n_kmers);
                const int64_t ro = uniform_i64(rd->rank_off), eo = uniform_i64(rd->event_off);
                ms[r] = map_start + ro; rk[r] = ranks + ro; ev[r] = event_mean + eo;
                maxK = K[r] > maxK ? K[r] : maxK;
            }
        }
    }
    double shift[R], scale[R];
    long long n[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { shift[r] = 0.0; scale[r] = 0.0; n[r] = 0; }
    // serial-phase role of this lane: pass 0 lane 5 r + c owns sum c of read r; pass 1 lane r owns read r's residual sum
    const int sr0 = lane / 5, sc0 = lane - 5 * sr0;
    const unsigned long long below = (1ull << lane) - 1ull;
    double acc = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        if (TABLE) {
            if (pass == 1) __syncthreads();                      // every wave is done with the first pass's table
            for (int q = threadIdx.x; q < n_states; q += 64 * W) {
                const double ls = model[q].level_stdv, v = ls * ls;
                table[q] = double2{model[q].level_mean, pass == 0 ? 1. / v : v};
            }
            __syncthreads();
        }
        int carry_rank[R];                                       // prev_kmer_rank = -1 (squiggle_read.cpp:351)
#pragma unroll
        for (int r = 0; r < R; ++r) carry_rank[r] = -1;
        const double* row = pass == 0 ? &terms[wave][sr0 < R ? sr0 : 0][sc0][0] : &terms[wave][lane < R ? lane : 0][0][0];
        const bool adder = pass == 0 ? lane < 5 * R : lane < R;
        // Two chunks in flight ahead of the one being consumed: chunk c + 2's map entries and ranks (coalesced) and chunk c + 1's event
        // means (a gather through the map entries that arrived during chunk c - 1) are requested before chunk c's terms are formed, so
        // every request has a whole chunk -- the serial phase included -- to land
        int st_a[R], rank_a[R];                                  // map entries and ranks of the chunk after next
        int st_c[R], rank_c[R]; float e_c[R];                    // the next chunk: map entries, ranks, event means
        auto request = [&](int base0) {
            const int ki = base0 + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool in = live[r] && ki < K[r];
                st_a[r] = in ? ms[r][ki] : -1;
                rank_a[r] = in ? (int)rk[r][ki] : 0;
            }
        };
        auto gather = [&]() {                                    // (st_a, rank_a) -> (st_c, rank_c), and the event means they point at
#pragma unroll
            for (int r = 0; r < R; ++r) {
                st_c[r] = st_a[r]; rank_c[r] = rank_a[r];
                e_c[r] = st_c[r] != -1 ? ev[r][st_c[r]] : 0.0f; // raw_events: get_unscaled_level of the run's first event
            }
        };
        request(0);
        gather();
        request(64);
        for (int base0 = 0; base0 < maxK; base0 += 64) {
            int st_[R], rank_[R];
            double mu_[R], x_[R]; float e_[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { st_[r] = st_c[r]; rank_[r] = rank_c[r]; e_[r] = e_c[r]; }
            if (base0 + 64 < maxK) { gather(); request(base0 + 128); }
            // the states of the k-mers that have events (a superset of the 'M' entries)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool has = st_[r] != -1;
                if (TABLE) { const double2 t = table[rank_[r]]; mu_[r] = t.x; x_[r] = t.y; }
                else {
                    const double ls = has ? model[rank_[r]].level_stdv : 1.0, v = ls * ls;
                    mu_[r] = has ? model[rank_[r]].level_mean : 0.0; x_[r] = pass == 0 ? 1. / v : v;
                }
            }
            unsigned long long any = 0ull;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool has = st_[r] != -1;
                const int rank = has ? rank_[r] : -1;
                const unsigned long long hm = __ballot(has);
                const unsigned long long before = hm & below;
                const int src = before ? 63 - __clzll((long long)before) : 0;
                const int prev = __shfl(rank, src, 64);
                const bool isM = has && rank != (before ? prev : carry_rank[r]);
                // branch-free: a lane that is not an 'M' entry forms its terms with a zero factor (mu and e are finite; a -0.0 term
                // leaves every sum unchanged, like +0.0)
                const double mu = mu_[r], e = (double)e_[r];
                double t0, t1 = 0., t2 = 0., t3 = 0., t4 = 0.;
                if (pass == 0) {
                    const double inv_var = isM ? x_[r] : 0.0;    // 1. / (ls * ls)
                    t0 = inv_var; t1 = mu * inv_var; t2 = mu * mu * inv_var; t3 = e * inv_var; t4 = mu * e * inv_var;
                } else {
                    const double yi = (e - shift[r] - scale[r] * mu);
                    t0 = (isM ? yi * yi : 0.0) / x_[r];           // / (ls * ls)
                }
                terms[wave][r][0][lane] = t0;
                if (pass == 0) { terms[wave][r][1][lane] = t1; terms[wave][r][2][lane] = t2; terms[wave][r][3][lane] = t3; terms[wave][r][4][lane] = t4; }
                const unsigned long long mm = __ballot(isM);
                n[r] += __popcll(mm);
                any |= mm;
                if (hm) carry_rank[r] = __builtin_amdgcn_readlane(rank, 63 - __clzll((long long)hm));      // (a wave-uniform lane index)
            }
            // ordered accumulation: lane (r, c) adds its 64 terms in k-mer order (a zero term leaves a non-negative-zero sum unchanged,
            // so lanes that are not 'M' entries, and reads that have ended, need no masking).  The tiles are this wave's own: LDS
            // operations of one wave complete in order, the fences only keep the compiler from moving them across each other.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (any && adder) {
#pragma unroll 16
                for (int q = 0; q < 64; ++q) acc += row[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (pass == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!live[r]) continue;
                if (n[r] < 200) { if (lane == 0) calibrated[ri[r]] = 0; live[r] = false; continue; }      // minNumEventsToRescale: not recalibrated
                const double a00 = readlane_f64(acc, 5 * r), a01 = readlane_f64(acc, 5 * r + 1), a11 = readlane_f64(acc, 5 * r + 2);
                const double b0 = readlane_f64(acc, 5 * r + 3), b1 = readlane_f64(acc, 5 * r + 4);
                fullpivlu_solve_2x2(a00, a01, a01, a11, b0, b1, shift[r], scale[r]);
                n[r] = 0;
            }
            acc = 0.0;
            maxK = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) if (live[r] && K[r] > maxK) maxK = K[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!live[r]) continue;
        double var = readlane_f64(acc, r);
        var /= (double)(unsigned long long)n[r];
        var = sqrt(var);
        if (lane == 0) {
            np_read_dev* rd = reads + ri[r];
            rd->shift = shift[r]; rd->scale = scale[r]; rd->var = var; rd->log_var = np_log_glibc(var);   // set4 (squiggle_read.cpp:38-65), glibc's log restated
            calibrated[ri[r]] = var > 2.5 ? 0 : 1;                                          // MIN_CALIBRATION_VAR (:320)
        }
    }
}

// Round 6.  The kernel above holds 148 736 B of LDS per 512-thread workgroup -- the 64 KB table plus 83 KB of term tiles --
// so a CU runs ONE workgroup, two waves per SIMD, and those waves wait on a counter for 54 % of their cycles (profiles/r05_glue.md).  The table
// cannot shrink (two doubles per state: the model's levels are decimal fractions); the tiles can: here a chunk is 32 k-mers and the two HALVES
// of a wave form the terms of two different reads at once (lanes 0-31: read 2p, lanes 32-63: read 2p + 1), so a read's tile is 5 x 33 doubles
// instead of 5 x 65 and SIXTEEN waves of four reads fit beside the table (84 480 + 65 536 B): one 1 024-thread workgroup = four waves per
// SIMD.  Per 64 k-mers of a read the wave issues what it issued before -- half as many term-forming instructions per chunk (each covers two
// reads), twice as many chunks, the ordered additions 32 at a time -- and the terms, their order and every rounding are unchanged.
template <int W, int R>
__global__ void __launch_bounds__(64 * W) np_recalibrate_half_kernel(int n_reads, np_read_dev* reads, const float* event_mean,
                                                                     const uint16_t* ranks, const np_state_dev* model, int n_states,
                                                                     const int32_t* n_pairs, const int32_t* map_start,
                                                                     int32_t* calibrated, const uint32_t* order)
{
    static_assert(R % 2 == 0 && 5 * R <= 64, "reads per wave: an even number, five adder lanes each");
    constexpr int P = R / 2, CH = 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = lane >> 5, hl = lane & 31;
    __shared__ double terms[W][R][5][CH + 1];
    __shared__ double2 table[NP_RC_STATES];                     // .x = level_mean, .y = 1 / level_stdv^2 (pass 0) or level_stdv^2 (pass 1)
    int ri[R], K[R]; bool live[R];
    int maxK = 0;
    // this lane's read of pair p is 2 p + half: its arrays, length and liveness as per-lane values
    const int32_t* msl[P]; const uint16_t* rkl[P]; const float* evl[P]; int Kl[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { msl[p] = nullptr; rkl[p] = nullptr; evl[p] = nullptr; Kl[p] = 0; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int slot = (blockIdx.x * W + wave) * R + r;
        ri[r] = slot < n_reads ? __builtin_amdgcn_readfirstlane(order ? (int)order[slot] : slot) : -1;
        live[r] = false; K[r] = 0;
        if (ri[r] >= 0) {
            if (__builtin_amdgcn_readfirstlane(n_pairs[ri[r]]) <= 0) { if (lane == 0) calibrated[ri[r]] = 0; }
            else {
                const np_read_dev* rd = reads + ri[r];
                live[r] = true; K[r] = __builtin_amdgcn_readfirstlane((int)rd->n_kmers);
                const int64_t ro = uniform_i64(rd->rank_off), eo = uniform_i64(rd->event_off);
                if (half == (r & 1)) { msl[r >> 1] = map_start + ro; rkl[r >> 1] = ranks + ro; evl[r >> 1] = event_mean + eo; Kl[r >> 1] = K[r]; }
                maxK = K[r] > maxK ? K[r] : maxK;
            }
        }
    }
    double shift[R], scale[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { shift[r] = 0.0; scale[r] = 0.0; }
    int nl[P];                                                   // 'M' entries of this lane's read of pair p so far (equal in all lanes of a half)
#pragma unroll
    for (int p = 0; p < P; ++p) nl[p] = 0;
    const int sr0 = lane / 5, sc0 = lane - 5 * sr0;             // pass 0: lane 5 r + c owns sum c of read r; pass 1: lane r owns read r's residual sum
    const uint32_t below = (1u << hl) - 1u;
    double acc = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) __syncthreads();                          // every wave is done with the first pass's table
        for (int q = threadIdx.x; q < n_states; q += 64 * W) {
            const double ls = model[q].level_stdv, v = ls * ls;
            table[q] = double2{model[q].level_mean, pass == 0 ? 1. / v : v};
        }
        __syncthreads();
        int carry[P];                                            // prev_kmer_rank = -1 (squiggle_read.cpp:351), per half
        double shl[P], scl[P];                                   // this lane's read's shift / scale (pass 1)
#pragma unroll
        for (int p = 0; p < P; ++p) { carry[p] = -1; shl[p] = half ? shift[2 * p + 1] : shift[2 * p]; scl[p] = half ? scale[2 * p + 1] : scale[2 * p]; }
        const double* row = pass == 0 ? &terms[wave][sr0 < R ? sr0 : 0][sc0][0] : &terms[wave][lane < R ? lane : 0][0][0];
        const bool adder = pass == 0 ? lane < 5 * R : lane < R;
        // two chunks in flight ahead of the one being consumed (as in the kernel above): chunk c + 2's map entries and ranks, chunk c + 1's event means
        int st_a[P], rank_a[P];
        int st_c[P], rank_c[P]; float e_c[P];
        auto request = [&](int base0) {
            const int ki = base0 + hl;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const bool in = ki < Kl[p];                      // (a dead read has Kl = 0)
                st_a[p] = in ? msl[p][ki] : -1;
                rank_a[p] = in ? (int)rkl[p][ki] : 0;
            }
        };
        auto gather = [&]() {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                st_c[p] = st_a[p]; rank_c[p] = rank_a[p];
                e_c[p] = st_c[p] != -1 ? evl[p][st_c[p]] : 0.0f;
            }
        };
        request(0);
        gather();
        request(CH);
        for (int base0 = 0; base0 < maxK; base0 += CH) {
            int st_[P], rank_[P]; float e_[P];
#pragma unroll
            for (int p = 0; p < P; ++p) { st_[p] = st_c[p]; rank_[p] = rank_c[p]; e_[p] = e_c[p]; }
            if (base0 + CH < maxK) { gather(); request(base0 + 2 * CH); }
            unsigned long long any = 0ull;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const bool has = st_[p] != -1;
                const double2 t = table[rank_[p]];
                const int rank = has ? rank_[p] : -1;
                const unsigned long long hm64 = __ballot(has);
                const uint32_t hm = half ? (uint32_t)(hm64 >> 32) : (uint32_t)hm64;          // the 'has' lanes of this lane's half
                const uint32_t before = hm & below;
                const int src = (half << 5) + (before ? 31 - __clz((int)before) : 0);
                const int prev = __shfl(rank, src, 64);
                const bool isM = has && rank != (before ? prev : carry[p]);
                const double mu = t.x, e = (double)e_[p];
                double t0, t1 = 0., t2 = 0., t3 = 0., t4 = 0.;
                if (pass == 0) {
                    const double inv_var = isM ? t.y : 0.0;     // 1. / (ls * ls)
                    t0 = inv_var; t1 = mu * inv_var; t2 = mu * mu * inv_var; t3 = e * inv_var; t4 = mu * e * inv_var;
                } else {
                    const double yi = (e - shl[p] - scl[p] * mu);
                    t0 = (isM ? yi * yi : 0.0) / t.y;            // / (ls * ls)
                }
                double (*tile)[CH + 1] = terms[wave][2 * p + half];
                tile[0][hl] = t0;
                if (pass == 0) { tile[1][hl] = t1; tile[2][hl] = t2; tile[3][hl] = t3; tile[4][hl] = t4; }
                const unsigned long long mm64 = __ballot(isM);
                nl[p] += __popc(half ? (uint32_t)(mm64 >> 32) : (uint32_t)mm64);
                any |= mm64;
                // the rank of the half's last k-mer with events carries into the next chunk
                const int last = __shfl(rank, (half << 5) + (hm ? 31 - __clz((int)hm) : 0), 64);
                if (hm) carry[p] = last;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (any && adder) {
#pragma unroll 16
                for (int q = 0; q < CH; ++q) acc += row[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (pass == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!live[r]) continue;
                const int n_r = __builtin_amdgcn_readlane(nl[r >> 1], (r & 1) << 5);
                if (n_r < 200) {                                 // minNumEventsToRescale: not recalibrated
                    if (lane == 0) calibrated[ri[r]] = 0;
                    live[r] = false;
                    if (half == (r & 1)) Kl[r >> 1] = 0;
                    continue;
                }
                const double a00 = readlane_f64(acc, 5 * r), a01 = readlane_f64(acc, 5 * r + 1), a11 = readlane_f64(acc, 5 * r + 2);
                const double b0 = readlane_f64(acc, 5 * r + 3), b1 = readlane_f64(acc, 5 * r + 4);
                fullpivlu_solve_2x2(a00, a01, a01, a11, b0, b1, shift[r], scale[r]);
            }
#pragma unroll
            for (int p = 0; p < P; ++p) nl[p] = 0;
            acc = 0.0;
            maxK = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) if (live[r] && K[r] > maxK) maxK = K[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!live[r]) continue;
        const int n_r = __builtin_amdgcn_readlane(nl[r >> 1], (r & 1) << 5);
        double var = readlane_f64(acc, r);
        var /= (double)(unsigned long long)n_r;
        var = sqrt(var);
        if (lane == 0) {
            np_read_dev* rd = reads + ri[r];
            rd->shift = shift[r]; rd->scale = scale[r]; rd->var = var; rd->log_var = np_log_glibc(var);   // set4 (squiggle_read.cpp:38-65), glibc's log restated
            calibrated[ri[r]] = var > 2.5 ? 0 : 1;                                          // MIN_CALIBRATION_VAR (:320)
        }
    }
}

// Thread per work item.  The two sequences of a methylation group (unmethylated, methylated) sit side by side and share their window
// bounds: the odd lane takes the even lane's result (DPP) instead of repeating the two closest-event searches -- checked per pair, not
// assumed: neighbours of different reads or bounds each search for themselves.  Only the item's second half (e_start, e_stop, stride,
// flags: 16 of its 32 bytes) is read and written.
__device__ __forceinline__ void resolve_item(bool valid, int64_t jj, np_hmm_job_dev* jobs, const np_read_dev* reads, const int32_t* n_pairs,
                                             const double* events_per_base, const int32_t* calibrated, const int32_t* map_start, const int32_t* kpos)
{
    static_assert(sizeof(np_hmm_job_dev) == 32 && offsetof(np_hmm_job_dev, e_start) == 16 && offsetof(np_hmm_job_dev, read) == 12, "np_hmm_job_dev layout");
    const int read = (int)jobs[jj].read;
    uint4* tail = reinterpret_cast<uint4*>(reinterpret_cast<char*>(jobs + jj) + 16);
    uint4 t = *tail;                                                             // e_start, e_stop, stride, flags
    const int2 kp = *reinterpret_cast<const int2*>(kpos + 2 * jj);
    const int k1 = kp.x, k2 = kp.y;
    // the even neighbour's item (lane - 1; lane 0 is even, an odd lane always has its partner in the wave)
    const int p_read = wave_shr1_i(read, -1), p_k1 = wave_shr1_i(k1, -1), p_k2 = wave_shr1_i(k2, -2);
    const bool share = (threadIdx.x & 1) && p_read == read && p_k1 == k1 && p_k2 == k2;
    int e1 = -1, e2 = -1; int ok = 0;
    if (!share) {
        // failed alignment, failed calibration (squiggle_read.cpp:320-323) or events-per-base QC (:332): no events, no scoring
        bool o = n_pairs[read] > 0 && !(events_per_base[read] > 5.0) && (!calibrated || calibrated[read] != 0);
        if (o) {
            const np_read_dev* rd = reads + read;
            const int32_t* ms = map_start + rd->rank_off;
            const int K = (int)rd->n_kmers;
            if (k1 < 0 || k1 >= K || k2 < 0 || k2 >= K) o = false;
            else {
                e1 = closest_event(ms, K, k1);
                e2 = closest_event(ms, K, k2);
                const int d = e2 - e1;
                if (e1 < 0 || e2 < 0 || (d < 0 ? -d : d) <= 10) o = false;      // basemods.cpp:356
            }
        }
        ok = o ? 1 : 0;
    }
    const int q1 = wave_shr1_i(e1, -1), q2 = wave_shr1_i(e2, -1), qok = wave_shr1_i(ok, 0);
    if (share) { e1 = q1; e2 = q2; ok = qok; }
    if (ok) {
        t.x = (uint32_t)e1; t.y = (uint32_t)e2;
        t.z = (uint32_t)(e1 <= e2 ? 1 : -1);                                     // basemods.cpp:370
        t.w &= ~(uint32_t)NP_JOB_SKIP;
    } else {
        t.x = 0; t.y = 0; t.z = 1;
        t.w |= (uint32_t)NP_JOB_SKIP;                                            // classify drops it, score = NaN
    }
    if (valid) *tail = t;
}

__global__ void __launch_bounds__(256) np_resolve_kernel(int64_t n_jobs, np_hmm_job_dev* jobs, const np_read_dev* reads,
                                                         const int32_t* n_pairs, const double* events_per_base,
                                                         const int32_t* calibrated, const int32_t* map_start, const int32_t* kpos)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    resolve_item(j < n_jobs, j < n_jobs ? j : n_jobs - 1, jobs, reads, n_pairs, events_per_base, calibrated, map_start, kpos);
}

// ---- the same kernels over a SLOT LAYOUT (np_set_job_layout, round 5) ---------------------------------------------------------------
// The device builder lays work items out in per-read slot ranges sized for the worst case (one group per min_separation + 1 bases:
// 497 slots for a 5 450-base read that has 182 groups), and every kernel that walked "all n_jobs items" spent 63 % of its memory traffic
// on slots that hold nothing -- a 32-byte sector per slot whether 4 or 32 bytes of it are wanted (profiles/r05_pmc.json: np_resolve_kernel
// 62 KB fetched + 32 KB written per read).  With the layout a workgroup takes NP_SLOT_RPB consecutive reads and visits their LIVE items
// only; unused slots are neither read nor written (their scores are NaN from one fill).
#define NP_SLOT_RPB 12
struct slot_block {
    int64_t base[NP_SLOT_RPB];
    int pre[NP_SLOT_RPB + 1];
};
__device__ __forceinline__ int slot_block_init(slot_block& B, np_slots L)
{
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int q = 0; q < NP_SLOT_RPB; ++q) {
            const int r = blockIdx.x * NP_SLOT_RPB + q;
            const int ng = r < L.n_reads ? L.n_groups[r] : 0;
            B.base[q] = r < L.n_reads ? 2 * L.group_off[r] : 0;
            B.pre[q] = acc;
            acc += 2 * (ng > 0 ? ng : 0);
        }
        B.pre[NP_SLOT_RPB] = acc;
    }
    __syncthreads();
    return B.pre[NP_SLOT_RPB];
}
__device__ __forceinline__ int64_t slot_item(const slot_block& B, int idx)       // idx < B.pre[NP_SLOT_RPB]
{
    int q = 0;
#pragma unroll
    for (int t = 1; t < NP_SLOT_RPB; ++t) q += idx >= B.pre[t] ? 1 : 0;
    return B.base[q] + (idx - B.pre[q]);
}

__global__ void __launch_bounds__(256) np_resolve_slots_kernel(np_slots L, np_hmm_job_dev* jobs, const np_read_dev* reads, const int32_t* n_pairs,
                                                               const double* events_per_base, const int32_t* calibrated, const int32_t* map_start,
                                                               const int32_t* kpos)
{
    __shared__ slot_block B;
    const int total = slot_block_init(B, L);
    // (a read's live items are an even number, so an item's parity is its thread's: the odd lane's partner is lane - 1 here too)
    for (int i0 = 0; i0 < total; i0 += 256) {
        const int idx = i0 + threadIdx.x;
        resolve_item(idx < total, slot_item(B, idx < total ? idx : total - 1), jobs, reads, n_pairs, events_per_base, calibrated, map_start, kpos);
    }
}

// (size_class / job_bin: np_kernels.h -- the host entry points bin small batches themselves)

// NP_BIN_ITEMS work items per workgroup: the workgroup's histogram (7 KB of LDS) is cleared and flushed once per 4096 items,
// and the global counters -- a few hot bins take nearly all items -- see one atomic per bin and workgroup instead of sixteen.
#define NP_BIN_ITEMS 4096
__global__ void __launch_bounds__(256) np_bin_count_kernel(const np_hmm_job_dev* jobs, int64_t n_jobs, uint32_t* hist,
                                                           float* out_scores, uint32_t flank_len)
{
    __shared__ uint32_t h[NP_NBINS];
    for (int i = threadIdx.x; i < NP_NBINS; i += 256) h[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * NP_BIN_ITEMS;
#pragma unroll 4
    for (int t = 0; t < NP_BIN_ITEMS / 256; ++t) {
        const int64_t j = base + t * 256 + threadIdx.x;
        if (j < n_jobs) {
            const int bin = np_job_bin(jobs[j], flank_len);
            if (bin >= 0) atomicAdd(&h[bin], 1u);
            else if (out_scores) out_scores[j] = __builtin_nanf("");          // skipped / unsupported item
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NP_NBINS; i += 256) if (h[i]) atomicAdd(&hist[i], h[i]);
}

// one workgroup: per class, exclusive scan over its buckets -> cursor[]; class_count[] = class totals
__global__ void __launch_bounds__(64) np_bin_scan_kernel(const uint32_t* hist, uint32_t* cursor, uint32_t* class_count)
{
    const int c = threadIdx.x;
    if (c >= NP_NUM_CLASSES) return;
    uint32_t run = 0;
    for (int b = 0; b < NP_CPL * NP_EBUCKETS; ++b) { cursor[c * NP_CPL * NP_EBUCKETS + b] = run; run += hist[c * NP_CPL * NP_EBUCKETS + b]; }
    class_count[c] = run;
}

__global__ void __launch_bounds__(256) np_bin_scatter_kernel(const np_hmm_job_dev* jobs, int64_t n_jobs, uint32_t* cursor,
                                                             uint32_t* order, uint32_t flank_len)
{
    __shared__ uint32_t h[NP_NBINS];       // per-block counts, then per-block base offsets
    for (int i = threadIdx.x; i < NP_NBINS; i += 256) h[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * NP_BIN_ITEMS;
    int bin[NP_BIN_ITEMS / 256];
    uint32_t local[NP_BIN_ITEMS / 256];
#pragma unroll
    for (int t = 0; t < NP_BIN_ITEMS / 256; ++t) {
        const int64_t j = base + t * 256 + threadIdx.x;
        bin[t] = -1; local[t] = 0;
        if (j < n_jobs) {
            bin[t] = np_job_bin(jobs[j], flank_len);
            if (bin[t] >= 0) local[t] = atomicAdd(&h[bin[t]], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NP_NBINS; i += 256) { const uint32_t n = h[i]; if (n) h[i] = atomicAdd(&cursor[i], n); }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NP_BIN_ITEMS / 256; ++t)
        if (bin[t] >= 0)
            order[(size_t)(bin[t] / (NP_CPL * NP_EBUCKETS)) * (size_t)n_jobs + h[bin[t]] + local[t]] = (uint32_t)(base + t * 256 + threadIdx.x);
}

// the two binning passes over a slot layout: a workgroup's reads hold ~4 400 live items (12 x 2 x 182), the scatter takes them 4 096 at a time
__global__ void __launch_bounds__(256) np_bin_count_slots_kernel(np_slots L, const np_hmm_job_dev* jobs, uint32_t* hist, uint32_t flank_len)
{
    __shared__ slot_block B;
    __shared__ uint32_t h[NP_NBINS];
    for (int i = threadIdx.x; i < NP_NBINS; i += 256) h[i] = 0;
    const int total = slot_block_init(B, L);
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int bin = np_job_bin(jobs[slot_item(B, idx)], flank_len);
        if (bin >= 0) atomicAdd(&h[bin], 1u);                             // (a skipped item's score is NaN already: the launcher's fill)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NP_NBINS; i += 256) if (h[i]) atomicAdd(&hist[i], h[i]);
}

__global__ void __launch_bounds__(256) np_bin_scatter_slots_kernel(np_slots L, const np_hmm_job_dev* jobs, int64_t n_jobs, uint32_t* cursor,
                                                                   uint32_t* order, uint32_t flank_len)
{
    __shared__ slot_block B;
    __shared__ uint32_t h[NP_NBINS];
    const int total = slot_block_init(B, L);
    for (int c0 = 0; c0 < total; c0 += NP_BIN_ITEMS) {
        for (int i = threadIdx.x; i < NP_NBINS; i += 256) h[i] = 0;
        __syncthreads();
        int bin[NP_BIN_ITEMS / 256]; uint32_t local[NP_BIN_ITEMS / 256]; uint32_t item[NP_BIN_ITEMS / 256];
#pragma unroll
        for (int t = 0; t < NP_BIN_ITEMS / 256; ++t) {
            const int idx = c0 + t * 256 + threadIdx.x;
            bin[t] = -1; local[t] = 0; item[t] = 0;
            if (idx < total) {
                const int64_t j = slot_item(B, idx);
                item[t] = (uint32_t)j;
                bin[t] = np_job_bin(jobs[j], flank_len);
                if (bin[t] >= 0) local[t] = atomicAdd(&h[bin[t]], 1u);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < NP_NBINS; i += 256) { const uint32_t n = h[i]; if (n) h[i] = atomicAdd(&cursor[i], n); }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NP_BIN_ITEMS / 256; ++t)
            if (bin[t] >= 0)
                order[(size_t)(bin[t] / (NP_CPL * NP_EBUCKETS)) * (size_t)n_jobs + h[bin[t]] + local[t]] = item[t];
        __syncthreads();
    }
}

} // namespace

// bins: device scratch of 2 * NP_NBINS uint32 (histogram, cursors)
hipError_t np_launch_classify(const np_hmm_job_dev* jobs, int64_t n_jobs, uint32_t* class_count, uint32_t* order,
                              float* out_scores, uint32_t flank_len, uint32_t* bins, np_slots lay, hipStream_t s)
{
    if (n_jobs <= 0) return hipSuccess;
    uint32_t* hist = bins;
    uint32_t* cursor = bins + NP_NBINS;
    hipError_t e = hipMemsetAsync(bins, 0, 2 * NP_NBINS * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (lay.n_reads > 0) {
        // every item's score starts as NaN (what a skipped item and an unused slot end with); the kernels visit live items only
        if (out_scores) { e = hipMemsetD32Async((hipDeviceptr_t)out_scores, 0x7fc00000, (size_t)n_jobs, s); if (e != hipSuccess) return e; }
        const unsigned nbs = (unsigned)((lay.n_reads + NP_SLOT_RPB - 1) / NP_SLOT_RPB);
        hipLaunchKernelGGL(np_bin_count_slots_kernel, dim3(nbs), dim3(256), 0, s, lay, jobs, hist, flank_len);
        hipLaunchKernelGGL(np_bin_scan_kernel, dim3(1), dim3(64), 0, s, hist, cursor, class_count);
        hipLaunchKernelGGL(np_bin_scatter_slots_kernel, dim3(nbs), dim3(256), 0, s, lay, jobs, n_jobs, cursor, order, flank_len);
        return hipGetLastError();
    }
    const unsigned nb = (unsigned)((n_jobs + NP_BIN_ITEMS - 1) / NP_BIN_ITEMS);
    hipLaunchKernelGGL(np_bin_count_kernel, dim3(nb), dim3(256), 0, s, jobs, n_jobs, hist, out_scores, flank_len);
    hipLaunchKernelGGL(np_bin_scan_kernel, dim3(1), dim3(64), 0, s, hist, cursor, class_count);
    hipLaunchKernelGGL(np_bin_scatter_kernel, dim3(nb), dim3(256), 0, s, jobs, n_jobs, cursor, order, flank_len);
    return hipGetLastError();
}

hipError_t np_launch_build_map(int n_reads, np_read_dev* reads, const int64_t* pair_off, const np_pair* pairs,
                               const int32_t* pair_begin, const int32_t* n_pairs, int32_t* map_start, int32_t* map_stop,
                               double* events_per_base, double indel_bias, hipStream_t s)
{
    if (n_reads <= 0) return hipSuccess;
    if (map_stop) hipLaunchKernelGGL(np_build_map_kernel<true>, dim3(n_reads), dim3(64), 0, s, n_reads, reads, pair_off, pairs,
                                     pair_begin, n_pairs, map_start, map_stop, events_per_base, indel_bias);
    else hipLaunchKernelGGL(np_build_map_kernel<false>, dim3(n_reads), dim3(64), 0, s, n_reads, reads, pair_off, pairs,
                            pair_begin, n_pairs, map_start, map_stop, events_per_base, indel_bias);
    return hipGetLastError();
}

namespace {
// EventAlignmentRecord discards a record whose first and last aligned event coincide (alignment_db.cpp:83-86): every
// work item of such a read is unbounded.  deg_kpos holds, per read, the read-strand k-mer positions of the first and last
// aligned base that pass the record's filter (-1: none).  Thread per work item, after np_resolve_kernel.
__device__ __forceinline__ void discard_item(int64_t j, np_hmm_job_dev* jobs, const np_read_dev* reads, const int32_t* map_start, const int32_t* deg_kpos)
{
    np_hmm_job_dev job = jobs[j];
    if (job.flags & NP_JOB_SKIP) return;
    const np_read_dev* rd = reads + job.read;
    const int K = (int)rd->n_kmers;
    const int k1 = deg_kpos[2 * job.read], k2 = deg_kpos[2 * job.read + 1];
    bool drop = k1 < 0 || k1 >= K || k2 < 0 || k2 >= K;
    if (!drop) {
        const int32_t* ms = map_start + rd->rank_off;
        drop = closest_event(ms, K, k1) == closest_event(ms, K, k2);
    }
    if (drop) { job.e_start = 0; job.e_stop = 0; job.stride = 1; job.flags |= NP_JOB_SKIP; jobs[j] = job; }
}
__global__ void __launch_bounds__(256) np_discard_degenerate_kernel(int64_t n_jobs, np_hmm_job_dev* jobs, const np_read_dev* reads,
                                                                    const int32_t* map_start, const int32_t* deg_kpos)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < n_jobs) discard_item(j, jobs, reads, map_start, deg_kpos);
}
__global__ void __launch_bounds__(256) np_discard_degenerate_slots_kernel(np_slots L, np_hmm_job_dev* jobs, const np_read_dev* reads,
                                                                          const int32_t* map_start, const int32_t* deg_kpos)
{
    __shared__ slot_block B;
    const int total = slot_block_init(B, L);
    for (int idx = threadIdx.x; idx < total; idx += 256) discard_item(slot_item(B, idx), jobs, reads, map_start, deg_kpos);
}
} // namespace

hipError_t np_launch_discard_degenerate(int64_t n_jobs, np_hmm_job_dev* jobs, const np_read_dev* reads, const int32_t* map_start,
                                        const int32_t* deg_kpos, np_slots lay, hipStream_t s)
{
    if (n_jobs <= 0) return hipSuccess;
    if (lay.n_reads > 0) {
        hipLaunchKernelGGL(np_discard_degenerate_slots_kernel, dim3((unsigned)((lay.n_reads + NP_SLOT_RPB - 1) / NP_SLOT_RPB)), dim3(256), 0, s, lay, jobs, reads,
                           map_start, deg_kpos);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(np_discard_degenerate_kernel, dim3((unsigned)((n_jobs + 255) / 256)), dim3(256), 0, s, n_jobs, jobs, reads,
                       map_start, deg_kpos);
    return hipGetLastError();
}

hipError_t np_launch_resolve(int64_t n_jobs, np_hmm_job_dev* jobs, const np_read_dev* reads, const int32_t* n_pairs,
                             const double* events_per_base, const int32_t* calibrated, const int32_t* map_start,
                             const int32_t* kpos, np_slots lay, hipStream_t s)
{
    if (n_jobs <= 0) return hipSuccess;
    if (lay.n_reads > 0) {
        hipLaunchKernelGGL(np_resolve_slots_kernel, dim3((unsigned)((lay.n_reads + NP_SLOT_RPB - 1) / NP_SLOT_RPB)), dim3(256), 0, s, lay, jobs, reads, n_pairs,
                           events_per_base, calibrated, map_start, kpos);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(np_resolve_kernel, dim3((unsigned)((n_jobs + 255) / 256)), dim3(256), 0, s,
                       n_jobs, jobs, reads, n_pairs, events_per_base, calibrated, map_start, kpos);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Self-test: np_div_exact (np_device.h) against the IEEE fp32 divide on pseudo-random operand pairs drawn from
// the ranges the emission uses (numerator x - mean, denominator stdv).  Counts bitwise mismatches.
// ---------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ uint64_t splitmix64(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) np_selftest_div_kernel(uint64_t n_per_thread, uint64_t seed, unsigned long long* mismatches)
{
    uint64_t s = seed + 0xD1B54A32D192ED03ull * ((uint64_t)blockIdx.x * 256 + threadIdx.x + 1);
    unsigned long long bad = 0;
    for (uint64_t i = 0; i < n_per_thread; ++i) {
        const uint64_t u = splitmix64(s);
        // denominator: random mantissa, exponent in [2^-5, 2^6)  (stdv after scaling lives in ~[0.3, 12])
        const uint32_t dm = (uint32_t)u & 0x7fffffu, de = 122u + (uint32_t)((u >> 23) % 11u);
        const float d = __builtin_bit_cast(float, (de << 23) | dm);
        // numerator: random sign/mantissa, exponent in [2^-40, 2^15) and exact zero now and then
        const uint32_t nm = (uint32_t)(u >> 32) & 0x7fffffu, ne = 87u + (uint32_t)((u >> 55) % 55u);
        float n = __builtin_bit_cast(float, (ne << 23) | nm | ((uint32_t)(u >> 63) << 31));
        if ((u & 0xfff000000ull) == 0) n = 0.0f;
        const float r = (float)(1.0 / (double)d);
        const float want = n / d;
        const float got = np_div_exact(n, d, r);
        bad += (__builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, got));
    }
    if (bad) atomicAdd(mismatches, bad);
}
} // namespace

namespace {
// f4: one thread per CpG group.  The TSV writer prints diff = sum_ll_m - sum_ll_u (doubles; for 1D reads the float scores
// themselves) with "%.2lf" (src/nanopolish_call_methylation.cpp:538-545), and scripts/calculate_methylation_frequency.py:41-49
// parses that text back: llr = float(text); skip if abs(llr) < call_threshold * num_motifs; methylated iff llr > 0.
// printf's "%.2lf" is the correctly rounded (ties to even) decimal of the exact binary value: r = RN(100 x) as an integer,
// corrected by the exact residual 100 x - r (one fma: x is a difference of two floats, so 100 x - r is representable);
// float(text) is then the double nearest to r / 100, i.e. the correctly rounded quotient.
// The TSV's "%.2lf" round trip of the log-likelihood ratio and the frequency script's call rule
// (src/nanopolish_call_methylation.cpp:531-550, scripts/calculate_methylation_frequency.py:41-49): false = the group does not count
__device__ __forceinline__ bool site_call(float u, float m, int nm, double call_threshold, bool& methylated)
{
    const double x = (double)m - (double)u;
    if (!(__builtin_fabs(x) < __builtin_inf())) return false;           // NaN: a skipped group; +-inf never passes a real run
    double r = __builtin_rint(x * 100.0);
    const double e = __builtin_fma(x, 100.0, -r);
    const bool odd = __builtin_fmod(__builtin_fabs(r), 2.0) == 1.0;
    if (e > 0.5 || (e == 0.5 && odd)) r += 1.0;
    else if (e < -0.5 || (e == -0.5 && odd)) r -= 1.0;
    const double llr = r / 100.0;
    if (__builtin_fabs(llr) < call_threshold * (double)nm) return false; // ambiguous call
    methylated = llr > 0;
    return true;
}

__global__ void __launch_bounds__(256) np_site_table_kernel(int64_t n_groups, const float* scores, const int32_t* first_site,
                                                            const int32_t* n_motif, const np_hmm_job_dev* jobs, const int64_t* read_base,
                                                            double call_threshold, int64_t n_pos, int32_t* table)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    const int nm = n_motif[g];
    bool meth;
    if (!site_call(scores[2 * g], scores[2 * g + 1], nm, call_threshold, meth)) return;
    int64_t row = first_site[g];
    if (read_base) row += read_base[jobs[2 * g].read];
    if (row < 0 || row >= n_pos) return;
    atomicAdd(&table[3 * row], 1);
    atomicAdd(&table[3 * row + 1], nm);
    if (meth) atomicAdd(&table[3 * row + 2], nm);
}

// ---- the motif sites of the resident genome as a rank structure: one bit per base (a recognition site starts here, inside its contig) and the
// number of sites before every 64-base word.  rank(pos) = word_rank[pos >> 6] + popcount(mask[pos >> 6] below pos): the row of a site in a table
// that has one row per SITE (312 679 for the bench's 5 Mb genome: 7.5 MB instead of 120 MB; a 3.1 Gb genome: 28 M sites, 672 MB instead of 74 GB).
__global__ void __launch_bounds__(256) np_site_mask_kernel(const char* genome, const int64_t* contig_off, int n_contigs, int alphabet, int64_t n_pos, uint64_t* mask)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    if (p < n_pos) {
        int lo = 0, hi = n_contigs;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (contig_off[mid] <= p) lo = mid; else hi = mid; }
        const int64_t c0 = contig_off[lo];
        const sites_t S = sites_of(alphabet);
        hit = site_at(genome + c0, 0, (int)(contig_off[lo + 1] - c0), (int)(p - c0), S) >= 0;
    }
    const unsigned long long b = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && (p >> 6) < (n_pos + 63) / 64) mask[p >> 6] = b;
}
#define NP_RANK_CHUNK 2048        // words per workgroup of the scan
// pass 1: word_rank[w] = sites before word w INSIDE its chunk; chunk_total[c]
__global__ void __launch_bounds__(256) np_site_rank_local_kernel(const uint64_t* mask, int64_t n_words, uint32_t* word_rank, uint32_t* chunk_total)
{
    __shared__ uint32_t part[256];
    const int64_t w0 = (int64_t)blockIdx.x * NP_RANK_CHUNK + threadIdx.x * 8;
    uint32_t cnt[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { cnt[j] = w0 + j < n_words ? (uint32_t)__popcll(mask[w0 + j]) : 0u; sum += cnt[j]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (w0 + j < n_words) word_rank[w0 + j] = run; run += cnt[j]; }
    if (threadIdx.x == 255) chunk_total[blockIdx.x] = part[255];
}
// pass 2, one workgroup: chunk_total -> exclusive prefix in place; *n_sites = the total
__global__ void __launch_bounds__(1024) np_site_rank_chunks_kernel(uint32_t* chunk_total, int64_t n_chunks, int64_t* n_sites)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < n_chunks; c0 += 1024) {
        const int64_t c = c0 + threadIdx.x;
        const uint32_t mine = c < n_chunks ? chunk_total[c] : 0u;
        part[threadIdx.x] = mine;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        if (c < n_chunks) chunk_total[c] = carry + part[threadIdx.x] - mine;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_sites = (int64_t)carry;
}
// pass 3: add the chunk's base; word_rank[n_words] = the total
__global__ void __launch_bounds__(256) np_site_rank_add_kernel(uint32_t* word_rank, const uint32_t* chunk_total, int64_t n_words, const int64_t* n_sites)
{
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w < n_words) word_rank[w] += chunk_total[w / NP_RANK_CHUNK];
    if (w == n_words) word_rank[w] = (uint32_t)*n_sites;
}

// The same aggregation keyed as the reference keys it: (contig, start, end) of the group (nanopolish_call_methylation.cpp:532-550;
// calculate_methylation_frequency.py:16-23 -- `key = (c, start, end)`), for reads that OVERLAP on a genome.  A read's groups are its motif sites
// chained by gaps <= min_separation (basemods.cpp:306-320), i.e. the intersection of a GENOME cluster with the read's segment: a read that ends
// (or starts) inside a cluster reports a shorter group with another end (start), and the script counts that key on its own.  Two dense tables
// hold every such key without a hash: a group whose end IS its cluster's end is keyed by its start (columns 0-2: for a given start that end is
// unique); a group with its cluster's start but an earlier end is keyed by its end (columns 3-5); a group cut on both sides -- a read shorter than
// one cluster -- is counted in *n_overflow and left out (none in any run so far).  Whether a position ends / starts its cluster is read off the
// resident genome: no motif site within min_separation after / before it, inside the contig.
__global__ void __launch_bounds__(256) np_site_table_genome_kernel(int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* last_site,
                                                                   const int32_t* n_motif, const np_hmm_job_dev* jobs, const int64_t* read_base,
                                                                   const char* genome, const int64_t* contig_off, int n_contigs, int alphabet,
                                                                   int min_separation, double call_threshold, int64_t n_pos, int32_t* table,
                                                                   unsigned long long* n_overflow, const uint64_t* site_mask, const uint32_t* word_rank)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    const int nm = n_motif[g];
    bool meth;
    if (!site_call(scores[2 * g], scores[2 * g + 1], nm, call_threshold, meth)) return;
    const int64_t base = read_base[jobs[2 * g].read];
    const int64_t s = base + first_site[g], e = base + last_site[g];
    if (s < 0 || e >= n_pos || e < s) return;
    int lo = 0, hi = n_contigs;                                            // the contig that holds s: contig_off[lo] <= s < contig_off[lo + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (contig_off[mid] <= s) lo = mid; else hi = mid; }
    const int64_t c0 = contig_off[lo];
    const int clen = (int)(contig_off[lo + 1] - c0);
    const char* ref = genome + c0;
    const sites_t S = sites_of(alphabet);
    const int ls = (int)(s - c0), le = (int)(e - c0);
    bool end_is_clusters = true, start_is_clusters = true;
    for (int d = 1; d <= min_separation; ++d) {
        if (site_at(ref, 0, clen, le + d, S) >= 0) end_is_clusters = false;
        if (site_at(ref, 0, clen, ls - d, S) >= 0) start_is_clusters = false;
    }
    // the row of a key: its position -- or, with a site index (np_genome_site_index_dev), the ORDINAL of the motif site at that position: a table of
    // n_sites rows instead of one row per base (both keys' positions are motif sites: a group's first and last)
    auto row_of = [&](int64_t pos) -> int64_t {
        if (!site_mask) return pos;
        const uint64_t w = site_mask[pos >> 6];
        return (int64_t)word_rank[pos >> 6] + __popcll(w & ((1ull << (pos & 63)) - 1ull));
    };
    int32_t* row;
    if (end_is_clusters) row = table + 6 * row_of(s);
    else if (start_is_clusters) row = table + 6 * row_of(e) + 3;
    else { atomicAdd(n_overflow, 1ull); return; }
    atomicAdd(row, 1);
    atomicAdd(row + 1, nm);
    if (meth) atomicAdd(row + 2, nm);
}

// profile_hmm_score_set's combination (src/hmm/nanopolish_profile_hmm.cpp:41-55): score = (+)_j (score_j - log n) with
// add_logs -> p7_FLogsum on (float) casts of the doubles (nanopolish_common.h:97-104), thread per set.
__global__ void __launch_bounds__(256) np_score_set_combine_kernel(int64_t n_sets, const int64_t* set_off, const int64_t* member_idx,
                                                                   const float* member_scores, const float* logsum, const double* log_n, float* out)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_sets) return;
    const int64_t b = set_off[q], n = set_off[q + 1] - b;
    if (n <= 0) { out[q] = NP_NEG_INF; return; }
    // glibc's log, restated (csrc/np_log.h) -- or, in host-constants mode, log(n) as the process's own libm computed it (n <= 64)
    const double pen = log_n && n <= 64 ? log_n[n] : np_log_glibc((double)(uint64_t)n);
    auto sc = [&](int64_t t) { return (double)member_scores[member_idx ? member_idx[b + t] : b + t]; };
    double score = sc(0) - pen;
    for (int64_t t = 1; t < n; ++t) {
        const double alt = sc(t) - pen;
        const float fa = (float)score, fb = (float)alt;
        const float mx = fa > fb ? fa : fb, mn = fa < fb ? fa : fb;
        score = (mn == NP_NEG_INF || (mx - mn) >= 15.7f) ? (double)mx : (double)(mx + logsum[(int)((mx - mn) * 1000.f)]);
    }
    out[q] = (float)score;
}
} // namespace

hipError_t np_launch_site_table(int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* n_motif,
                                const np_hmm_job_dev* jobs, const int64_t* read_base, double call_threshold, int64_t n_pos,
                                int32_t* table, hipStream_t s)
{
    if (n_groups <= 0) return hipSuccess;
    hipLaunchKernelGGL(np_site_table_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, n_groups, scores, first_site,
                       n_motif, jobs, read_base, call_threshold, n_pos, table);
    return hipGetLastError();
}

hipError_t np_launch_site_table_genome(int64_t n_groups, const float* scores, const int32_t* first_site, const int32_t* last_site, const int32_t* n_motif,
                                       const np_hmm_job_dev* jobs, const int64_t* read_base, const char* genome, const int64_t* contig_off, int n_contigs,
                                       int alphabet, int min_separation, double call_threshold, int64_t n_pos, int32_t* table,
                                       unsigned long long* n_overflow, const uint64_t* site_mask, const uint32_t* word_rank, hipStream_t s)
{
    if (n_groups <= 0) return hipSuccess;
    hipLaunchKernelGGL(np_site_table_genome_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, n_groups, scores, first_site, last_site,
                       n_motif, jobs, read_base, genome, contig_off, n_contigs, alphabet, min_separation, call_threshold, n_pos, table, n_overflow,
                       site_mask, word_rank);
    return hipGetLastError();
}

// chunk_scratch: uint32[ceil(n_words / NP_RANK_CHUNK)]
hipError_t np_launch_genome_site_index(const char* genome, const int64_t* contig_off, int n_contigs, int alphabet, int64_t n_pos, uint64_t* site_mask,
                                       uint32_t* word_rank, int64_t* n_sites, uint32_t* chunk_scratch, hipStream_t s)
{
    const int64_t n_words = (n_pos + 63) / 64, n_chunks = (n_words + NP_RANK_CHUNK - 1) / NP_RANK_CHUNK;
    if (n_pos <= 0) return hipMemsetAsync(n_sites, 0, sizeof(int64_t), s);
    hipLaunchKernelGGL(np_site_mask_kernel, dim3((unsigned)((n_words * 64 + 255) / 256)), dim3(256), 0, s, genome, contig_off, n_contigs, alphabet, n_pos, site_mask);
    hipLaunchKernelGGL(np_site_rank_local_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, site_mask, n_words, word_rank, chunk_scratch);
    hipLaunchKernelGGL(np_site_rank_chunks_kernel, dim3(1), dim3(1024), 0, s, chunk_scratch, n_chunks, n_sites);
    hipLaunchKernelGGL(np_site_rank_add_kernel, dim3((unsigned)((n_words + 1 + 255) / 256)), dim3(256), 0, s, word_rank, chunk_scratch, n_words, n_sites);
    return hipGetLastError();
}
int64_t np_site_rank_chunks(int64_t n_pos) { const int64_t n_words = (n_pos + 63) / 64; return (n_words + NP_RANK_CHUNK - 1) / NP_RANK_CHUNK; }

hipError_t np_launch_score_set_combine(int64_t n_sets, const int64_t* set_off, const int64_t* member_idx, const float* member_scores,
                                       const float* logsum, const double* log_n, float* out, hipStream_t s)
{
    if (n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(np_score_set_combine_kernel, dim3((unsigned)((n_sets + 255) / 256)), dim3(256), 0, s, n_sets, set_off, member_idx,
                       member_scores, logsum, log_n, out);
    return hipGetLastError();
}

hipError_t np_launch_selftest_div(uint64_t n_samples, uint64_t seed, unsigned long long* d_mismatches, hipStream_t s)
{
    const unsigned blocks = 4096;
    const uint64_t per_thread = (n_samples + (uint64_t)blocks * 256 - 1) / ((uint64_t)blocks * 256);
    hipLaunchKernelGGL(np_selftest_div_kernel, dim3(blocks), dim3(256), 0, s, per_thread, seed, d_mismatches);
    return hipGetLastError();
}

// shape: 0 = the default (NP_RC_W waves x NP_RC_R reads: 8 x 4), 1 / 2 = 16 x 2 and 12 x 3, kept for A/B runs (option "recal_shape");
// gpurun r05d, 40 000 reads, glue family: 5.997 ms for 8 x 4, 6.19 for 12 x 3, 6.45-6.49 for 16 x 2
#ifndef NP_RC_W
#define NP_RC_W 8
#endif
#ifndef NP_RC_R
#define NP_RC_R 4
#endif
template <int W, int R>
static hipError_t launch_recal(int n_reads, np_read_dev* reads, const float* event_mean, const uint16_t* ranks, const np_state_dev* model, int n_states,
                               const int32_t* n_pairs, const int32_t* map_start, int32_t* calibrated, const uint32_t* order, hipStream_t s)
{
    const int nb = (n_reads + W * R - 1) / (W * R);
    if (n_states <= NP_RC_STATES)
        hipLaunchKernelGGL((np_recalibrate_kernel<W, R, true>), dim3(nb), dim3(64 * W), 0, s, n_reads, reads, event_mean, ranks, model, n_states, n_pairs, map_start, calibrated, order);
    else
        hipLaunchKernelGGL((np_recalibrate_kernel<W, R, false>), dim3(nb), dim3(64 * W), 0, s, n_reads, reads, event_mean, ranks, model, n_states, n_pairs, map_start, calibrated, order);
    return hipGetLastError();
}
hipError_t np_launch_recalibrate(int n_reads, np_read_dev* reads, const float* event_mean, const uint16_t* ranks,
                                 const np_state_dev* model, int n_states, const int32_t* n_pairs, const int32_t* map_start,
                                 int32_t* calibrated, const uint32_t* order, int shape, hipStream_t s)
{
    if (n_reads <= 0) return hipSuccess;
    if (shape == 3 && n_states <= NP_RC_STATES) {          // round 6: half-wave chunks, sixteen waves x four reads, four waves per SIMD
        const int nb = (n_reads + 16 * 4 - 1) / (16 * 4);
        hipLaunchKernelGGL((np_recalibrate_half_kernel<16, 4>), dim3(nb), dim3(1024), 0, s, n_reads, reads, event_mean, ranks, model, n_states, n_pairs, map_start, calibrated, order);
        return hipGetLastError();
    }
    if (shape == 1) return launch_recal<16, 2>(n_reads, reads, event_mean, ranks, model, n_states, n_pairs, map_start, calibrated, order, s);
    if (shape == 2) return launch_recal<12, 3>(n_reads, reads, event_mean, ranks, model, n_states, n_pairs, map_start, calibrated, order, s);
    return launch_recal<NP_RC_W, NP_RC_R>(n_reads, reads, event_mean, ranks, model, n_states, n_pairs, map_start, calibrated, order, s);
}
