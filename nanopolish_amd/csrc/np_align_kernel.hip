// np_align_kernel.hip -- adaptive_banded_simple_event_align (src/nanopolish_raw_loader.cpp:77-379) for gfx950.
//
// One wavefront per read, ~E+K+2 sequential band steps.  Design (DESIGN.md section "Kernel A"):
//   * The 100-cell band lives in registers, anchored by k-mer index instead of by band offset: cell (event e,
//     k-mer k) of band b = e+k+2 sits in ring slot (k mod 128); lane l owns slots l and l+64.  With that
//     anchoring the three DP sources never depend on the band's move history:
//         up   = band b-1, same slot        left = band b-1, slot-1        diag = band b-2, slot-1
//     so one wave-rotate (2 DPP wave_ror:1 + 2 selects) of the previous band per step serves `left`, and the
//     rotate kept from the step before serves `diag`.  A slot whose k-mer is outside the band window holds -inf,
//     which is exactly the reference's is_offset_valid(...) ? BAND_ARRAY(...) : -INFINITY.
//   * Suzuki's move rule reads the band's first and last cell (ll, ur) with v_readlane; all band geometry is
//     wave-uniform scalar state.
//   * Candidates are evaluated as the reference does: fp32 cell + fp64 transition constant + fp32 emission in
//     fp64, rounded to fp32, compared in fp32, later candidate wins ties (raw_loader.cpp:259-274).
//   * The trace is 2 bits per cell, packed with 4 ballots per band into 32 bytes (vs 100 bytes in the
//     reference); unfilled cells read back as FROM_D exactly like the reference's zero-initialised trace.
//   * Backtrack + QC run in the same wave right after the fill (uniform walk over the packed trace).
#include "np_kernels.h"

#define NP_ALIGN_BLOCK 256
#define NP_RING 128
#define NP_MARGIN 14   // (128 - 100) / 2

namespace {

struct slot_t {
    int k;          // k-mer index currently mapped to this ring slot
    np_gauss g;     // its scaled Gaussian (valid when 0 <= k < K)
};

__device__ __forceinline__ int ring_kmer(int slot, int llk)
{
    const int base = llk - NP_MARGIN;
    return base + ((slot - base) & (NP_RING - 1));
}

__device__ __forceinline__ np_gauss load_kmer(const np_align_args& a, const uint16_t* rk, int k, int K,
                                              double scale, double shift, double var, double log_var)
{
    const uint32_t rank = (k >= 0 && k < K) ? rk[k] : 0u;
    return np_scale_state(a.model, rank, scale, shift, var, log_var);
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// value of ring slot s (uniform) of a band held as (r0 = slots 0..63, r1 = slots 64..127)
__device__ __forceinline__ float ring_read(float r0, float r1, int s)
{
    const float a = readlane_f(r0, s & 63), b = readlane_f(r1, s & 63);
    return (s & 64) ? b : a;
}

struct cell_out { float v; uint32_t from; };

__device__ __forceinline__ cell_out align_cell(int k, int b, int llk, int E, int K, float x, const np_gauss& g,
                                               float up, float left, float diag,
                                               double lp_skip, double lp_stay, double lp_step, double lp_trim)
{
    cell_out o; o.v = NP_NEG_INF; o.from = 0;
    const int e = b - 2 - k;
    const bool inwin = k >= llk && k <= llk + (NP_ALN_BANDWIDTH - 1);
    if (inwin) {
        if (k == -1) {
            // start cell of band 0 (raw_loader.cpp:152-157) and the trim column (:216-225)
            if (e == -1) { o.v = 0.0f; }
            else if (e >= 0 && e < E) { o.v = (float)(lp_trim * (double)(e + 1)); o.from = 1; }
        } else if (k >= 0 && k < K && e >= 0 && e < E) {
            const float em = np_emission(x, g);
            const float score_d = (float)((double)diag + lp_step + (double)em);
            const float score_u = (float)((double)up + lp_stay + (double)em);
            const float score_l = (float)((double)left + lp_skip);
            float mx = score_d; uint32_t from = 0;                     // FROM_D
            mx = score_u > mx ? score_u : mx; from = (mx == score_u) ? 1u : from;   // FROM_U
            mx = score_l > mx ? score_l : mx; from = (mx == score_l) ? 2u : from;   // FROM_L
            o.v = mx; o.from = from;
        }
    }
    return o;
}

__global__ void __launch_bounds__(NP_ALIGN_BLOCK) np_event_align_kernel(np_align_args a)
{
    const int lane = threadIdx.x & 63;
    const int wave_slot = blockIdx.x * (NP_ALIGN_BLOCK / 64) + (threadIdx.x >> 6);
    uint64_t* __restrict__ trace = a.trace + (size_t)wave_slot * a.trace_stride;
    if (a.dbg && wave_slot == 0 && lane == 0) a.dbg[8] = 10;

    for (;;) {
        // Ticket grab without an `if (lane == 0)`: hipcc threads a lane-0 branch at the loop top together with a
        // lane-0 branch at the loop bottom and then runs the (convergent) readfirstlane on a partial wave.
        const int ri = __builtin_amdgcn_readfirstlane((int)atomicAdd(a.counter, lane == 0 ? 1u : 0u));
        if (a.dbg && wave_slot == 0 && lane == 0) { a.dbg[8] = 11; a.dbg[9] = (uint32_t)ri; }
        if (ri >= a.n_reads) break;

        const np_read_dev* rd = a.reads + ri;
        const int E = (int)rd->n_events, K = (int)rd->n_kmers;
        const float* __restrict__ ev = a.event_mean + rd->event_off;
        const uint16_t* __restrict__ rk = a.ranks + rd->rank_off;
        const double scale = rd->scale, shift = rd->shift, var = rd->var, log_var = rd->log_var;
        const double lp_skip = rd->lp_skip, lp_stay = rd->lp_stay, lp_step = rd->lp_step, lp_trim = rd->lp_trim;
        const int n_bands = E + K + 2;
        const int64_t pbase = a.pair_off[ri];
        const int cap = (int)(a.pair_off[ri + 1] - pbase);

        if (E <= 0 || K <= 0 || (uint64_t)n_bands * 4 > a.trace_stride || cap < E + K + 2) {
            if (lane == 0) { a.pair_begin[ri] = cap; a.n_pairs[ri] = 0; }
            continue;
        }

        if (a.dbg && ri == 0 && lane == 0) { a.dbg[8] = 12; a.dbg[10] = (uint32_t)E; a.dbg[11] = (uint32_t)K; }
        // ---------------- fill ----------------
        int llk = -1 - NP_ALN_BANDWIDTH / 2;            // band_lower_left[0].kmer_idx, raw_loader.cpp:150-151
        slot_t s0, s1;
        s0.k = ring_kmer(lane, llk);      s0.g = load_kmer(a, rk, s0.k, K, scale, shift, var, log_var);
        s1.k = ring_kmer(lane + 64, llk); s1.g = load_kmer(a, rk, s1.k, K, scale, shift, var, log_var);
        if (a.dbg && ri == 0 && lane == 0) a.dbg[8] = 13;
        float p0 = NP_NEG_INF, p1 = NP_NEG_INF;   // band b-1
        float d0 = NP_NEG_INF, d1 = NP_NEG_INF;   // band b-2 rotated by one slot
        float best = NP_NEG_INF; int best_e = 0;  // end-cell search (:309-324), tracked by the owner of k-mer K-1
        const int end_slot = (K - 1) & (NP_RING - 1);

        for (int b = 0; b < n_bands; ++b) {
            if (a.dbg && ri == 0 && lane == 0) { a.dbg[0] = 1; a.dbg[1] = (uint32_t)b; a.dbg[3] = (uint32_t)n_bands; }
            if (b >= 2) {
                // Suzuki's rule on band b-1 (:179-195)
                const float ll = ring_read(p0, p1, llk & (NP_RING - 1));
                const float ur = ring_read(p0, p1, (llk + NP_ALN_BANDWIDTH - 1) & (NP_RING - 1));
                const bool ll_ob = ll == NP_NEG_INF, ur_ob = ur == NP_NEG_INF;
                const bool right = (ll_ob && ur_ob) ? ((b & 1) == 1) : (ll < ur);
                if (right) {
                    llk += 1;
                    // ring slots that fell 14 behind the window are re-targeted 128 k-mers ahead
                    if (s0.k < llk - NP_MARGIN) { s0.k += NP_RING; s0.g = load_kmer(a, rk, s0.k, K, scale, shift, var, log_var); }
                    if (s1.k < llk - NP_MARGIN) { s1.k += NP_RING; s1.g = load_kmer(a, rk, s1.k, K, scale, shift, var, log_var); }
                }
            }
            // left sources: band b-1 rotated by one slot
            const float r0 = np_wave_ror1(p0), r1 = np_wave_ror1(p1);
            const float l0 = lane == 0 ? r1 : r0;
            const float l1 = lane == 0 ? r0 : r1;

            const int e0 = b - 2 - s0.k, e1 = b - 2 - s1.k;
            const float x0 = (e0 >= 0 && e0 < E) ? ev[e0] : 0.0f;
            const float x1 = (e1 >= 0 && e1 < E) ? ev[e1] : 0.0f;
            const cell_out c0 = align_cell(s0.k, b, llk, E, K, x0, s0.g, p0, l0, d0, lp_skip, lp_stay, lp_step, lp_trim);
            const cell_out c1 = align_cell(s1.k, b, llk, E, K, x1, s1.g, p1, l1, d1, lp_skip, lp_stay, lp_step, lp_trim);

            // packed trace: 4 x 64-bit ballots per band
            const uint64_t m00 = __ballot(c0.from & 1u), m01 = __ballot(c0.from >> 1);
            const uint64_t m10 = __ballot(c1.from & 1u), m11 = __ballot(c1.from >> 1);
            if (lane < 4) {
                const uint64_t w = lane == 0 ? m00 : lane == 1 ? m01 : lane == 2 ? m10 : m11;
                trace[(size_t)b * 4 + lane] = w;
            }

            // end search: cell (e, K-1), in-window, any e in [0,E)
            {
                const bool mine0 = (end_slot < 64) && lane == end_slot;
                const bool mine1 = (end_slot >= 64) && lane == end_slot - 64;
                if (mine0 || mine1) {
                    const int k = mine0 ? s0.k : s1.k;
                    const int e = mine0 ? e0 : e1;
                    const float v = mine0 ? c0.v : c1.v;
                    if (k == K - 1 && e >= 0 && e < E && k >= llk && k <= llk + NP_ALN_BANDWIDTH - 1) {
                        const float sc = (float)((double)v + (double)(E - e) * lp_trim);
                        if (sc > best) { best = sc; best_e = e; }
                    }
                }
            }
            d0 = l0; d1 = l1;
            p0 = c0.v; p1 = c1.v;
        }

        // ---------------- backtrack (:326-361) + QC (:365-372) ----------------
        const int owner = end_slot & 63;
        const float best_u = readlane_f(best, owner);
        int curr_e = __builtin_amdgcn_readlane(best_e, owner);
        int curr_k = K - 1;
        np_pair* __restrict__ pairs = a.pairs + pbase;

        // make the trace written by lanes 0..3 visible to every lane of this wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);

        int n_out = 0, curr_gap = 0, max_gap = 0;
        double sum_emission = 0.0;
        if (best_u != NP_NEG_INF) {
            while (curr_k >= 0 && curr_e >= 0) {
                if (a.dbg && ri == 0 && lane == 0) { a.dbg[0] = 2; a.dbg[2] = (uint32_t)n_out; a.dbg[4] = (uint32_t)curr_k; a.dbg[5] = (uint32_t)curr_e; }
                if (lane == 0) { np_pair p; p.ref_pos = curr_k; p.read_pos = curr_e; pairs[cap - 1 - n_out] = p; }
                n_out++;
                const np_gauss g = load_kmer(a, rk, curr_k, K, scale, shift, var, log_var);
                sum_emission += (double)np_emission(ev[curr_e], g);
                const int band = curr_e + curr_k + 2;
                const int slot = curr_k & (NP_RING - 1);
                const uint64_t* w = trace + (size_t)band * 4 + ((slot >> 6) << 1);
                const uint64_t b0 = __builtin_nontemporal_load(w), b1 = __builtin_nontemporal_load(w + 1);
                const uint32_t from = (uint32_t)((b0 >> (slot & 63)) & 1u) | ((uint32_t)((b1 >> (slot & 63)) & 1u) << 1);
                if (from == 0u) { curr_k -= 1; curr_e -= 1; curr_gap = 0; }
                else if (from == 1u) { curr_e -= 1; curr_gap = 0; }
                else { curr_k -= 1; curr_gap += 1; max_gap = curr_gap > max_gap ? curr_gap : max_gap; }
            }
        }
        if (a.dbg && ri == 0 && lane == 0) a.dbg[0] = 3;
        if (lane == 0) {
            bool failed = true;
            if (n_out > 0) {
                const double avg_log_emission = sum_emission / (double)n_out;
                const bool spanned = pairs[cap - n_out].ref_pos == 0;     // out.back() is always k-mer K-1
                failed = avg_log_emission < a.min_average_log_emission || !spanned || max_gap > a.max_gap_threshold;
            }
            a.pair_begin[ri] = cap - n_out;
            a.n_pairs[ri] = failed ? 0 : n_out;
        }
    }
}

} // namespace

int np_align_block_threads(void) { return NP_ALIGN_BLOCK; }

hipError_t np_launch_event_align(const np_align_args& a, int n_blocks, hipStream_t s)
{
    hipLaunchKernelGGL(np_event_align_kernel, dim3(n_blocks), dim3(NP_ALIGN_BLOCK), 0, s, a);
    return hipGetLastError();
}
