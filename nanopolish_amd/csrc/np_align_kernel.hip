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
//   * Prologue per read: the scaled Gaussian of every k-mer (fp64 math of squiggle_read.h:217-226, plus the
//     correctly rounded reciprocal of sigma) goes into a per-wave slab, 16 B per k-mer; the band loop then only
//     does fp32 emissions and the reference's fp64 candidate sums.
//   * The inner step is branch-free per lane: DP cells are computed by every lane and masked; the trim column
//     (k-mer -1) and the end-cell search only exist while the window touches k = -1 / k = K-1 and sit behind
//     wave-uniform branches.  Event means are prefetched one band ahead into ping-pong registers (the band loop is
//     unrolled by two so no register rotation -- and therefore no early s_waitcnt -- is needed).  A ring slot always
//     holds the parameters of its next k-mer (k+128) in `pend`, requested when the slot is re-targeted and consumed
//     128 right-moves later.
//   * Candidates are evaluated as the reference does: fp32 cell + fp64 transition constant + fp32 emission in
//     fp64, rounded to fp32, compared in fp32, later candidate wins ties (raw_loader.cpp:259-274).
//   * The trace is 2 bits per cell: each lane packs the codes of its two slots, 4 bits per band, and stores one dword per
//     8 bands (coalesced 256 B = 32 bytes per band, vs 100 bytes in the
//     reference); unfilled cells read back as FROM_D exactly like the reference's zero-initialised trace.
//   * Back-track: the walk state is scalar; the trace is pulled in 64-band chunks (lane i holds band hi-i, the
//     next chunk is prefetched), each step is two v_readlane + bit tests.  Pairs are collected 64 at a time in a
//     register pair and stored coalesced; their emissions are computed 64-wide at every flush and added to the QC
//     sum in walk order (the reference's summation order, raw_loader.cpp:338-341).
#include "np_kernels.h"

#define NP_ALIGN_BLOCK 256
#define NP_RING 128
#define NP_MARGIN 14   // (128 - 100) / 2

namespace {

__device__ __forceinline__ int ring_kmer(int slot, int llk)
{
    const int base = llk - NP_MARGIN;
    return base + ((slot - base) & (NP_RING - 1));
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// value of ring slot s (uniform) of a band held as (r0 = slots 0..63, r1 = slots 64..127)
__device__ __forceinline__ float ring_read(float r0, float r1, int s)
{
    const int a = __builtin_amdgcn_readlane(__builtin_bit_cast(int, r0), s & 63);
    const int b = __builtin_amdgcn_readlane(__builtin_bit_cast(int, r1), s & 63);
    return __builtin_bit_cast(float, (s & 64) ? b : a);      // both are SGPRs: a scalar select
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// parameter record of k-mer k from this wave's slab; the index is clamped (never a branch, never a default value:
// a conditional load would make hipcc wait for it on the spot) -- records of k outside [0,K) are never used
__device__ __forceinline__ float4 load_kp(const float4* __restrict__ kp, int k, int K) { return kp[clampi(k, 0, K - 1)]; }

__device__ __forceinline__ np_gauss as_gauss(const float4 v)
{
    np_gauss g; g.mean = v.x; g.stdv = v.y; g.cl = v.z; g.rinv = v.w;
    return g;
}

struct cell_out { float v; uint32_t from; };

// DP cell (raw_loader.cpp:240-289), computed unconditionally and masked by `valid`.
__device__ __forceinline__ cell_out dp_cell(bool valid, float x, const float4 gp, float up, float left, float diag,
                                            double lp_skip, double lp_stay, double lp_step)
{
    const float em = np_emission(x, as_gauss(gp));
    const float score_d = (float)((double)diag + lp_step + (double)em);
    const float score_u = (float)((double)up + lp_stay + (double)em);
    const float score_l = (float)((double)left + lp_skip);
    float mx = score_d; uint32_t from = 0;                                   // FROM_D
    mx = score_u > mx ? score_u : mx; from = (mx == score_u) ? 1u : from;   // FROM_U
    mx = score_l > mx ? score_l : mx; from = (mx == score_l) ? 2u : from;   // FROM_L
    cell_out o;
    o.v = valid ? mx : NP_NEG_INF;
    o.from = valid ? from : 0u;
    return o;
}

// Everything the fill carries from band to band.
struct fill_t {
    int llk;                // band_lower_left[b].kmer_idx (wave-uniform)
    int k0, k1;             // k-mer mapped to this lane's two ring slots
    float4 g0, g1;          // their scaled Gaussians (mean, stdv, cl, 1/stdv)
    float4 n0, n1;          // the records of k0+128 / k1+128, requested when the slot was last re-targeted
    float p0, p1;           // band b-1
    float d0, d1;           // band b-2 rotated by one slot
    float best; int best_e; // end-cell search (:309-324), tracked by the owner of k-mer K-1
    uint32_t tacc;          // trace codes of the current 8-band group (4 bits per band)
};

struct read_t {
    int E, K, lane, end_slot;
    const float* __restrict__ ev;
    const float4* __restrict__ kp;
    uint32_t* __restrict__ trace32;
    double lp_skip, lp_stay, lp_step, lp_trim;
};

// One band.  (xc0, xc1): event means of this band's cells (loaded during the previous band);
// (xl0, xl1): receive the loads for the next band.
__device__ __forceinline__ void band_step(fill_t& F, const read_t& R, const int b, const float xc0, const float xc1,
                                          float& xl0, float& xl1)
{
    const int lane = R.lane, E = R.E, K = R.K;
    if (b >= 2) {
        // Suzuki's rule on band b-1 (:179-195)
        const float ll = ring_read(F.p0, F.p1, F.llk & (NP_RING - 1));
        const float ur = ring_read(F.p0, F.p1, (F.llk + NP_ALN_BANDWIDTH - 1) & (NP_RING - 1));
        const bool ll_ob = ll == NP_NEG_INF, ur_ob = ur == NP_NEG_INF;
        const bool right = (ll_ob && ur_ob) ? ((b & 1) == 1) : (ll < ur);
        if (right) {
            F.llk += 1;
            // the slot that fell 14 behind the window takes its next k-mer (k+128), whose record was requested the
            // last time the slot moved, and requests the one after that
            // (branch-free on purpose: every lane re-requests its `next` record on every right move -- an L1 hit for all
            //  but the re-targeted slot -- so that no load sits inside a divergent branch, where hipcc would wait for it
            //  immediately; the record consumed here was requested at least one band ago)
            // Exactly one ring slot falls out per right move: slot (llk - 15) mod 128, i.e. one lane of ONE of the two
            // slot registers -- which one is wave-uniform, so only that register is touched.
            if (((F.llk - NP_MARGIN - 1) & 64) == 0) {
                const bool t0 = F.k0 < F.llk - NP_MARGIN;
                F.g0.x = t0 ? F.n0.x : F.g0.x; F.g0.y = t0 ? F.n0.y : F.g0.y; F.g0.z = t0 ? F.n0.z : F.g0.z; F.g0.w = t0 ? F.n0.w : F.g0.w;
                F.k0 += t0 ? NP_RING : 0;
                F.n0 = load_kp(R.kp, F.k0 + NP_RING, K);
            } else {
                const bool t1 = F.k1 < F.llk - NP_MARGIN;
                F.g1.x = t1 ? F.n1.x : F.g1.x; F.g1.y = t1 ? F.n1.y : F.g1.y; F.g1.z = t1 ? F.n1.z : F.g1.z; F.g1.w = t1 ? F.n1.w : F.g1.w;
                F.k1 += t1 ? NP_RING : 0;
                F.n1 = load_kp(R.kp, F.k1 + NP_RING, K);
            }
        }
    }
    const int llk = F.llk;
    // left sources: band b-1 rotated by one slot
    const float r0 = np_wave_ror1(F.p0), r1 = np_wave_ror1(F.p1);
    const float l0 = lane == 0 ? r1 : r0;
    const float l1 = lane == 0 ? r0 : r1;

    const int klo = llk > 0 ? llk : 0;
    const int khi = (llk + NP_ALN_BANDWIDTH - 1) < (K - 1) ? (llk + NP_ALN_BANDWIDTH - 1) : (K - 1);
    const int e0 = b - 2 - F.k0, e1 = b - 2 - F.k1;
    const bool v0 = F.k0 >= klo && F.k0 <= khi && (unsigned)e0 < (unsigned)E;
    const bool v1 = F.k1 >= klo && F.k1 <= khi && (unsigned)e1 < (unsigned)E;
    // prefetch the next band's event means (same k-mer, next event), clamped: always a plain load
    xl0 = R.ev[(uint32_t)clampi(e0 + 1, 0, E - 1)];
    xl1 = R.ev[(uint32_t)clampi(e1 + 1, 0, E - 1)];

    cell_out c0 = dp_cell(v0, xc0, F.g0, F.p0, l0, F.d0, R.lp_skip, R.lp_stay, R.lp_step);
    cell_out c1 = dp_cell(v1, xc1, F.g1, F.p1, l1, F.d1, R.lp_skip, R.lp_stay, R.lp_step);

    if (llk <= -1) {
        // the window still contains k-mer -1: start cell of band 0 (:152-157) and the trim column (:216-225).
        // k = -1 lives in ring slot 127 (lane 63, second register); its event is b - 1.
        const int et = b - 1;
        if (lane == 63 && F.k1 == -1) {
            if (et == -1) { c1.v = 0.0f; c1.from = 0u; }
            else if (et >= 0 && et < E) { c1.v = (float)(R.lp_trim * (double)(et + 1)); c1.from = 1u; }
            else { c1.v = NP_NEG_INF; c1.from = 0u; }
        }
    }

    // packed trace: every lane keeps the 2-bit codes of its two slots, 4 bits per band, and stores one dword per
    // 8 bands (64 lanes x 4 B = 256 B coalesced = 32 B/band).  The back-track reads the word back into the SAME lane.
    F.tacc |= (c0.from | (c1.from << 2)) << ((b & 7) * 4);
    if ((b & 7) == 7) { R.trace32[(size_t)(b >> 3) * 64 + lane] = F.tacc; F.tacc = 0u; }

    if (khi == K - 1 && llk <= K - 1) {
        // end search: cell (e, K-1) while it is inside the window, any e in [0,E) (:309-324)
        const bool mine0 = (R.end_slot < 64) && lane == R.end_slot;
        const bool mine1 = (R.end_slot >= 64) && lane == R.end_slot - 64;
        if (mine0 || mine1) {
            const int k = mine0 ? F.k0 : F.k1;
            const int e = mine0 ? e0 : e1;
            const float v = mine0 ? c0.v : c1.v;
            if (k == K - 1 && e >= 0 && e < E) {
                const float sc = (float)((double)v + (double)(E - e) * R.lp_trim);
                if (sc > F.best) { F.best = sc; F.best_e = e; }
            }
        }
    }
    F.d0 = l0; F.d1 = l1;
    F.p0 = c0.v; F.p1 = c1.v;
}

__global__ void __launch_bounds__(NP_ALIGN_BLOCK, 7) np_event_align_kernel(np_align_args a)
{
    const int lane = threadIdx.x & 63;
    const int wave_slot = blockIdx.x * (NP_ALIGN_BLOCK / 64) + (threadIdx.x >> 6);
    uint64_t* __restrict__ trace = a.trace + (size_t)wave_slot * a.trace_stride;
    float4* __restrict__ kp = a.kparams + (size_t)wave_slot * a.kp_stride;     // per-wave slab of scaled k-mer parameters

    for (;;) {
        // Ticket grab without an `if (lane == 0)`: hipcc threads a lane-0 branch at the loop top together with a
        // lane-0 branch at the loop bottom and then runs the (convergent) readfirstlane on a partial wave.
        const int ri = __builtin_amdgcn_readfirstlane((int)atomicAdd(a.counter, lane == 0 ? 1u : 0u));
        if (ri >= a.n_reads) break;

        const np_read_dev* rd = a.reads + ri;
        const int E = (int)rd->n_events, K = (int)rd->n_kmers;
        const float* __restrict__ ev = a.event_mean + rd->event_off;
        const uint16_t* __restrict__ rk = a.ranks + rd->rank_off;
        const np_state_dev* __restrict__ model = a.model;
        const double scale = rd->scale, shift = rd->shift, var = rd->var, log_var = rd->log_var;
        const int n_bands = E + K + 2;
        const int64_t pbase = a.pair_off[ri];
        const int cap = (int)(a.pair_off[ri + 1] - pbase);
        np_pair* __restrict__ pairs = a.pairs + pbase;

        const bool ok = !(E <= 0 || K <= 0 || (uint64_t)((n_bands + 7) >> 3) * 32 > a.trace_stride || (uint64_t)K > a.kp_stride || cap < E + K + 2);
        int n_out = 0, max_gap = 0, last_k = -1;
        double sum_emission = 0.0;
        if (ok) {
            // ---------------- prologue: per-k-mer scaled Gaussians ----------------
            for (int k = lane; k < K; k += 64) {
                const uint32_t r = rk[k];
                const np_gauss g = np_make_gauss(model[r].level_mean, model[r].level_stdv, model[r].level_log_stdv, scale, shift, var, log_var);
                kp[k] = make_float4(g.mean, g.stdv, g.cl, g.rinv);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);

            // ---------------- fill ----------------
            read_t R;
            R.E = E; R.K = K; R.lane = lane; R.end_slot = (K - 1) & (NP_RING - 1);
            R.ev = ev; R.kp = kp; R.trace32 = (uint32_t*)trace;
            R.lp_skip = rd->lp_skip; R.lp_stay = rd->lp_stay; R.lp_step = rd->lp_step; R.lp_trim = rd->lp_trim;
            fill_t F;
            F.llk = -1 - NP_ALN_BANDWIDTH / 2;              // band_lower_left[0].kmer_idx, raw_loader.cpp:150-151
            F.k0 = ring_kmer(lane, F.llk); F.k1 = ring_kmer(lane + 64, F.llk);
            F.g0 = load_kp(kp, F.k0, K); F.g1 = load_kp(kp, F.k1, K);
            F.n0 = load_kp(kp, F.k0 + NP_RING, K); F.n1 = load_kp(kp, F.k1 + NP_RING, K);
            F.p0 = F.p1 = F.d0 = F.d1 = NP_NEG_INF;
            F.best = NP_NEG_INF; F.best_e = 0; F.tacc = 0u;
            float xa0 = 0.0f, xa1 = 0.0f, xb0 = 0.0f, xb1 = 0.0f;      // ping-pong event-mean registers
            int b = 0;
            for (; b + 1 < n_bands; b += 2) {
                band_step(F, R, b, xa0, xa1, xb0, xb1);
                band_step(F, R, b + 1, xb0, xb1, xa0, xa1);
            }
            if (b < n_bands) band_step(F, R, b, xa0, xa1, xb0, xb1);
            if ((n_bands & 7) != 0) R.trace32[(size_t)((n_bands - 1) >> 3) * 64 + lane] = F.tacc;      // last, partial group

            // ---------------- backtrack (:326-361) + QC sums (:338-341) ----------------
            const int owner = R.end_slot & 63;
            const float best_u = readlane_f(F.best, owner);
            int curr_e = __builtin_amdgcn_readlane(F.best_e, owner);
            int curr_k = K - 1;

            // the trace was written by lanes 0..3 of this wave: complete the stores before other lanes read them back
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);

            if (best_u != NP_NEG_INF) {
                // trace words: `cw` holds the 8-band group `cg` of the current cell for this lane's two slots, `nw` the
                // group below it (prefetched: the walk only moves down, by one or two bands per step)
                const uint32_t* __restrict__ t32 = (const uint32_t*)trace;
                int cg = (curr_e + curr_k + 2) >> 3;
                uint32_t cw = t32[(size_t)cg * 64 + lane];
                uint32_t nw = cg > 0 ? t32[(size_t)(cg - 1) * 64 + lane] : 0u;
                int pk = 0, pe = 0;                    // pair buffer: lane j holds pair number (n_out & ~63) + j
                int curr_gap = 0;
                while (curr_k >= 0 && curr_e >= 0) {
                    const int band = curr_e + curr_k + 2;
                    if ((band >> 3) != cg) {
                        cg -= 1; cw = nw;
                        nw = cg > 0 ? t32[(size_t)(cg - 1) * 64 + lane] : 0u;
                    }
                    // record the pair
                    const int j = n_out & 63;
                    pk = lane == j ? curr_k : pk;
                    pe = lane == j ? curr_e : pe;
                    last_k = curr_k;
                    n_out++;
                    // 2-bit trace code of cell (curr_e, curr_k): the lane that owns slot k mod 128, nibble (band mod 8)
                    const int slot = curr_k & (NP_RING - 1);
                    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cw, slot & 63);
                    const uint32_t from = (w >> ((band & 7) * 4 + ((slot >> 6) << 1))) & 3u;
                    if (from == 0u) { curr_k -= 1; curr_e -= 1; curr_gap = 0; }
                    else if (from == 1u) { curr_e -= 1; curr_gap = 0; }
                    else { curr_k -= 1; curr_gap += 1; max_gap = curr_gap > max_gap ? curr_gap : max_gap; }

                    const bool done = !(curr_k >= 0 && curr_e >= 0);
                    if (j == 63 || done) {
                        // flush: store up to 64 pairs (descending addresses) and add their emissions in walk order
                        const int cnt = j + 1, first = n_out - cnt;
                        float em = 0.0f;
                        if (lane < cnt) {
                            np_pair p; p.ref_pos = pk; p.read_pos = pe;
                            pairs[cap - 1 - (first + lane)] = p;
                            em = np_emission(ev[pe], as_gauss(load_kp(kp, pk, K)));
                        }
                        for (int q = 0; q < cnt; ++q) sum_emission += (double)readlane_f(em, q);
                    }
                }
            }
        }
        {
            // QC (:365-372); out.back() is always k-mer K-1, so `spanned` reduces to "the walk ended on k-mer 0"
            bool failed = true;
            if (n_out > 0) {
                const double avg_log_emission = sum_emission / (double)n_out;
                failed = avg_log_emission < a.min_average_log_emission || last_k != 0 || max_gap > a.max_gap_threshold;
            }
            // two lanes, two arrays (deliberately not an `if (lane == 0)`, see the ticket grab above)
            int32_t* dst = lane == 0 ? a.pair_begin + ri : a.n_pairs + ri;
            const int32_t val = lane == 0 ? cap - n_out : (failed ? 0 : n_out);
            if (lane < 2) *dst = val;
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
