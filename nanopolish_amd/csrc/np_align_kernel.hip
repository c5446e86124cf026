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
//   * The inner step is branch-free per lane: DP cells are computed by every lane and masked.  The band loop runs in
//     three phases: a generic step (band geometry, trim column k = -1, end-cell search) for the first and last ~300
//     bands, and a FAST step for the middle of the read, where every cell of the 100-wide window exists, so validity
//     is a pair of wave-uniform lane masks that only change on a right move.
//   * Event means are prefetched one band ahead straight into the loop-carried register (the load is issued after
//     the emission has consumed the old value), through a range-checked buffer descriptor: out-of-range events read
//     as 0 and only feed masked cells, so there is no clamp.  A ring slot always holds the parameters of its next
//     k-mer (k+128) in `n0/n1`, requested when the slot is re-targeted and consumed 128 right-moves later; the
//     re-target itself is five v_mov under a one-lane exec mask.
//   * Issue-cycle budget (tools/valu_rates.hip, measured on MI355X): fp32 add/fma/mov/int add issue in ~2 cycles per
//     wave, fp64 ops, conversions, v_cmp, v_cndmask (VOP3), DPP, v_readlane and 3-operand integer ops in ~4, and the
//     scalar unit also sustains one instruction per ~4 cycles per SIMD -- so scalar band bookkeeping is as expensive
//     as vector work and is kept out of the FAST step.  Back-to-back VOP2 v_cndmask (implicit VCC) issue at ~20.
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

__device__ __forceinline__ double uniform_f64(double v)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
template <class T> __device__ __forceinline__ T* uniform_ptr(T* p)
{
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return (T*)(((uint64_t)hi << 32) | lo);
}
// Everything the fill carries from band to band.
struct fill_t {
    int llk;                // band_lower_left[b].kmer_idx (wave-uniform)
    int kb0, kb1;           // 4 * (k-mer mapped to this lane's two ring slots)
    float g0m, g0s, g0c, g0r;   // scaled Gaussian of slot l: mean, stdv, log-constant, 1/stdv (scalars, so that a
    float g1m, g1s, g1c, g1r;   // re-target can update them in place) -- and of slot l + 64
    float4 n0, n1;          // the records of k0+128 / k1+128, requested when the slot was last re-targeted
    float p0, p1;           // band b-1
    float d0, d1;           // band b-2 rotated by one slot
    float best; int best_e; // end-cell search (:309-324), tracked by the owner of k-mer K-1
    uint32_t tacc;          // trace codes of the last (up to) 8 bands, 4 bits per band, newest in the top nibble
    uint64_t vm0, vm1;      // FAST phase: lanes whose slot (l, l + 64) is inside the window (wave-uniform lane masks)
};

struct read_t {
    int E, K, lane, end_slot;
    __amdgpu_buffer_rsrc_t ev;      // event means of the read, E * 4 bytes
    __amdgpu_buffer_rsrc_t kp;      // this wave's k-mer parameter slab, K * 16 bytes
    uint32_t* __restrict__ trace32;
    double lp_skip, lp_stay, lp_step, lp_trim;
};

// bit pattern of ring slot s (uniform) of a band held as (r0 = slots 0..63, r1 = slots 64..127); all scalar
__device__ __forceinline__ int ring_read_bits(float r0, float r1, int s)
{
    int a = __builtin_amdgcn_readlane(__builtin_bit_cast(int, r0), s & 63);
    int b = __builtin_amdgcn_readlane(__builtin_bit_cast(int, r1), s & 63);
    asm("" : "+s"(a)); asm("" : "+s"(b));
    return (s & 64) ? b : a;
}
// keeps a wave-uniform value in a scalar register, so that what is computed from it is selected onto the scalar unit
__device__ __forceinline__ int pin_s(int x) { asm("" : "+s"(x)); return x; }
// wave_ror:1 where every lane has a source lane, so the destination's previous content needs no initialisation
__device__ __forceinline__ float wave_ror1_all(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false));
}
// t where the lane's bit of the (wave-uniform) mask m is set, else f: one v_cndmask_b32_e64 on an SGPR pair
__device__ __forceinline__ float sel_mask(uint64_t m, float t, float f)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}
// lanes of slot register `half` (0: slots 0..63, 1: slots 64..127) whose slot is inside the window [llk, llk+99]
__device__ __forceinline__ uint64_t window_mask(int lane, int half, int llk)
{
    return __builtin_amdgcn_ballot_w64((uint32_t)((lane + 64 * half - llk) & (NP_RING - 1)) < (uint32_t)NP_ALN_BANDWIDTH);
}

// The lane selected by m0 (slot register 0) or m1 (slot register 1) -- one of the two masks is empty -- takes over its
// pending record and moves on by 128 k-mers.  Plain v_mov under a one-lane exec mask: straight-line code that updates
// the registers in place (selects cost twice the issue cycles, and a branch per register costs the compiler's copies).
__device__ __forceinline__ void retarget(fill_t& F, uint64_t m0, uint64_t m1)
{
    uint64_t save;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[m0]\n\t"
                 "v_mov_b32 %[a0], %[x0]\n\tv_mov_b32 %[a1], %[x1]\n\tv_mov_b32 %[a2], %[x2]\n\tv_mov_b32 %[a3], %[x3]\n\t"
                 "v_add_u32 %[k0], 0x200, %[k0]\n\t"
                 "s_mov_b64 exec, %[m1]\n\t"
                 "v_mov_b32 %[b0], %[y0]\n\tv_mov_b32 %[b1], %[y1]\n\tv_mov_b32 %[b2], %[y2]\n\tv_mov_b32 %[b3], %[y3]\n\t"
                 "v_add_u32 %[k1], 0x200, %[k1]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(save), [a0] "+v"(F.g0m), [a1] "+v"(F.g0s), [a2] "+v"(F.g0c), [a3] "+v"(F.g0r), [k0] "+v"(F.kb0),
                   [b0] "+v"(F.g1m), [b1] "+v"(F.g1s), [b2] "+v"(F.g1c), [b3] "+v"(F.g1r), [k1] "+v"(F.kb1)
                 : [m0] "s"(m0), [m1] "s"(m1), [x0] "v"(F.n0.x), [x1] "v"(F.n0.y), [x2] "v"(F.n0.z), [x3] "v"(F.n0.w),
                   [y0] "v"(F.n1.x), [y1] "v"(F.n1.y), [y2] "v"(F.n1.z), [y3] "v"(F.n1.w));
}

// One band.  (x0, x1): event means of this band's two cells on entry (loaded during the previous band), of the next
// band's on exit.  TRIM: the window may still contain k-mer -1.  END: the window may contain k-mer K-1.
// FAST: the middle of the read -- every cell of the window [llk, llk+99] exists (0 <= llk, llk+99 < K-1, and the events
// of the window's first and last k-mer are inside [0, E)), so a slot is valid iff it is inside the window, which only
// changes on a right move: the lane masks F.vm0/F.vm1 replace the per-band geometry.
template <bool TRIM, bool END, bool FAST>
__device__ __forceinline__ void band_step(fill_t& F, const read_t& R, const int b, float& x0, float& x1)
{
    const int lane = R.lane, E = R.E, K = R.K;
    const int llk = F.llk;
    // left sources: band b-1 rotated by one slot
    const float r0 = wave_ror1_all(F.p0), r1 = wave_ror1_all(F.p1);
    const float l0 = lane == 0 ? r1 : r0;
    const float l1 = lane == 0 ? r0 : r1;

    // A slot holds a cell of this band iff its k-mer is inside the window and its event e = b-2-k exists, i.e. iff
    // e is in [max(0, b-2-khi), min(E-1, b-2-klo)]: one unsigned range test on 4*(e+1), the byte offset of the NEXT
    // band's event mean, which the prefetch needs anyway.
    const int klo = llk > 0 ? llk : 0;
    const int khi = (llk + NP_ALN_BANDWIDTH - 1) < (K - 1) ? (llk + NP_ALN_BANDWIDTH - 1) : (K - 1);
    const int off0 = 4 * (b - 1) - F.kb0, off1 = 4 * (b - 1) - F.kb1;
    bool v0 = false, v1 = false;
    if (!FAST) {
        const int elo = (b - 2 - khi) > 0 ? (b - 2 - khi) : 0;
        const int ehi = (b - 2 - klo) < (E - 1) ? (b - 2 - klo) : (E - 1);
        const int cnt = ehi - elo + 1 > 0 ? ehi - elo + 1 : 0;
        const int tb = pin_s(4 * (b - 1) - 4 * (elo + 1));       // one scalar, so the test is one subtract + one compare
        v0 = (uint32_t)(tb - F.kb0) < (uint32_t)(4 * cnt);
        v1 = (uint32_t)(tb - F.kb1) < (uint32_t)(4 * cnt);
    }

    // emissions of both cells: np_emission / np_div_exact, operation for operation (v_pk_*_f32 would halve the
    // instruction count but not the issue cycles -- tools/valu_rates.hip -- and forces the parameters into register pairs)
    const float emx = np_emission(x0, np_gauss{F.g0m, F.g0s, F.g0c, F.g0r});
    const float emy = np_emission(x1, np_gauss{F.g1m, F.g1s, F.g1c, F.g1r});
    // the next band's event means (same k-mer, next event) go into the registers the emissions have just released: the
    // loop-carried value is the load's own destination, so nothing is copied (a copy would have to wait for the load)
    x0 = buf_f32(R.ev, off0);
    x1 = buf_f32(R.ev, off1);

    // DP cells (raw_loader.cpp:240-289), computed unconditionally and masked: fp32 cell + fp64 constant + fp32 emission
    // in fp64, rounded to fp32; max, then FROM_U / FROM_L override on equality in that order (later candidate wins)
    const double em0 = (double)emx, em1 = (double)emy;
    const float sd0 = (float)((double)F.d0 + R.lp_step + em0), sd1 = (float)((double)F.d1 + R.lp_step + em1);
    const float su0 = (float)((double)F.p0 + R.lp_stay + em0), su1 = (float)((double)F.p1 + R.lp_stay + em1);
    const float sl0 = (float)((double)l0 + R.lp_skip), sl1 = (float)((double)l1 + R.lp_skip);
    const float m0 = __builtin_fmaxf(__builtin_fmaxf(sd0, su0), sl0);
    const float m1 = __builtin_fmaxf(__builtin_fmaxf(sd1, su1), sl1);
    // the code of a slot outside the band is never read back: the walk only visits finite cells, whose best
    // predecessor is finite, hence inside its band
    uint32_t f0 = (m0 == sl0) ? 2u : ((m0 == su0) ? 1u : 0u);
    uint32_t f1 = (m1 == sl1) ? 2u : ((m1 == su1) ? 1u : 0u);
    float c0, c1;
    if (FAST) { c0 = sel_mask(F.vm0, m0, NP_NEG_INF); c1 = sel_mask(F.vm1, m1, NP_NEG_INF); }
    else { c0 = v0 ? m0 : NP_NEG_INF; c1 = v1 ? m1 : NP_NEG_INF; }

    if (TRIM && llk <= -1) {
        // the window still contains k-mer -1: start cell of band 0 (:152-157) and the trim column (:216-225).
        // k = -1 lives in ring slot 127 (lane 63, second register); its event is b - 1.
        const int et = b - 1;
        if (lane == 63 && F.kb1 == -4) {
            if (et == -1) { c1 = 0.0f; f1 = 0u; }
            else if (et >= 0 && et < E) { c1 = (float)(R.lp_trim * (double)(et + 1)); f1 = 1u; }
            else { c1 = NP_NEG_INF; f1 = 0u; }
        }
    }

    // packed trace: every lane keeps the 2-bit codes of its two slots, 4 bits per band, and stores one dword per
    // 8 bands (64 lanes x 4 B = 256 B coalesced = 32 B/band).  The back-track reads the word back into the SAME lane.
    // (the word shifts down one nibble per band, so after 8 bands band b%8 == 0 sits in bits 0..3: no variable shift)
    F.tacc = (F.tacc >> 4) | ((f0 | (f1 << 2)) << 28);
    if ((b & 7) == 7) R.trace32[(size_t)(b >> 3) * 64 + lane] = F.tacc;

    if (END && khi == K - 1 && llk <= K - 1) {
        // end search: cell (e, K-1) while it is inside the window, any e in [0,E) (:309-324)
        const bool mine0 = (R.end_slot < 64) && lane == R.end_slot;
        const bool mine1 = (R.end_slot >= 64) && lane == R.end_slot - 64;
        if (mine0 || mine1) {
            const int kb = mine0 ? F.kb0 : F.kb1;
            const int e = b - 2 - (kb >> 2);
            const float v = mine0 ? c0 : c1;
            if (kb == 4 * (K - 1) && e >= 0 && e < E) {
                const float sc = (float)((double)v + (double)(E - e) * R.lp_trim);
                if (sc > F.best) { F.best = sc; F.best_e = e; }
            }
        }
    }
    F.d0 = l0; F.d1 = l1;
    F.p0 = c0; F.p1 = c1;

    // The event means requested above must have landed before the parameter request below is issued: loads return in
    // order, and a wait placed after a conditional request would have to assume it was not made and drain everything.
    // Here they have had the whole band to arrive; the parameter records then have the whole next band.
    // (the band's results are operands too, which pins the statement -- and so the wait -- behind the band's arithmetic)
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(c0), "+v"(c1));
    F.p0 = c0; F.p1 = c1;
    if (b >= 1) {
        // Suzuki's rule for band b+1, on this band (:179-195).
        const int ll = pin_s(ring_read_bits(c0, c1, llk & (NP_RING - 1)));
        const int ur = pin_s(ring_read_bits(c0, c1, (llk + NP_ALN_BANDWIDTH - 1) & (NP_RING - 1)));
        // both -inf (the AND of two non-NaN patterns is -inf's only then): alternate; else right iff ll < ur, where a
        // single -inf compares as the reference's is_offset_valid ? value : -INFINITY does
        const bool both_ob = (ll & ur) == (int)0xff800000;
        const bool right = both_ob ? ((b & 1) == 0) : (__builtin_bit_cast(float, ll) < __builtin_bit_cast(float, ur));
        if (right) {
            F.llk = llk + 1;
            // Exactly one ring slot falls out per right move: slot (llk - 15) mod 128, one lane of one of the two slot
            // registers.  It takes its next k-mer (k+128), whose record was requested the last time the slot moved
            // (128 right-moves ago), and requests the one after that.  Every lane re-requests its `next` record (an L1
            // hit for all but the re-targeted slot): no load sits inside a divergent branch, where hipcc would wait for
            // it on the spot.
            const int out = (F.llk - NP_MARGIN - 1) & (NP_RING - 1);
            const uint64_t bit = 1ull << (out & 63);
            retarget(F, (out & 64) ? 0ull : bit, (out & 64) ? bit : 0ull);
            __builtin_amdgcn_sched_barrier(0);      // request after the moves have read the old records: same registers
            F.n0 = buf_f32x4(R.kp, F.kb0 * 4 + 16 * NP_RING);
            F.n1 = buf_f32x4(R.kp, F.kb1 * 4 + 16 * NP_RING);
            if (FAST) { F.vm0 = window_mask(lane, 0, F.llk); F.vm1 = window_mask(lane, 1, F.llk); }
        }
    }
}

__global__ void __launch_bounds__(NP_ALIGN_BLOCK, 7) np_event_align_kernel(np_align_args a)
{
    const int lane = threadIdx.x & 63;
    // readfirstlane: tells the compiler the value is wave-uniform, so pointers derived from it stay in SGPRs (buffer
    // descriptors must be scalar; a VGPR descriptor costs a waterfall loop per load)
    const int wave_slot = __builtin_amdgcn_readfirstlane(blockIdx.x * (NP_ALIGN_BLOCK / 64) + (threadIdx.x >> 6));
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
            R.ev = make_rsrc(ev, (uint32_t)E * 4u); R.kp = make_rsrc(kp, (uint32_t)K * 16u); R.trace32 = 
 uniform_ptr((uint32_t*)trace);

            // wave-uniform constants in scalar registers (a VALU fp64 add takes one SGPR-pair operand): 8 VGPRs saved
            R.lp_skip = uniform_f64(rd->lp_skip); R.lp_stay = uniform_f64(rd->lp_stay);
            R.lp_step = uniform_f64(rd->lp_step); R.lp_trim = uniform_f64(rd->lp_trim);
            fill_t F;
            F.llk = -1 - NP_ALN_BANDWIDTH / 2;              // band_lower_left[0].kmer_idx, raw_loader.cpp:150-151
            {
                const int k0 = ring_kmer(lane, F.llk), k1 = ring_kmer(lane + 64, F.llk);
                F.kb0 = 4 * k0; F.kb1 = 4 * k1;
                const float4 g0 = buf_f32x4(R.kp, 16 * k0), g1 = buf_f32x4(R.kp, 16 * k1);
                F.g0m = g0.x; F.g0s = g0.y; F.g0c = g0.z; F.g0r = g0.w; F.g1m = g1.x; F.g1s = g1.y; F.g1c = g1.z; F.g1r = g1.w;
                F.n0 = buf_f32x4(R.kp, 16 * (k0 + NP_RING)); F.n1 = buf_f32x4(R.kp, 16 * (k1 + NP_RING));
            }
            F.p0 = F.p1 = F.d0 = F.d1 = NP_NEG_INF;
            F.best = NP_NEG_INF; F.best_e = 0; F.tacc = 0u;
            float x0 = 0.0f, x1 = 0.0f;                     // event means of the current band's two cells
            int b = 0;
            // Three phases, so that the long middle of the read pays for no band geometry, trim column or end search.
            // llk and u = b-2-llk (the event of the window's first k-mer) never decrease and exactly one of them grows
            // per band, so the FAST conditions
            //   llk >= 0,  u >= 99  (the window's last k-mer has event u-99 >= 0)  -- become true once, and
            //   llk + 99 < K-1,  u <= E-1                                           -- become false once;
            // and while min(K-2 - (llk+99), E-1 - u) = s > 0 the next s bands are FAST whatever the moves are.
            for (; b < n_bands && !(F.llk >= 0 && b - 2 - F.llk >= NP_ALN_BANDWIDTH - 1); ++b)
                band_step<true, true, false>(F, R, b, x0, x1);
            F.vm0 = window_mask(lane, 0, F.llk); F.vm1 = window_mask(lane, 1, F.llk);
            for (;;) {
                const int ks = (K - 2) - (F.llk + NP_ALN_BANDWIDTH - 1), es = (E - 1) - (b - 2 - F.llk);
                int stop = b + (ks < es ? ks : es);
                stop = stop < n_bands ? stop : n_bands;
                if (stop <= b) break;
                for (; b < stop; ++b) band_step<false, false, true>(F, R, b, x0, x1);
            }
            for (; b < n_bands; ++b) band_step<true, true, false>(F, R, b, x0, x1);
            if ((n_bands & 7) != 0) R.trace32[(size_t)((n_bands - 1) >> 3) * 64 + lane] = F.tacc >> (4 * (8 - (n_bands & 7)));   // last, partial group

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
