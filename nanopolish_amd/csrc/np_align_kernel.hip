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

// Build-time switches of the band step (A/B builds, tools/align_variants.sh):
#ifndef NP_A_INTCMP
#define NP_A_INTCMP 1     // Suzuki's rule in the FAST phase as one scalar unsigned compare (reads whose cells are all <= 0)
#endif
#ifndef NP_A_ONEREC
#define NP_A_ONEREC 1     // one pending k-mer record per lane (its two ring slots re-target alternately) instead of two
#endif
#ifndef NP_A_EARLYSUZ
#define NP_A_EARLYSUZ 1   // FAST: read the band ends for Suzuki's rule before packing the trace codes
#endif
#ifndef NP_A_UNROLL8
#define NP_A_UNROLL8 0    // FAST phase in blocks of 8 bands aligned with the trace groups: the trace store, the loop control and
#endif                    // the event-mean offsets (instruction immediates) are paid once per block, not once per band.
                          // Measured SLOWER than two bands per iteration (52.9 vs 51.6 ms per 32768 reads) although it issues
                          // ~9 % fewer instructions per band: the fill is bound by its vector instructions (22 double-rate
                          // ones per band), not by the scalar bookkeeping the blocks remove.  Needs NP_A_CMPX.
#ifndef NP_A_TRACEASM
#define NP_A_TRACEASM 1   // FAST: the four trace compares write four scalar pairs before the selects read them (no hazard s_nops)
#endif
#ifndef NP_A_CMPX
#define NP_A_CMPX 1       // re-target: the one-lane exec masks come from v_cmpx on the slot index instead of scalar shifts
#endif
#ifndef NP_BT_DEPTH
#define NP_BT_DEPTH 8     // back-track: trace groups (8 bands each) requested ahead of the walk
#endif
#ifndef NP_A_DDBL
#define NP_A_DDBL 1       // the diagonal source is kept as the double the previous band converted for `left`
#endif
// ablations: timing experiments only, results are WRONG with any of these set
#ifndef NP_ABL
#define NP_ABL 0          // bit mask: 1 no trace, 2 fp32 candidate sums, 4 fixed move pattern (keeps the control flow of the
#endif                    // other ablations comparable), 8 no event loads, 16 trivial emission, 32 no lane-0 wrap selects,
                          // 64 no k-mer record loads, 128 no back-track
#ifndef NP_PROBE_SALU
#define NP_PROBE_SALU 0   // issue-cost probes: this many extra scalar / vector adds per FAST band (timing experiments)
#endif
#ifndef NP_PROBE_VALU
#define NP_PROBE_VALU 0
#endif
#ifndef NP_A_WAVES
#define NP_A_WAVES 7      // resident waves per SIMD the register budget is set for
#endif


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
    int kb0, kb1;           // 4 * (k-mer mapped to this lane's two ring slots); inside the FAST blocks of 8 bands instead the
                            // byte offset of the block's first event mean, 4 * (b0 - 1) - 4 k (band POS adds 4 * POS)
    float g0m, g0s, g0c, g0r;   // scaled Gaussian of slot l: mean, -stdv, log-constant, 1/stdv (scalars, so that a
    float g1m, g1s, g1c, g1r;   // re-target can update them in place) -- and of slot l + 64
#if NP_A_ONEREC
    float4 n0;              // the record this lane needs at its NEXT re-target.  A lane's two slots re-target alternately, 64
    int nko;                // right moves apart, and the k-mers they take advance by 64 each time: one pending record
                            // (byte offset nko in the slab) serves both slots
#else
    float4 n0, n1;          // the records of k0+128 / k1+128, requested when the slot was last re-targeted
#endif
    float p0, p1;           // band b-1
#if NP_A_DDBL
    double d0, d1;          // band b-2 rotated by one slot, as the double the previous band made of its `left`
#else
    float d0, d1;           // band b-2 rotated by one slot
#endif
    float best; int best_e; // end-cell search (:309-324): wave-uniform
    uint32_t tacc;          // trace codes of the last (up to) 8 bands, 4 bits per band, newest in the top nibble
    uint64_t vm0, vm1;      // FAST phase: lanes whose slot (l, l + 64) is inside the window (wave-uniform lane masks)
};

struct read_t {
    int E, K, lane, end_slot;
    bool nonpos;                    // every emission constant of the read is <= 0, hence every DP cell is (see the prologue)
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
// EO: kb0 / kb1 currently hold block-relative event offsets (FAST blocks), which DROP by 4 * 128 when the k-mer grows by 128.
template <bool EO>
__device__ __forceinline__ void retarget(fill_t& F, uint64_t m0, uint64_t m1, const int lane_id)
{
    uint64_t save;
#if NP_A_ONEREC && NP_A_CMPX
    // (this variant is called with m0 = the slot that falls out, 0..127, in its low word; the lane of slot register 0 that
    //  holds slot `out` is lane == out, of slot register 1 lane == out - 64: two v_cmpx on the lane id select it)
    const int out = (int)(uint32_t)m0, out1 = out - 64;
    if (EO) {
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_eq_u32_e32 %[o0], %[ln]\n\t"
                 "v_mov_b32 %[a0], %[x0]\n\tv_mov_b32 %[a1], %[x1]\n\tv_mov_b32 %[a2], %[x2]\n\tv_mov_b32 %[a3], %[x3]\n\t"
                 "v_subrev_u32 %[k0], 0x200, %[k0]\n\t"
                 "v_add_u32 %[nk], 0x400, %[nk]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_cmpx_eq_u32_e32 %[o1], %[ln]\n\t"
                 "v_mov_b32 %[b0], %[x0]\n\tv_mov_b32 %[b1], %[x1]\n\tv_mov_b32 %[b2], %[x2]\n\tv_mov_b32 %[b3], %[x3]\n\t"
                 "v_subrev_u32 %[k1], 0x200, %[k1]\n\t"
                 "v_add_u32 %[nk], 0x400, %[nk]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(save), [a0] "+v"(F.g0m), [a1] "+v"(F.g0s), [a2] "+v"(F.g0c), [a3] "+v"(F.g0r), [k0] "+v"(F.kb0),
                   [b0] "+v"(F.g1m), [b1] "+v"(F.g1s), [b2] "+v"(F.g1c), [b3] "+v"(F.g1r), [k1] "+v"(F.kb1), [nk] "+v"(F.nko)
                 : [o0] "s"(out), [o1] "s"(out1), [ln] "v"(lane_id), [x0] "v"(F.n0.x), [x1] "v"(F.n0.y), [x2] "v"(F.n0.z), [x3] "v"(F.n0.w)
                 : "vcc");
    } else {
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_eq_u32_e32 %[o0], %[ln]\n\t"
                 "v_mov_b32 %[a0], %[x0]\n\tv_mov_b32 %[a1], %[x1]\n\tv_mov_b32 %[a2], %[x2]\n\tv_mov_b32 %[a3], %[x3]\n\t"
                 "v_add_u32 %[k0], 0x200, %[k0]\n\t"
                 "v_add_u32 %[nk], 0x400, %[nk]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_cmpx_eq_u32_e32 %[o1], %[ln]\n\t"
                 "v_mov_b32 %[b0], %[x0]\n\tv_mov_b32 %[b1], %[x1]\n\tv_mov_b32 %[b2], %[x2]\n\tv_mov_b32 %[b3], %[x3]\n\t"
                 "v_add_u32 %[k1], 0x200, %[k1]\n\t"
                 "v_add_u32 %[nk], 0x400, %[nk]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(save), [a0] "+v"(F.g0m), [a1] "+v"(F.g0s), [a2] "+v"(F.g0c), [a3] "+v"(F.g0r), [k0] "+v"(F.kb0),
                   [b0] "+v"(F.g1m), [b1] "+v"(F.g1s), [b2] "+v"(F.g1c), [b3] "+v"(F.g1r), [k1] "+v"(F.kb1), [nk] "+v"(F.nko)
                 : [o0] "s"(out), [o1] "s"(out1), [ln] "v"(lane_id), [x0] "v"(F.n0.x), [x1] "v"(F.n0.y), [x2] "v"(F.n0.z), [x3] "v"(F.n0.w)
                 : "vcc");
    }
    return;
#elif NP_A_ONEREC
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[m0]\n\t"
                 "v_mov_b32 %[a0], %[x0]\n\tv_mov_b32 %[a1], %[x1]\n\tv_mov_b32 %[a2], %[x2]\n\tv_mov_b32 %[a3], %[x3]\n\t"
                 "v_add_u32 %[k0], 0x200, %[k0]\n\t"
                 "v_add_u32 %[nk], 0x400, %[nk]\n\t"
                 "s_mov_b64 exec, %[m1]\n\t"
                 "v_mov_b32 %[b0], %[x0]\n\tv_mov_b32 %[b1], %[x1]\n\tv_mov_b32 %[b2], %[x2]\n\tv_mov_b32 %[b3], %[x3]\n\t"
                 "v_add_u32 %[k1], 0x200, %[k1]\n\t"
                 "v_add_u32 %[nk], 0x400, %[nk]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(save), [a0] "+v"(F.g0m), [a1] "+v"(F.g0s), [a2] "+v"(F.g0c), [a3] "+v"(F.g0r), [k0] "+v"(F.kb0),
                   [b0] "+v"(F.g1m), [b1] "+v"(F.g1s), [b2] "+v"(F.g1c), [b3] "+v"(F.g1r), [k1] "+v"(F.kb1), [nk] "+v"(F.nko)
                 : [m0] "s"(m0), [m1] "s"(m1), [x0] "v"(F.n0.x), [x1] "v"(F.n0.y), [x2] "v"(F.n0.z), [x3] "v"(F.n0.w));
    return;
#else
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
#endif
}

// One band.  (x0, x1): event means of this band's two cells on entry (loaded during the previous band), of the next
// band's on exit.  TRIM: the window may still contain k-mer -1.  END: the window may contain k-mer K-1.
// FAST: the middle of the read -- every cell of the window [llk, llk+99] exists (0 <= llk, llk+99 < K-1, and the events
// of the window's first and last k-mer are inside [0, E)), so a slot is valid iff it is inside the window, which only
// changes on a right move: the lane masks F.vm0/F.vm1 replace the per-band geometry.
// POS: b & 7 when the caller knows it (the FAST blocks), else -1.
template <bool TRIM, bool END, bool FAST, int POS = -1>
__device__ __forceinline__ void band_step(fill_t& F, const read_t& R, const int b, float& x0, float& x1)
{
    const int lane = R.lane, E = R.E, K = R.K;
    const int llk = F.llk;
    // left sources: band b-1 rotated by one slot
    const float r0 = wave_ror1_all(F.p0), r1 = wave_ror1_all(F.p1);
#if NP_ABL & 32
    const float l0 = r0, l1 = r1;
#else
    const float l0 = lane == 0 ? r1 : r0;
    const float l1 = lane == 0 ? r0 : r1;
#endif

    // A slot holds a cell of this band iff its k-mer is inside the window and its event e = b-2-k exists, i.e. iff
    // e is in [max(0, b-2-khi), min(E-1, b-2-klo)]: one unsigned range test on 4*(e+1), the byte offset of the NEXT
    // band's event mean, which the prefetch needs anyway.
    const int klo = llk > 0 ? llk : 0;
    const int khi = (llk + NP_ALN_BANDWIDTH - 1) < (K - 1) ? (llk + NP_ALN_BANDWIDTH - 1) : (K - 1);
    int off0 = 0, off1 = 0;
    if (POS < 0) { off0 = 4 * (b - 1) - F.kb0; off1 = 4 * (b - 1) - F.kb1; }
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
#if NP_ABL & 16
    const float emx = x0 * F.g0c, emy = x1 * F.g1c;
#else
    const float emx = np_emission_nd(x0, F.g0m, F.g0s, F.g0c, F.g0r);
    const float emy = np_emission_nd(x1, F.g1m, F.g1s, F.g1c, F.g1r);
#endif
    // the next band's event means (same k-mer, next event) go into the registers the emissions have just released: the
    // loop-carried value is the load's own destination, so nothing is copied (a copy would have to wait for the load)
#if NP_ABL & 8
    x0 = __builtin_bit_cast(float, off0 & 0x3f800000); x1 = __builtin_bit_cast(float, off1 & 0x3f800000);
#else
    if (POS >= 0) {
        // block-relative: the register part is the block's first offset (>= 4 for every in-window slot of a FAST band; a
        // slot outside the window may see its sum misjudged by the range check -- it only feeds masked cells), the band's
        // share is the instruction's immediate
        x0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R.ev, F.kb0 + 4 * POS, 0, 0));
        x1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R.ev, F.kb1 + 4 * POS, 0, 0));
    } else {
        x0 = buf_f32(R.ev, off0);
        x1 = buf_f32(R.ev, off1);
    }
#endif

    // DP cells (raw_loader.cpp:240-289), computed unconditionally and masked: fp32 cell + fp64 constant + fp32 emission
    // in fp64, rounded to fp32; max, then FROM_U / FROM_L override on equality in that order (later candidate wins)
#if NP_ABL & 2
    const float sd0 = (float)F.d0 + (float)R.lp_step + emx, sd1 = (float)F.d1 + (float)R.lp_step + emy;
    const float su0 = F.p0 + (float)R.lp_stay + emx, su1 = F.p1 + (float)R.lp_stay + emy;
    const float sl0 = l0 + (float)R.lp_skip, sl1 = l1 + (float)R.lp_skip;
    const double L0 = l0, L1 = l1;
#else
    const double em0 = (double)emx, em1 = (double)emy;
    const double L0 = (double)l0, L1 = (double)l1;           // `left` now, `diagonal` of the next band
    const float sd0 = (float)((double)F.d0 + R.lp_step + em0), sd1 = (float)((double)F.d1 + R.lp_step + em1);
    const float su0 = (float)((double)F.p0 + R.lp_stay + em0), su1 = (float)((double)F.p1 + R.lp_stay + em1);
    const float sl0 = (float)(L0 + R.lp_skip), sl1 = (float)(L1 + R.lp_skip);
#endif
#if NP_PROBE_SALU || NP_PROBE_VALU
    if (FAST) {
        int ps = b, pv = lane;
#pragma unroll
        for (int i = 0; i < NP_PROBE_SALU; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ps) : : "scc");
#pragma unroll
        for (int i = 0; i < NP_PROBE_VALU; ++i) asm volatile("v_add_u32 %0, 1, %0" : "+v"(pv));
        asm volatile("" : : "s"(ps), "v"(pv));
    }
#endif
    const float m0 = __builtin_fmaxf(__builtin_fmaxf(sd0, su0), sl0);
    const float m1 = __builtin_fmaxf(__builtin_fmaxf(sd1, su1), sl1);
    // the code of a slot outside the band is never read back: the walk only visits finite cells, whose best
    // predecessor is finite, hence inside its band
    uint32_t f0, f1;
#if NP_A_TRACEASM
    if (FAST) {
        // f0 | f1 << 2 in one go.  A vector compare's mask cannot be used by the very next vector instructions (two wait states
        // on gfx950; hipcc funnels all four compares through VCC and pads with s_nop): here the four compares write four
        // scalar pairs, and by the time a select reads its mask three other instructions have issued.
        uint64_t qa, qb, qc, qd;
        uint32_t t1;
        asm("v_cmp_eq_f32_e64 %[a], %[m0], %[u0]\n\t"
            "v_cmp_neq_f32_e64 %[b], %[m0], %[l0]\n\t"
            "v_cmp_eq_f32_e64 %[c], %[m1], %[u1]\n\t"
            "v_cmp_neq_f32_e64 %[d], %[m1], %[l1]\n\t"
            "v_cndmask_b32_e64 %[t0], 0, 1, %[a]\n\t"
            "v_cndmask_b32_e64 %[t1], 0, 4, %[c]\n\t"
            "v_cndmask_b32_e64 %[t0], 2, %[t0], %[b]\n\t"
            "v_cndmask_b32_e64 %[t1], 8, %[t1], %[d]\n\t"
            "v_or_b32_e32 %[t0], %[t0], %[t1]"
            : [t0] "=&v"(f0), [t1] "=&v"(t1), [a] "=&s"(qa), [b] "=&s"(qb), [c] "=&s"(qc), [d] "=&s"(qd)
            : [m0] "v"(m0), [u0] "v"(su0), [l0] "v"(sl0), [m1] "v"(m1), [u1] "v"(su1), [l1] "v"(sl1));
        f1 = 0u;
    } else
#endif
    {
        f0 = (m0 == sl0) ? 2u : ((m0 == su0) ? 1u : 0u);
        f1 = (m1 == sl1) ? 2u : ((m1 == su1) ? 1u : 0u);
    }
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
    auto encode_trace = [&]() {
#if !(NP_ABL & 1)
        F.tacc = (F.tacc >> 4) | ((f0 | (f1 << 2)) << 28);
        if (POS == 7 || (POS < 0 && (b & 7) == 7)) R.trace32[(size_t)(b >> 3) * 64 + lane] = F.tacc;
#endif
    };
#if NP_A_EARLYSUZ
    // FAST: the band ends of Suzuki's rule are read out (v_readlane -> scalar compares) BEFORE the trace codes are packed,
    // so that the scalar chain of the move decision runs while the vector unit packs -- in source order the wave would sit
    // through the v_readlane -> s_cselect -> s_cmp latency with nothing else to issue
    int ll_early = 0, ur_early = 0;
    if (FAST) {
        ll_early = ring_read_bits(c0, c1, llk & (NP_RING - 1));
        ur_early = ring_read_bits(c0, c1, (llk + NP_ALN_BANDWIDTH - 1) & (NP_RING - 1));
        asm volatile("" : "+s"(ll_early), "+s"(ur_early));
    }
#endif
    encode_trace();

    if (END && khi == K - 1 && llk <= K - 1) {
        // end search: cell (e, K-1) while it is inside the window, any e in [0,E) (:309-324).  Wave-uniform: the cell's ring
        // slot is R.end_slot, which holds k-mer K-1 iff that is the k-mer the ring maps the slot to at this window position
        const int e = b - 2 - (K - 1);
        if (ring_kmer(R.end_slot, llk) == K - 1 && e >= 0 && e < E) {
            const float v = ring_read(c0, c1, R.end_slot);
            const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int,
                                 (float)((double)v + (double)(E - e) * R.lp_trim))));
            if (sc > F.best) { F.best = sc; F.best_e = e; }
        }
    }
#if NP_A_DDBL
    F.d0 = L0; F.d1 = L1;
#else
    F.d0 = l0; F.d1 = l1;
#endif
    F.p0 = c0; F.p1 = c1;

    // The event means requested above must have landed before the parameter request below is issued: loads return in
    // order, and a wait placed after a conditional request would have to assume it was not made and drain everything.
    // Here they have had the whole band to arrive; the parameter records then have the whole next band.
    // (the band's results are operands too, which pins the statement -- and so the wait -- behind the band's arithmetic)
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(c0), "+v"(c1));
    F.p0 = c0; F.p1 = c1;
    if (FAST || b >= 1) {
        // Suzuki's rule for band b+1, on this band (:179-195).
#if NP_A_EARLYSUZ
        const int ll = FAST ? ll_early : pin_s(ring_read_bits(c0, c1, llk & (NP_RING - 1)));
        const int ur = FAST ? ur_early : pin_s(ring_read_bits(c0, c1, (llk + NP_ALN_BANDWIDTH - 1) & (NP_RING - 1)));
#else
        const int ll = pin_s(ring_read_bits(c0, c1, llk & (NP_RING - 1)));
        const int ur = pin_s(ring_read_bits(c0, c1, (llk + NP_ALN_BANDWIDTH - 1) & (NP_RING - 1)));
#endif
        // both -inf (the AND of two non-NaN patterns is -inf's only then): alternate; else right iff ll < ur, where a
        // single -inf compares as the reference's is_offset_valid ? value : -INFINITY does
#if NP_ABL & 4
        const bool right = (((b + 1) * 13) >> 5) != ((b * 13) >> 5);       // a fixed pattern with the same share of right moves
#else
        bool right;
        if (FAST && NP_A_INTCMP) {
            // FAST runs only for reads whose cells are all <= 0 (R.nonpos): for such floats (+0, negative, -inf) x < y is
            // bits(x) > bits(y) as unsigned integers -- scalar compares instead of a vector compare on two scalars
            // (spelled out for the scalar unit: hipcc lowers a select between wave-uniform conditions to vector code)
            int r, t;
            const int par = (b & 1) ^ 1;                                                          // both outside: alternate
            asm("s_and_b32 %[t], %[ll], %[ur]\n\t"
                "s_cmp_gt_u32 %[ll], %[ur]\n\t"
                "s_cselect_b32 %[r], 1, 0\n\t"
                "s_cmp_eq_u32 %[t], 0xff800000\n\t"
                "s_cselect_b32 %[r], %[par], %[r]"
                : [r] "=&s"(r), [t] "=&s"(t) : [ll] "s"(ll), [ur] "s"(ur), [par] "s"(par) : "scc");
            right = r != 0;
        } else {
            const bool both_ob = (ll & ur) == (int)0xff800000;
            right = both_ob ? ((b & 1) == 0) : (__builtin_bit_cast(float, ll) < __builtin_bit_cast(float, ur));
        }
#endif
        if (right) {
            F.llk = llk + 1;
            // Exactly one ring slot falls out per right move: slot (llk - 15) mod 128, one lane of one of the two slot
            // registers.  It takes its next k-mer (k+128), whose record was requested the last time the slot moved
            // (128 right-moves ago), and requests the one after that.  Every lane re-requests its `next` record (an L1
            // hit for all but the re-targeted slot): no load sits inside a divergent branch, where hipcc would wait for
            // it on the spot.
            const int out = (F.llk - NP_MARGIN - 1) & (NP_RING - 1);
#if NP_A_ONEREC && NP_A_CMPX
            retarget<(POS >= 0)>(F, (uint64_t)(uint32_t)out, 0ull, lane);
#else
            const uint64_t bit = 1ull << (out & 63);
            retarget<false>(F, (out & 64) ? 0ull : bit, (out & 64) ? bit : 0ull, lane);
#endif
            __builtin_amdgcn_sched_barrier(0);      // request after the moves have read the old records: same registers
#if NP_ABL & 64
            F.n0.x += 1.0f;
#elif NP_A_ONEREC
            F.n0 = buf_f32x4(R.kp, F.nko);
#else
            F.n0 = buf_f32x4(R.kp, F.kb0 * 4 + 16 * NP_RING);
            F.n1 = buf_f32x4(R.kp, F.kb1 * 4 + 16 * NP_RING);
#endif
            if (FAST) { F.vm0 = window_mask(lane, 0, F.llk); F.vm1 = window_mask(lane, 1, F.llk); }
        }
    }
}

__global__ void __launch_bounds__(NP_ALIGN_BLOCK, NP_A_WAVES) np_event_align_kernel(np_align_args a)
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
        const int ticket = __builtin_amdgcn_readfirstlane((int)atomicAdd(a.counter, lane == 0 ? 1u : 0u));
        if (ticket >= a.n_reads) break;
        // longest reads first (np_align_order_*): with ragged read lengths the kernel's tail is then made of short reads
        const int ri = a.order ? __builtin_amdgcn_readfirstlane((int)a.order[ticket]) : ticket;

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
            float cl_max = NP_NEG_INF;
            for (int k = lane; k < K; k += 64) {
                const uint32_t r = rk[k];
                const np_gauss g = np_make_gauss(model[r].level_mean, model[r].level_stdv, model[r].level_log_stdv, scale, shift, var, log_var);
                kp[k] = make_float4(g.mean, -g.stdv, g.cl, g.rinv);        // the record carries -sigma (np_emission_nd)
                cl_max = __builtin_fmaxf(cl_max, g.cl);
            }
            // An emission is cl - a^2/2 <= cl and the four transition constants are logarithms of probabilities: if no cl is
            // positive (r9.4: cl <= -1.1) and no constant is, every DP cell is a sum of non-positive terms, i.e. +0, negative
            // or -inf -- what the FAST phase's integer form of Suzuki's comparison relies on.  Other reads (a model with
            // sigma < 0.4 pA) take the generic step for every band.
            const bool nonpos = __builtin_amdgcn_ballot_w64(cl_max > 0.0f) == 0ull &&
                                !(rd->lp_skip > 0.0) && !(rd->lp_stay > 0.0) && !(rd->lp_step > 0.0) && !(rd->lp_trim > 0.0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);

            // ---------------- fill ----------------
            read_t R;
            R.E = E; R.K = K; R.lane = lane; R.end_slot = (K - 1) & (NP_RING - 1); R.nonpos = nonpos;
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
#if NP_A_ONEREC
                F.nko = 16 * ((k0 < k1 ? k0 : k1) + NP_RING);        // the slot with the smaller k-mer re-targets first
                F.n0 = buf_f32x4(R.kp, F.nko);
#else
                F.n0 = buf_f32x4(R.kp, 16 * (k0 + NP_RING)); F.n1 = buf_f32x4(R.kp, 16 * (k1 + NP_RING));
#endif
            }
            F.p0 = F.p1 = NP_NEG_INF; F.d0 = F.d1 = NP_NEG_INF;
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
            for (; !NP_A_INTCMP || R.nonpos;) {
                const int ks = (K - 2) - (F.llk + NP_ALN_BANDWIDTH - 1), es = (E - 1) - (b - 2 - F.llk);
                int stop = b + (ks < es ? ks : es);
                stop = stop < n_bands ? stop : n_bands;
                if (stop <= b) break;
#if NP_A_UNROLL8
                // blocks of 8 bands that coincide with the trace groups
                for (; b < stop && (b & 7) != 0; ++b) band_step<false, false, true>(F, R, b, x0, x1);
                // (inside the blocks kb0 / kb1 hold the block-relative event offsets: converted here and back below)
                F.kb0 = 4 * (b - 1) - F.kb0; F.kb1 = 4 * (b - 1) - F.kb1;
                for (; b + 7 < stop; b += 8) {
                    band_step<false, false, true, 0>(F, R, b, x0, x1); band_step<false, false, true, 1>(F, R, b + 1, x0, x1);
                    band_step<false, false, true, 2>(F, R, b + 2, x0, x1); band_step<false, false, true, 3>(F, R, b + 3, x0, x1);
                    band_step<false, false, true, 4>(F, R, b + 4, x0, x1); band_step<false, false, true, 5>(F, R, b + 5, x0, x1);
                    band_step<false, false, true, 6>(F, R, b + 6, x0, x1); band_step<false, false, true, 7>(F, R, b + 7, x0, x1);
                    F.kb0 += 32; F.kb1 += 32;
                }
                F.kb0 = 4 * (b - 1) - F.kb0; F.kb1 = 4 * (b - 1) - F.kb1;
#elif NP_A_DDBL
                // two bands per iteration: the doubles made of `left` become the next band's diagonal without a register copy
                for (; b + 1 < stop; b += 2) { band_step<false, false, true>(F, R, b, x0, x1); band_step<false, false, true>(F, R, b + 1, x0, x1); }
#endif
                for (; b < stop; ++b) band_step<false, false, true>(F, R, b, x0, x1);
            }
            for (; b < n_bands; ++b) band_step<true, true, false>(F, R, b, x0, x1);
            if ((n_bands & 7) != 0) R.trace32[(size_t)((n_bands - 1) >> 3) * 64 + lane] = F.tacc >> (4 * (8 - (n_bands & 7)));   // last, partial group

            // ---------------- backtrack (:326-361) + QC sums (:338-341) ----------------
            const float best_u = F.best;
            int curr_e = F.best_e;
            int curr_k = K - 1;

            // the trace was written by lanes 0..3 of this wave: complete the stores before other lanes read them back
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);

            if (best_u != NP_NEG_INF && !(NP_ABL & 128)) {
                // Scalar walk.  The kernel is instruction-issue bound (~2.3 cycles per wave-instruction of any kind, measured
                // with tools/align_variants.sh probes), and a walk of ~0.63 steps per band is a fifth of its instructions, so
                // the step is written out by hand: 19 instructions.
                //   * A step reads the 2-bit code of its cell out of the lane that owns the k-mer's ring slot (v_readlane) and
                //     parks it in lane j of a vector register (v_writelane): nothing else is recorded.  When 64 codes have
                //     gathered, each lane rebuilds ITS pair from the chunk's start position and the population counts, below
                //     its lane id, of the "k drops" / "e drops" ballots (two v_mbcnt pairs for 64 pairs); the longest run of
                //     FROM_L steps (max_gap, :352-357) comes out of the same ballots.
                //   * The trace of a read (32 B per band, ~430 KB) was written milliseconds ago and comes back from HBM: a
                //     group of 8 bands lasts ~5 steps, an HBM read 2000+ cycles.  NP_BT_DEPTH groups (64 bands) are in flight:
                //     the group loop is unrolled NP_BT_DEPTH times so that every position owns one register of the queue,
                //     reloaded (for the group NP_BT_DEPTH below) as soon as its group is left.  The loads are issued and
                //     awaited by hand (hipcc would wait for the NEWEST load at every use): loads return in order, so "at most
                //     NP_BT_DEPTH - 1 operations outstanding" means the oldest request -- this position's -- has landed.
                const uint32_t* __restrict__ t32 = (const uint32_t*)trace;
                int cg = (curr_e + curr_k + 2) >> 3;
                uint32_t tq[NP_BT_DEPTH];
#define NP_BT_LOAD(dst, g) { const uint32_t* p_ = t32 + (size_t)((g) > 0 ? (g) : 0) * 64 + lane; \
                             asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p_) : "memory"); }
#pragma unroll
                for (int i = 0; i < NP_BT_DEPTH; ++i) NP_BT_LOAD(tq[i], cg - i);
                int j = 0, from = 0, gap = 0;
                int k0 = curr_k, e0 = curr_e;          // position of the chunk's first step
                int vfrom = 0;                         // lane j: the code of step j of the chunk
                // up to 64 pairs: stored at descending addresses (their emissions are added up after the walk)
                auto flush = [&]() {
                    const bool valid = lane < j;
                    const uint64_t mk = __builtin_amdgcn_ballot_w64(valid && vfrom != 1);          // FROM_D, FROM_L: k drops
                    const uint64_t me = __builtin_amdgcn_ballot_w64(valid && vfrom != 2);          // FROM_D, FROM_U: e drops
                    if (valid) {
                        np_pair p;
                        p.ref_pos = k0 - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                        p.read_pos = e0 - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(me >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)me, 0u));
                        pairs[cap - 1 - (n_out + lane)] = p;
                    }
                    // runs of FROM_L steps: ml has bit i set iff step i is FROM_L (the steps of the chunk are bits 0 .. j-1)
                    const uint64_t all = j >= 64 ? ~0ull : ((1ull << j) - 1ull);
                    const uint64_t ml = all & ~me;
                    const int lead = ml == all ? j : __builtin_ctzll(~ml);                         // run that continues the previous chunk's
                    int longest = 0;
                    for (uint64_t m = ml; m != 0; m &= m << 1) longest++;                         // longest run inside the chunk
                    longest = longest > gap + lead ? longest : gap + lead;
                    max_gap = longest > max_gap ? longest : max_gap;
                    // run still open at the chunk's end
                    if (ml == all) gap += j;
                    else gap = __builtin_clzll(~(ml << (64 - j)));                                 // (bits below 64 - j are set in the operand)
                    n_out += j; j = 0; k0 = curr_k; e0 = curr_e;
                };
                int done = 0;
                while (!done) {
#pragma unroll
                    for (int pos = 0; pos < NP_BT_DEPTH; ++pos) {
                        asm volatile("s_waitcnt vmcnt(%[n])" : [t] "+v"(tq[pos]) : [n] "n"(NP_BT_DEPTH - 1) : "memory");
                        // steps inside trace group cg: band - 8 cg = (k + e) - lb is the nibble index, negative once the walk
                        // has left the group
                        const int lb = 8 * cg - 2;
                        for (;;) {
                        int t_, nib_, c_, w_;
                        // (the step counter j lives in M0 inside the loop: v_writelane takes its lane select from M0, because a
                        //  second scalar register next to the data operand would exceed the constant-bus limit)
                        asm volatile("s_mov_b32 m0, %[j]\n\t"
                                     "1:\n\t"
                                     "s_add_i32 %[t], %[k], %[e]\n\t"
                                     "s_sub_i32 %[nib], %[t], %[lb]\n\t"
                                     "s_cmp_lt_i32 %[nib], 0\n\t"
                                     "s_cbranch_scc1 2f\n\t"
                                     "s_bfe_u32 %[c], %[k], 0x10006\n\t"              // second slot register: k bit 6
                                     "s_lshl1_add_u32 %[c], %[c], 0x20000\n\t"        // field width 2 | 2 * bit
                                     "s_lshl2_add_u32 %[c], %[nib], %[c]\n\t"         // + 4 * nibble
                                     "v_readlane_b32 %[w], %[wreg], %[k]\n\t"        // (the lane select is taken modulo 64)
                                     "s_bfe_u32 %[from], %[w], %[c]\n\t"
                                     "v_writelane_b32 %[vf], %[from], m0\n\t"
                                     "s_add_i32 m0, m0, 1\n\t"
                                     "s_cmp_lg_u32 %[from], 1\n\t"
                                     "s_subb_u32 %[k], %[k], 0\n\t"                   // k -= (from != FROM_U)
                                     "s_cmp_lg_u32 %[from], 2\n\t"
                                     "s_subb_u32 %[e], %[e], 0\n\t"                   // e -= (from != FROM_L)
                                     "s_cmp_eq_u32 m0, 64\n\t"
                                     "s_cbranch_scc1 2f\n\t"
                                     "s_or_b32 %[t], %[k], %[e]\n\t"
                                     "s_cmp_ge_i32 %[t], 0\n\t"
                                     "s_cbranch_scc1 1b\n\t"
                                     "2:\n\t"
                                     "s_mov_b32 %[j], m0"
                                     : [k] "+s"(curr_k), [e] "+s"(curr_e), [j] "+s"(j), [vf] "+v"(vfrom), [from] "+s"(from),
                                       [t] "=&s"(t_), [nib] "=&s"(nib_), [c] "=&s"(c_), [w] "=&s"(w_)
                                     : [wreg] "v"(tq[pos]), [lb] "s"(lb)
                                     : "scc", "m0");
                        done = (curr_k | curr_e) >> 31;                                 // -1 once either index is negative
                        if (j == 64 || done) flush();
                        if (done || curr_e + curr_k - lb < 0) break;                    // (else: a chunk boundary inside the group)
                        }
                        if (done) break;
                        NP_BT_LOAD(tq[pos], cg - NP_BT_DEPTH);
                        cg -= 1;
                    }
                }
                // the last recorded pair's k-mer: the position before the last step
                last_k = curr_k + (from != 1 ? 1 : 0);
#undef NP_BT_LOAD
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                // QC sum (:338-341) over the stored pairs, 64 at a time.  The reference adds the emissions in walk order; only
                // the comparison of their mean with the threshold is used.  Any summation order of n doubles is within
                // n * 2^-53 * sum|x| of the exact sum, i.e. the mean within ~1e-11 relative for the longest reads: per-lane
                // partial sums decide unless the mean is within 1e-6 of the threshold, and only then (practically never) the
                // emissions are re-added serially in walk order.
                double acc = 0.0;
                for (int base = 0; base < n_out; base += 64) {
                    if (base + lane < n_out) {
                        const np_pair p = pairs[cap - 1 - (base + lane)];
                        const float4 rec = load_kp(kp, p.ref_pos, K);
                        acc += (double)np_emission_nd(ev[p.read_pos], rec.x, rec.y, rec.z, rec.w);
                    }
                }
                double tot = acc;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 64);
                tot = uniform_f64(tot);
                const double avg = tot / (double)n_out, thr = a.min_average_log_emission;
                const double dist = avg > thr ? avg - thr : thr - avg;
                if (dist <= 1e-6 * (1.0 + (avg < 0 ? -avg : avg))) {
                    tot = 0.0;
                    for (int base = 0; base < n_out; base += 64) {
                        const int cnt = n_out - base < 64 ? n_out - base : 64;
                        float em = 0.0f;
                        if (lane < cnt) {
                            const np_pair p = pairs[cap - 1 - (base + lane)];
                            const float4 rec = load_kp(kp, p.ref_pos, K);
                            em = np_emission_nd(ev[p.read_pos], rec.x, rec.y, rec.z, rec.w);
                        }
                        for (int q = 0; q < cnt; ++q) tot += (double)readlane_f(em, q);
                    }
                }
                sum_emission = tot;
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

// ---- issue order of the read queue: a counting sort of the reads by band count, longest first ----------------------
// One wave walks one read, and a read of b bands takes b dependent steps: a long read that starts late IS the kernel's tail
// (a 6x-mean read started last costs ~6 mean read times while the rest of the chip idles).  Longest-processing-time-first
// issue bounds the tail by the shortest reads instead.  1024 buckets of 128 bands (reads beyond 131k bands share the top
// bucket, which is issued first).
#define NP_ORDER_BUCKETS 1024
__device__ __forceinline__ int order_bucket(const np_read_dev& r)
{
    const uint32_t b = (r.n_events + r.n_kmers + 2u) >> 7;
    return (int)(b < NP_ORDER_BUCKETS - 1 ? b : NP_ORDER_BUCKETS - 1);
}
__global__ void __launch_bounds__(256) np_align_hist_kernel(int n_reads, const np_read_dev* __restrict__ reads, uint32_t* __restrict__ hist)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_reads) atomicAdd(&hist[order_bucket(reads[r])], 1u);
}
// one block: cursor[b] = number of reads in longer buckets (exclusive scan from the top)
__global__ void __launch_bounds__(NP_ORDER_BUCKETS) np_align_scan_kernel(const uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor)
{
    __shared__ uint32_t s[NP_ORDER_BUCKETS];
    const int t = threadIdx.x;                      // t = 0 is the LONGEST bucket
    const uint32_t mine = hist[NP_ORDER_BUCKETS - 1 - t];
    s[t] = mine;
    __syncthreads();
    for (int o = 1; o < NP_ORDER_BUCKETS; o <<= 1) {
        const uint32_t v = t >= o ? s[t - o] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    cursor[NP_ORDER_BUCKETS - 1 - t] = s[t] - mine;
}
__global__ void __launch_bounds__(256) np_align_scatter_kernel(int n_reads, const np_read_dev* __restrict__ reads, uint32_t* __restrict__ cursor,
                                                               uint32_t* __restrict__ order)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_reads) order[atomicAdd(&cursor[order_bucket(reads[r])], 1u)] = (uint32_t)r;
}

} // namespace

int np_align_block_threads(void) { return NP_ALIGN_BLOCK; }

// scratch: 2 * NP_ORDER_BUCKETS counters (zeroed here) followed by n_reads order entries
hipError_t np_launch_align_order(int n_reads, const np_read_dev* reads, uint32_t* scratch, hipStream_t s)
{
    uint32_t* hist = scratch; uint32_t* cursor = scratch + NP_ORDER_BUCKETS; uint32_t* order = scratch + 2 * NP_ORDER_BUCKETS;
    hipError_t e = hipMemsetAsync(hist, 0, NP_ORDER_BUCKETS * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const int nb = (n_reads + 255) / 256;
    hipLaunchKernelGGL(np_align_hist_kernel, dim3(nb), dim3(256), 0, s, n_reads, reads, hist);
    hipLaunchKernelGGL(np_align_scan_kernel, dim3(1), dim3(NP_ORDER_BUCKETS), 0, s, hist, cursor);
    hipLaunchKernelGGL(np_align_scatter_kernel, dim3(nb), dim3(256), 0, s, n_reads, reads, cursor, order);
    return hipGetLastError();
}

hipError_t np_launch_event_align(const np_align_args& a, int n_blocks, hipStream_t s)
{
    hipLaunchKernelGGL(np_event_align_kernel, dim3(n_blocks), dim3(NP_ALIGN_BLOCK), 0, s, a);
    return hipGetLastError();
}
