// np_align_kernel.hip -- adaptive_banded_simple_event_align (src/nanopolish_raw_loader.cpp:77-379) for gfx950.
//
// One wavefront per read, ~E+K+2 sequential band steps.  Design (DESIGN.md section "Kernel A"):
//   * The 100-cell band lives in registers, anchored by k-mer index instead of by band offset: cell (event e,
//     k-mer k) of band b = e+k+2 sits in ring slot (k mod 128).  With that anchoring the three DP sources never
//     depend on the band's move history:
//         up   = band b-1, same slot        left = band b-1, slot-1        diag = band b-2, slot-1
//     The ring is INTERLEAVED over the wave: lane l owns slots 2l (slot register 0) and 2l+1 (slot register 1).  The left
//     neighbour of an odd slot is then the lane's own even slot (no data movement at all), that of an even slot the
//     previous lane's odd slot: ONE DPP wave_ror:1 per band, whose wrap-around (lane 0 <- lane 63) is exactly slot 0's
//     neighbour, slot 127.  A slot whose k-mer is outside the band window holds -inf, which is exactly the reference's
//     is_offset_valid(...) ? BAND_ARRAY(...) : -INFINITY.  (A split layout -- slots l and l+64 -- needs two rotates plus two
//     lane-0 selects, one more fp64 conversion and twice the v_readlane of Suzuki's rule: round 2, 284 -> ~200 SIMD cycles
//     per band together with the two items below.)
//   * Suzuki's move rule reads the band's first and last cell (ll, ur) with v_readlane: 100 is even, so the two ends always
//     sit in DIFFERENT slot registers, one v_readlane each; all band geometry is wave-uniform scalar state.
//   * Prologue per read: the scaled Gaussian of every k-mer (fp64 math of squiggle_read.h:217-226, plus the
//     correctly rounded reciprocal of sigma) goes into a per-wave slab, 16 B per k-mer; the band loop then only
//     does fp32 emissions and the reference's fp64 candidate sums.
//   * The inner step is branch-free per lane: DP cells are computed by every lane and masked.  The band loop runs in
//     three phases: a generic step (band geometry, trim column k = -1, end-cell search) for the first and last ~300
//     bands, and a FAST step for the middle of the read, where every cell of the 100-wide window exists, so validity
//     is a pair of wave-uniform lane masks that only change on a right move -- by one bit rotation on the scalar unit.
//   * Event means are prefetched one band ahead straight into the loop-carried register (the load is issued after
//     the emission has consumed the old value), through a range-checked buffer descriptor: out-of-range events read
//     as 0 and only feed masked cells, so there is no clamp.
//   * The ring holds 128 k-mers for a window of 100: slots are re-targeted (k -> k + 128) EIGHT AT A TIME, whenever the
//     window's first k-mer reaches a multiple of 8 -- the eight slots 9..16 k-mers below the window, i.e. both slots of four
//     neighbouring lanes, which load their new parameter records themselves under a four-lane exec mask.  (An exec-mask
//     region costs ~14 plain vector instructions of issue time on this chip, tools/valu_rates.hip: one region per right move,
//     as in the first version of this kernel, was a tenth of the band.)  The new k-mers are 13+ right moves away from the
//     window, so the wait for the records (inside the region: the compiler never sees a load in flight) is not on anyone's path.
//   * Issue-cycle budget (tools/valu_rates.hip, measured on MI355X; v_add_f32 = 1): fp32 add/mul/fma, integer add/and/or with
//     register or inline-constant operands 1; fp64 ops, conversions, v_cmp, v_cndmask (VOP3), v_max3, DPP moves, carry ops,
//     anything with a scalar-register operand ~1.8; v_readlane with a scalar lane select 3.2; scalar ALU ops 2 (on their own
//     port).  The kernel is bound by vector issue, so the band step is written against that table.
//   * Candidates are evaluated as the reference does: fp32 cell + fp64 transition constant + fp32 emission in
//     fp64, rounded to fp32, compared in fp32, later candidate wins ties (raw_loader.cpp:259-274).
//   * The trace is 2 bits per cell (bit 1: the cell equals its `left` candidate, bit 0: its `up` candidate; left wins): each
//     lane shifts the four bits of its two slots into one register per band -- the compare masks enter as the carry of
//     v_addc(t, t) -- and stores one dword per 8 bands (coalesced 256 B = 32 bytes per band, vs 100 bytes in the reference);
//     unfilled cells read back as FROM_D exactly like the reference's zero-initialised trace.
//   * Back-track: the walk state is scalar; trace groups are prefetched NP_BT_DEPTH deep into registers, each step is one
//     v_readlane + bit tests.  Pairs are collected 64 at a time and stored coalesced; their emissions are summed for the QC
//     after the walk (the reference's summation order, raw_loader.cpp:338-341, when the decision is close).
#include "np_kernels.h"

#define NP_ALIGN_BLOCK 256
#define NP_RING 128
#ifndef NP_BT_DEPTH
#define NP_BT_DEPTH 8     // back-track: trace groups (8 bands each) requested ahead of the walk.  Re-measured with round 3's shorter step:
                          // 4 -> 41.2 ms, 8 -> 40.8 ms per 32768 reads.  NOT more than the register budget holds: the queue's loads are
                          // issued and awaited by hand (asm), which is only sound while every queue entry stays in the register the asm
                          // wrote -- at 12 the compiler spills queue entries (it may copy an asm output right away: the load has not
                          // landed) and the kernel faults
#endif
#ifndef NP_A_WAVES
#define NP_A_WAVES 8      // resident waves per SIMD the register budget is set for
#endif
// timing experiments only (results are WRONG with any bit set): 1 no trace, 2 fp32 candidate sums, 4 fixed move pattern (no band-end
// reads, no scalar decision chain), 16 trivial emission, 128 no back-track
#ifndef NP_ABL
#define NP_ABL 0
#endif
#ifndef NP_A_WALK_PRIO
#define NP_A_WALK_PRIO 3  // wave priority during the back-track (s_setprio): a dependent scalar chain that needs few issue slots but holds a
#endif                    // wave slot for as long as it takes; 45.6 -> 43.2 ms per 32768 reads
#ifndef NP_A_M0SEL
#define NP_A_M0SEL 0      // round 6 experiment: the pair loop keeps the lane select of slot register 1's band end in M0 (v_readlane with an SGPR
#endif                    // lane select issues in 8.5 cycles, with a constant in 4.5: is M0 the cheap form?)
#ifndef NP_A_FILL_PRIO
#define NP_A_FILL_PRIO 1  // ... and, lower, while a band's move decision (v_readlane -> scalar compare -> branch) is in flight: the wave cannot
#endif                    // issue the next band before it resolves; 43.8 -> 42.6 ms

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

// The ring covers the k-mers [base, base + 127], base = 8 floor(llk / 8) - 8: at least 8 below the window [llk, llk + 99]
// and at least 13 above it.  It moves up by 8 whenever llk reaches a multiple of 8 (retarget below).
__device__ __forceinline__ int ring_base(int llk) { return (llk & ~7) - 8; }
// k-mer that ring slot `slot` holds while the window starts at llk
__device__ __forceinline__ int ring_kmer(int slot, int llk)
{
    const int base = ring_base(llk);
    return base + ((slot - base) & (NP_RING - 1));
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
__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float4 load_kp(const float4* __restrict__ kp, int k, int K) { return kp[clampi(k, 0, K - 1)]; }

// Everything the fill carries from band to band.
struct fill_t {
    int llk;                // band_lower_left[b].kmer_idx (wave-uniform)
    int eo0, eo1;           // byte offset, in the read's event means, of the event the NEXT band pairs with the k-mer of this
                            // lane's slot 2l / 2l+1: on entry to band b it is 4 (b - 1 - k)
    f4 g0, g1;              // scaled Gaussian of the two slots' k-mers: mean, -stdv, log-constant, 1/stdv (a register quad
                            // each: a re-target loads it in place)
    float p0, p1;           // band b-1
    double d0, d1;          // band b-2 moved on by one slot, as the doubles band b-1 made for its `left` candidates
    float best; int best_e; // end-cell search (:309-324): wave-uniform
    uint32_t tacc;          // trace bits of the last (up to) 8 bands, 4 bits per band, newest in the low nibble
    uint64_t vm0, vm1;      // FAST phase: lanes whose slot (2l, 2l+1) is inside the window (wave-uniform lane masks)
    int sel0, sel1, swp;    // FAST phase: the lanes of slot register 0 / 1 that hold a band end; swp: register 1 holds the FIRST cell
};

struct read_t {
    int E, K, lane, lane4, end_slot;
    bool nonpos;                    // every emission constant of the read is <= 0, hence every DP cell is (see the prologue)
    __amdgpu_buffer_rsrc_t ev;      // event means of the read, E * 4 bytes
    __amdgpu_buffer_rsrc_t kp;      // this wave's k-mer parameter slab, K * 16 bytes
    i4 kpd;                         // the same descriptor as four scalars (for the loads written in assembly)
    __amdgpu_buffer_rsrc_t tr;      // this wave's trace slab (a buffer store, not a flat one: a flat access in flight makes hipcc
                                    // wait for EVERY outstanding load at the next use of any of them)
    double lp_skip, lp_stay, lp_step, lp_trim;
};

// bit pattern of ring slot s (uniform) of a band held as (r0 = even slots, r1 = odd slots); all scalar
__device__ __forceinline__ int ring_read_bits(float r0, float r1, int s)
{
    int a = __builtin_amdgcn_readlane(__builtin_bit_cast(int, r0), (s >> 1) & 63);
    int b = __builtin_amdgcn_readlane(__builtin_bit_cast(int, r1), (s >> 1) & 63);
    asm("" : "+s"(a)); asm("" : "+s"(b));
    return (s & 1) ? b : a;
}
__device__ __forceinline__ float ring_read(float r0, float r1, int s) { return __builtin_bit_cast(float, ring_read_bits(r0, r1, s)); }
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
// FAST-phase lane masks and band-end lanes for a window that starts at llk: 50 even and 50 odd slots, each a run of 50
// lanes (mod 64) that starts at lane ceil(llk / 2) resp. floor(llk / 2)
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n) { n &= 63; return n ? (x << n) | (x >> (64 - n)) : x; }
__device__ __forceinline__ void window_state(fill_t& F)
{
    const uint64_t m50 = (1ull << (NP_ALN_BANDWIDTH / 2)) - 1ull;
    F.vm0 = rotl64(m50, (F.llk + 1) >> 1);
    F.vm1 = rotl64(m50, F.llk >> 1);
    const int la = (F.llk >> 1) & 63, ua = ((F.llk + NP_ALN_BANDWIDTH - 1) >> 1) & 63;       // lanes of slots llk, llk + 99
    F.swp = F.llk & 1;
    F.sel0 = F.swp ? ua : la;
    F.sel1 = F.swp ? la : ua;
}

// FAST phase, right move: the window moves up by one slot.  The mask of the register that held its first slot rotates by one lane
// (llk even: the even register, whose mask then equals the odd register's rotated; llk odd: the odd register catches up), the
// band ends swap registers.  Written out for the scalar unit, every register updated in place: left to the compiler, the
// renamed values cost three s_mov on the path WITHOUT a right move and four more at the loop's back edge.
__device__ __forceinline__ void right_move_fast(fill_t& F)
{
    uint64_t rot, tmp; int t;
    asm("s_lshl_b64 %[rot], %[m0], 1\n\t"
        "s_lshr_b64 %[tmp], %[m0], 63\n\t"
        "s_or_b64 %[rot], %[rot], %[tmp]\n\t"
        "s_bitcmp1_b32 %[llk], 0\n\t"
        "s_cselect_b64 %[m1], %[m0], %[m1]\n\t"
        "s_cselect_b64 %[m0], %[m0], %[rot]\n\t"
        "s_add_i32 %[llk], %[llk], 1\n\t"
        "s_add_i32 %[t], %[s1], 1\n\t"
        "s_mov_b32 %[s1], %[s0]\n\t"
        "s_mov_b32 %[s0], %[t]\n\t"
#if NP_A_M0SEL
        "s_mov_b32 m0, %[s1]\n\t"
#endif
        "s_xor_b32 %[sw], %[sw], 1"
        : [rot] "=&s"(rot), [tmp] "=&s"(tmp), [t] "=&s"(t), [m0] "+s"(F.vm0), [m1] "+s"(F.vm1), [llk] "+s"(F.llk),
          [s0] "+s"(F.sel0), [s1] "+s"(F.sel1), [sw] "+s"(F.swp)
        : : "scc");
}

// The window's first k-mer has just reached a multiple of 8 (F.llk): the eight slots that hold the k-mers llk-16 .. llk-9 --
// both slots of the four lanes (slot >> 1) -- move on by 128 k-mers and load their new records.  kref16: 16 k = kref16 - 4 eo
// for the eo values as they are at the call.
__device__ __forceinline__ void retarget(fill_t& F, const read_t& R, const int kref16)
{
    const int grp = ((F.llk - 16) & (NP_RING - 1)) >> 3;
    const int base = kref16 + 16 * NP_RING;
    uint64_t save; int t0, t1;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_eq_u32_e32 %[grp], %[l4]\n\t"
                 "v_lshlrev_b32 %[t0], 2, %[e0]\n\t"
                 "v_lshlrev_b32 %[t1], 2, %[e1]\n\t"
                 "v_sub_u32 %[t0], %[base], %[t0]\n\t"
                 "v_sub_u32 %[t1], %[base], %[t1]\n\t"
                 "v_subrev_u32 %[e0], 0x200, %[e0]\n\t"
                 "v_subrev_u32 %[e1], 0x200, %[e1]\n\t"
                 "buffer_load_dwordx4 %[g0], %[t0], %[rs], 0 offen\n\t"
                 "buffer_load_dwordx4 %[g1], %[t1], %[rs], 0 offen\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(save), [t0] "=&v"(t0), [t1] "=&v"(t1), [e0] "+v"(F.eo0), [e1] "+v"(F.eo1), [g0] "+v"(F.g0), [g1] "+v"(F.g1)
                 : [grp] "s"(grp), [l4] "v"(R.lane4), [base] "s"(base), [rs] "s"(R.kpd)
                 : "vcc", "memory");
}

// One band.  (x0, x1): event means of this band's two cells on entry (loaded during the previous band), of the next
// band's on exit.  TRIM: the window may still contain k-mer -1.  END: the window may contain k-mer K-1.
// FAST: the middle of the read -- every cell of the window [llk, llk+99] exists (0 <= llk, llk+99 < K-1, and the events
// of the window's first and last k-mer are inside [0, E)), so a slot is valid iff it is inside the window, which only
// changes on a right move: the lane masks F.vm0/F.vm1 replace the per-band geometry.
// POS: 0 / 1 = first / second band of the FAST loop's pair (b - POS is even; the event offsets advance once per pair and the
// band's share is the load's immediate), -1 = stand-alone.
template <bool TRIM, bool END, bool FAST, int POS = -1>
__device__ __forceinline__ void band_step(fill_t& F, const read_t& R, const int b, float& x0, float& x1, float& n0, float& n1)
{
    const int lane = R.lane, E = R.E, K = R.K;
    const int llk = F.llk;
    if (POS >= 0) {
        // pair loop: the next band's event means (same k-mer, next event) are requested first, into the other pair of registers,
        // then ONE wait covers this band's two (requested a band ago; the new requests and a trace store may stay in flight).
        // (the register part of the offset is >= 4 for every in-window slot of a FAST band; a slot outside the window may see its
        //  sum misjudged by the range check -- it only feeds masked cells)
        n0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R.ev, F.eo0 + 4 * POS, 0, 0));
        n1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R.ev, F.eo1 + 4 * POS, 0, 0));
        asm volatile("" : "+v"(x0), "+v"(x1));
#if NP_A_FILL_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    // left sources: an odd slot's is the lane's own even slot, an even slot's the previous lane's odd slot
    const float l0 = wave_ror1_all(F.p1);

    // A slot holds a cell of this band iff its k-mer is inside the window and its event e = b-2-k exists, i.e. iff
    // e is in [max(0, b-2-khi), min(E-1, b-2-klo)]: one unsigned range test on eo = 4 (e + 1)
    const int klo = llk > 0 ? llk : 0;
    const int khi = (llk + NP_ALN_BANDWIDTH - 1) < (K - 1) ? (llk + NP_ALN_BANDWIDTH - 1) : (K - 1);
    const int eo1_in = F.eo1;
    bool v0 = false, v1 = false;
    if (!FAST) {
        const int elo = (b - 2 - khi) > 0 ? (b - 2 - khi) : 0;
        const int ehi = (b - 2 - klo) < (E - 1) ? (b - 2 - klo) : (E - 1);
        const int cnt = ehi - elo + 1 > 0 ? ehi - elo + 1 : 0;
        const int tb = pin_s(4 * (elo + 1));
        v0 = (uint32_t)(F.eo0 - tb) < (uint32_t)(4 * cnt);
        v1 = (uint32_t)(F.eo1 - tb) < (uint32_t)(4 * cnt);
    }

    // emissions of both cells: np_emission / np_div_exact, operation for operation (v_pk_*_f32 would halve the
    // instruction count but not the issue cycles -- tools/valu_rates.hip -- and forces the parameters into register pairs)
#if NP_ABL & 16
    const float emx = x0 * F.g0.z, emy = x1 * F.g1.z;
#else
    const float emx = np_emission_nd(x0, F.g0.x, F.g0.y, F.g0.z, F.g0.w);
    const float emy = np_emission_nd(x1, F.g1.x, F.g1.y, F.g1.z, F.g1.w);
#endif
    // stand-alone step: the next band's event means go into the registers the emissions have just released: the loop-carried
    // value is the load's own destination, so nothing is copied (a copy would have to wait for the load)
    if (POS < 0) {
        n0 = buf_f32(R.ev, F.eo0); n1 = buf_f32(R.ev, F.eo1);
        F.eo0 += 4; F.eo1 += 4;
    }

    // DP cells (raw_loader.cpp:240-289), computed unconditionally and masked: fp32 cell + fp64 constant + fp32 emission
    // in fp64, rounded to fp32; max, then FROM_U / FROM_L override on equality in that order (later candidate wins).
    // (double)p0 is `up` of the even slot AND `left` of the odd one; the two `left` doubles are the next band's diagonals.
#if NP_ABL & 2
    const double P0 = (double)F.p0, L0 = (double)l0;
    const float sd0 = (float)F.d0 + (float)R.lp_step + emx, sd1 = (float)F.d1 + (float)R.lp_step + emy;
    const float su0 = F.p0 + (float)R.lp_stay + emx, su1 = F.p1 + (float)R.lp_stay + emy;
    const float sl0 = l0 + (float)R.lp_skip, sl1 = F.p0 + (float)R.lp_skip;
#else
    const double em0 = (double)emx, em1 = (double)emy;
    const double P0 = (double)F.p0, P1 = (double)F.p1, L0 = (double)l0;
    const float sd0 = (float)(F.d0 + R.lp_step + em0), sd1 = (float)(F.d1 + R.lp_step + em1);
    const float su0 = (float)(P0 + R.lp_stay + em0), su1 = (float)(P1 + R.lp_stay + em1);
#if NP_ABL & 32
    const float sl0 = NP_NEG_INF, sl1 = NP_NEG_INF;      // timing experiment only (results WRONG): the left (k-mer skip) candidates for free
#else
    const float sl0 = (float)(L0 + R.lp_skip), sl1 = (float)(P0 + R.lp_skip);
#endif
#endif
    const float m0 = __builtin_fmaxf(__builtin_fmaxf(sd0, su0), sl0);
    const float m1 = __builtin_fmaxf(__builtin_fmaxf(sd1, su1), sl1);

    // FAST: the band ends of Suzuki's rule are read out (v_readlane -> scalar compares) BEFORE the trace bits are packed,
    // so that the scalar chain of the move decision runs while the vector unit packs.  (The ends are inside the window:
    // the unmasked maxima are the cells.)
    int xs = 0, ys = 0;
    if (FAST && !(NP_ABL & 4)) {
#if NP_A_FILL_PRIO
        if (POS >= 0) __builtin_amdgcn_s_setprio(NP_A_FILL_PRIO);
#endif
        xs = __builtin_amdgcn_readlane(__builtin_bit_cast(int, m0), F.sel0);
#if NP_A_M0SEL
        if (POS >= 0) asm volatile("v_readlane_b32 %0, %1, m0" : "=s"(ys) : "v"(m1));
        else
#endif
        ys = __builtin_amdgcn_readlane(__builtin_bit_cast(int, m1), F.sel1);
        asm volatile("" : "+s"(xs), "+s"(ys));
    }

    // Trace: per slot bit 1 = "the cell is its left candidate", bit 0 = "the cell is its up candidate" (left wins: FROM_L = 2,
    // FROM_U = 1, FROM_D = 0 in the reference's numbering, and the pattern 3 reads as FROM_L); odd slot in bits 3..2, even
    // slot in bits 1..0 of the band's nibble.  The bits of a slot outside the band are never read back: the walk only
    // visits finite cells, whose best predecessor is finite, hence inside its band.
    if (FAST) {
        // A vector compare's mask cannot be used by the very next vector instructions (hipcc funnels compares through VCC and
        // pads with s_nop): the four compares write four scalar pairs, and by the time an add-with-carry shifts a mask into the
        // accumulator three other instructions have issued.
        uint64_t qa, qb, qc, qd, qj;
#if !(NP_ABL & 1)
        asm("v_cmp_eq_f32_e64 %[a], %[m1], %[l1]\n\t"
            "v_cmp_eq_f32_e64 %[b], %[m1], %[u1]\n\t"
            "v_cmp_eq_f32_e64 %[c], %[m0], %[l0]\n\t"
            "v_cmp_eq_f32_e64 %[d], %[m0], %[u0]\n\t"
            "v_addc_co_u32_e64 %[t], %[j], %[t], %[t], %[a]\n\t"
            "v_addc_co_u32_e64 %[t], %[j], %[t], %[t], %[b]\n\t"
            "v_addc_co_u32_e64 %[t], %[j], %[t], %[t], %[c]\n\t"
            "v_addc_co_u32_e64 %[t], %[j], %[t], %[t], %[d]"
            : [t] "+v"(F.tacc), [a] "=&s"(qa), [b] "=&s"(qb), [c] "=&s"(qc), [d] "=&s"(qd), [j] "=&s"(qj)
            : [m0] "v"(m0), [u0] "v"(su0), [l0] "v"(sl0), [m1] "v"(m1), [u1] "v"(su1), [l1] "v"(sl1));
#endif
    }
    uint32_t f0 = 0u, f1 = 0u;
    if (!FAST) {
        f0 = (m0 == sl0) ? 2u : ((m0 == su0) ? 1u : 0u);
        f1 = (m1 == sl1) ? 2u : ((m1 == su1) ? 1u : 0u);
    }
    float c0, c1;
    if (FAST) { c0 = sel_mask(F.vm0, m0, NP_NEG_INF); c1 = sel_mask(F.vm1, m1, NP_NEG_INF); }
    else { c0 = v0 ? m0 : NP_NEG_INF; c1 = v1 ? m1 : NP_NEG_INF; }

    if (TRIM && llk <= -1) {
        // the window still contains k-mer -1: start cell of band 0 (:152-157) and the trim column (:216-225).
        // k = -1 lives in ring slot 127 (lane 63, odd slot); its event is b - 1.
        const int et = b - 1;
        if (lane == 63 && eo1_in == 4 * b) {
            if (et == -1) { c1 = 0.0f; f1 = 0u; }
            else if (et >= 0 && et < E) { c1 = (float)(R.lp_trim * (double)(et + 1)); f1 = 1u; }
            else { c1 = NP_NEG_INF; f1 = 0u; }
        }
    }

    // packed trace: every lane keeps the bits of its two slots, 4 bits per band, and stores one dword per 8 bands
    // (64 lanes x 4 B = 256 B coalesced = 32 B/band).  The back-track reads the word back into the SAME lane.
    // (the word shifts up one nibble per band, so after 8 bands band b%8 == 0 sits in bits 31..28: no variable shift)
#if !(NP_ABL & 1)
    if (!FAST) F.tacc = (F.tacc << 4) | (f1 << 2) | f0;
    if (POS == 1 ? (b & 7) == 7 : (POS < 0 && (b & 7) == 7))
        __builtin_amdgcn_raw_buffer_store_b32((int)F.tacc, R.tr, 4 * lane, (pin_s(b) >> 3) * 256, 0);    // (pin_s: no second induction variable)
#endif

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
    F.d0 = L0; F.d1 = P0;
    F.p0 = c0; F.p1 = c1;

    if (FAST && POS >= 0 && !(NP_ABL & 4)) {
        // Suzuki's rule (see below) in the pair loop, where the band's parity is known.  ll < ur for floats that are +0, negative
        // or -inf (R.nonpos) is bits(ll) > bits(ur) as unsigned integers; "both ends -inf: alternate, right on even bands" needs
        // nothing on odd bands (equal patterns compare false) and on even bands is the same compare against min(ur, bits(-FLT_MAX)):
        // -inf has the largest pattern, so ll > that is "ll is -inf" exactly when ur is -inf, and unchanged otherwise.
        // The compare's SCC feeds the branch directly.
        // (plain integer code on scalar values: hipcc emits s_cselect / s_min_u32 / s_cmp_gt_u32 + s_cbranch_scc -- the compare's SCC
        //  feeds the branch directly.  `asm goto` is not an option: this compiler drops the statement.)
        const uint32_t xu = (uint32_t)pin_s(xs), yu = (uint32_t)pin_s(ys);
        const bool sw = F.swp != 0;
        const uint32_t ll = sw ? yu : xu;
        uint32_t ur = sw ? xu : yu;
        if (POS == 0) ur = ur < 0xff7fffffu ? ur : 0xff7fffffu;
        if (ll > ur) {
            right_move_fast(F);
            if ((F.llk & 7) == 0) retarget(F, R, 16 * (pin_s(b) - POS - 1));
        }
        return;
    }
    if (FAST || b >= 1) {
        // Suzuki's rule for band b+1, on this band (:179-195): both ends -inf (the AND of two non-NaN patterns is -inf's only
        // then): alternate; else right iff ll < ur, where a single -inf compares as the reference's
        // is_offset_valid ? value : -INFINITY does
        bool right;
        if (FAST && (NP_ABL & 4)) {
            right = (((b + 1) * 13) >> 5) != ((b * 13) >> 5);       // a fixed pattern with the same share of right moves
        } else if (FAST) {
            // FAST runs only for reads whose cells are all <= 0 (R.nonpos): for such floats (+0, negative, -inf) x < y is
            // bits(x) > bits(y) as unsigned integers -- scalar compares instead of a vector compare on two scalars
            // (spelled out for the scalar unit: hipcc lowers a select between wave-uniform conditions to vector code)
            int r, t, ll, ur;
#define NP_SUZUKI(PAR_CONSTRAINT, PAR)                                                                                           \
            asm("s_cmp_lg_u32 %[sw], 0\n\t"                                                                                       \
                "s_cselect_b32 %[ll], %[y], %[x]\n\t"                                                                             \
                "s_cselect_b32 %[ur], %[x], %[y]\n\t"                                                                             \
                "s_and_b32 %[t], %[ll], %[ur]\n\t"                                                                                \
                "s_cmp_gt_u32 %[ll], %[ur]\n\t"                                                                                   \
                "s_cselect_b32 %[r], 1, 0\n\t"                                                                                    \
                "s_cmp_eq_u32 %[t], 0xff800000\n\t"                                                                               \
                "s_cselect_b32 %[r], %[par], %[r]"                                                                                \
                : [r] "=&s"(r), [t] "=&s"(t), [ll] "=&s"(ll), [ur] "=&s"(ur)                                                      \
                : [x] "s"(xs), [y] "s"(ys), [sw] "s"(F.swp), [par] PAR_CONSTRAINT(PAR) : "scc")
            // both ends outside: alternate, starting with a right move on even bands (the band's parity is known in the pair loop)
            { const int par = (b & 1) ^ 1; NP_SUZUKI("s", par); }
#undef NP_SUZUKI
            right = r != 0;
        } else {
            const int ll = pin_s(ring_read_bits(c0, c1, llk & (NP_RING - 1)));
            const int ur = pin_s(ring_read_bits(c0, c1, (llk + NP_ALN_BANDWIDTH - 1) & (NP_RING - 1)));
            const bool both_ob = (ll & ur) == (int)0xff800000;
            right = both_ob ? ((b & 1) == 0) : (__builtin_bit_cast(float, ll) < __builtin_bit_cast(float, ur));
        }
        if (right) {
            F.llk = llk + 1;
            if (FAST) {
                // the window moves up by one slot: the mask of the register that held its first slot rotates by one lane (llk even:
                // the even register, whose mask then equals the odd register's rotated; llk odd: the odd register catches up);
                // the band ends swap registers
                const uint64_t rot = (F.vm0 << 1) | (F.vm0 >> 63);
                const bool odd = (llk & 1) != 0;
                F.vm1 = odd ? F.vm0 : F.vm1;
                F.vm0 = odd ? F.vm0 : rot;
                const int s0 = F.sel0;
                F.sel0 = F.sel1 + 1; F.sel1 = s0; F.swp ^= 1;
            }
            if ((F.llk & 7) == 0) retarget(F, R, 16 * b);
        }
    }
}

// fill and back-track of a read by the same wave (trace scratch per resident wave)
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

        const int n_rows = (n_bands + 7) >> 3;
        const bool ok = !(E <= 0 || K <= 0 || (uint64_t)n_rows * 32 > a.trace_stride || (uint64_t)K > a.kp_stride || cap < E + K + 2);
        int n_out = 0, max_gap = 0, last_k = -1;
        double sum_emission = 0.0;
        float best_u = NP_NEG_INF;
        int curr_e = 0;
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

            {
                // ---------------- fill ----------------
                read_t R;
                R.E = E; R.K = K; R.lane = lane; R.lane4 = lane >> 2; R.end_slot = (K - 1) & (NP_RING - 1); R.nonpos = nonpos;
                R.ev = make_rsrc(ev, (uint32_t)E * 4u); R.kp = make_rsrc(kp, (uint32_t)K * 16u); R.tr = make_rsrc(trace, (uint32_t)n_rows * 256u);
                {
                    const uint64_t u = (uint64_t)uniform_ptr(kp);
                    R.kpd = i4{(int)(uint32_t)u, (int)(uint32_t)(u >> 32), (int)((uint32_t)K * 16u), 0x00020000};      // == make_rsrc(kp, 16 K)
                }

                // wave-uniform constants in scalar registers (a VALU fp64 add takes one SGPR-pair operand): 8 VGPRs saved
                R.lp_skip = uniform_f64(rd->lp_skip); R.lp_stay = uniform_f64(rd->lp_stay);
                R.lp_step = uniform_f64(rd->lp_step); R.lp_trim = uniform_f64(rd->lp_trim);
                fill_t F;
                F.llk = -1 - NP_ALN_BANDWIDTH / 2;              // band_lower_left[0].kmer_idx, raw_loader.cpp:150-151
                {
                    const int k0 = ring_kmer(2 * lane, F.llk), k1 = ring_kmer(2 * lane + 1, F.llk);
                    F.eo0 = 4 * (-1 - k0); F.eo1 = 4 * (-1 - k1);                    // band 0
                    const float4 g0 = buf_f32x4(R.kp, 16 * k0), g1 = buf_f32x4(R.kp, 16 * k1);
                    F.g0 = f4{g0.x, g0.y, g0.z, g0.w}; F.g1 = f4{g1.x, g1.y, g1.z, g1.w};
                }
                F.p0 = F.p1 = NP_NEG_INF; F.d0 = F.d1 = (double)NP_NEG_INF;
                F.best = NP_NEG_INF; F.best_e = 0; F.tacc = 0u;
                F.vm0 = F.vm1 = 0ull; F.sel0 = F.sel1 = F.swp = 0;
                float x0 = 0.0f, x1 = 0.0f;                     // event means of the current band's two cells
                int b = 0;
                // Three phases, so that the long middle of the read pays for no band geometry, trim column or end search.
                // llk and u = b-2-llk (the event of the window's first k-mer) never decrease and exactly one of them grows
                // per band, so the FAST conditions
                //   llk >= 0,  u >= 99  (the window's last k-mer has event u-99 >= 0)  -- become true once, and
                //   llk + 99 < K-1,  u <= E-1                                           -- become false once;
                // and while min(K-2 - (llk+99), E-1 - u) = s > 0 the next s bands are FAST whatever the moves are.
                for (; b < n_bands && !(F.llk >= 0 && b - 2 - F.llk >= NP_ALN_BANDWIDTH - 1); ++b)
                    band_step<true, true, false>(F, R, b, x0, x1, x0, x1);
                window_state(F);
                for (; R.nonpos;) {
                    const int ks = (K - 2) - (F.llk + NP_ALN_BANDWIDTH - 1), es = (E - 1) - (b - 2 - F.llk);
                    int stop = b + (ks < es ? ks : es);
                    stop = stop < n_bands ? stop : n_bands;
                    if (stop <= b) break;
                    // two bands per iteration, the first one even: the doubles made of `left` become the next band's diagonal without
                    // a register copy, the event offsets advance once, the trace store and the parity of Suzuki's tie rule are
                    // compile-time properties of the position
                    if (b & 1) { band_step<false, false, true>(F, R, b, x0, x1, x0, x1); ++b; }
                    float y0, y1;
#if NP_A_M0SEL
                    asm volatile("s_mov_b32 m0, %0" : : "s"(F.sel1));
#endif
                    for (; b + 1 < stop; b += 2) {
                        band_step<false, false, true, 0>(F, R, b, x0, x1, y0, y1); band_step<false, false, true, 1>(F, R, b + 1, y0, y1, x0, x1);
                        F.eo0 += 8; F.eo1 += 8;
                    }
                    for (; b < stop; ++b) band_step<false, false, true>(F, R, b, x0, x1, x0, x1);
                }
                for (; b < n_bands; ++b) band_step<true, true, false>(F, R, b, x0, x1, x0, x1);
                if ((n_bands & 7) != 0) __builtin_amdgcn_raw_buffer_store_b32((int)(F.tacc << (4 * (8 - (n_bands & 7)))), R.tr, 4 * lane, ((n_bands - 1) >> 3) * 256, 0);   // last, partial group

                best_u = F.best; curr_e = F.best_e;
            }

            // ---------------- backtrack (:326-361) + QC sums (:338-341) ----------------
            int curr_k = K - 1;

            // the trace was written by lanes 0..3 of this wave: complete the stores before other lanes read them back
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);

            if (best_u != NP_NEG_INF && !(NP_ABL & 128)) {
#if NP_A_WALK_PRIO
                __builtin_amdgcn_s_setprio(NP_A_WALK_PRIO);
#endif
                // Scalar walk.  The kernel is instruction-issue bound (~2.3 cycles per wave-instruction of any kind, measured
                // with tools/align_variants.sh probes), and a walk of ~0.63 steps per band was a fifth of its instructions, so
                // the step is written out by hand: 14 instructions since round 3 (12 scalar + v_readlane + v_writelane; 21 before).
                //   * A step reads the 2 bits of its cell out of the lane that owns the k-mer's ring slot (v_readlane) and
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
                int j = 0, gap = 0;
                int k0 = curr_k, e0 = curr_e;          // position of the chunk's first step
                int vfrom = 0;                         // lane j: the code of step j of the chunk
                // Round 3: the step loop only leaves at the end of a trace group -- it tests neither "chunk full" nor "an index went
                // negative" per step (14 scalar instructions per step instead of 21, one branch instead of three: the walk is bound
                // by the CU's scalar-instruction throughput, profiles/r03_kernel_a_split.md).  A group is 8 bands and a step leaves at
                // least one, so a chunk that enters a group with at most 56 codes cannot overflow its 64 lanes; and a walk that runs
                // off the lattice inside a group (k or e below 0) only gathers a few garbage codes until the group ends: the flush
                // keeps the steps whose position is on the lattice -- a prefix, both indices only decrease.
                // up to 64 pairs: stored at descending addresses (their emissions are added up after the walk)
                auto flush = [&]() {
                    const bool in = lane < j;
                    const uint64_t mk = __builtin_amdgcn_ballot_w64(in && vfrom != 1);          // FROM_D, FROM_L (2, 3): k drops
                    const uint64_t me = __builtin_amdgcn_ballot_w64(in && vfrom < 2);           // FROM_D, FROM_U: e drops
                    np_pair p;
                    p.ref_pos = k0 - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    p.read_pos = e0 - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(me >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)me, 0u));
                    const bool valid = in && (p.ref_pos | p.read_pos) >= 0;
                    const int jv = __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid));     // (the valid steps are steps 0 .. jv-1)
                    if (valid) pairs[cap - 1 - (n_out + lane)] = p;
                    if (jv > 0) {
                        last_k = __builtin_amdgcn_readlane(p.ref_pos, jv - 1);                                   // the last recorded pair's k-mer
                        // runs of FROM_L steps: ml has bit i set iff step i is FROM_L (the steps of the chunk are bits 0 .. jv-1)
                        const uint64_t all = jv >= 64 ? ~0ull : ((1ull << jv) - 1ull);
                        const uint64_t ml = all & ~me;
                        const int lead = ml == all ? jv : __builtin_ctzll(~ml);                     // run that continues the previous chunk's
                        int longest = 0;
                        for (uint64_t m = ml; m != 0; m &= m << 1) longest++;                      // longest run inside the chunk
                        longest = longest > gap + lead ? longest : gap + lead;
                        max_gap = longest > max_gap ? longest : max_gap;
                        // run still open at the chunk's end
                        if (ml == all) gap += jv;
                        else gap = __builtin_clzll(~(ml << (64 - jv)));                              // (bits below 64 - jv are set in the operand)
                    }
                    n_out += jv; j = 0; k0 = curr_k; e0 = curr_e;
                };
                int done = 0;
                int nib = 8 * cg + 5 - (curr_k + curr_e);        // nibble of the current band inside its group's words, counted from the top
                while (!done) {
#pragma unroll
                    for (int pos = 0; pos < NP_BT_DEPTH; ++pos) {
                        asm volatile("s_waitcnt vmcnt(%[n])" : [t] "+v"(tq[pos]) : [n] "n"(NP_BT_DEPTH - 1) : "memory");
                        // steps inside trace group cg (band 8 cg in bits 31..28 of the lanes' words: nibble 7); the loop is entered with
                        // nib 0 or 1 (the step that left the group above crossed one or two bands) and runs while nib <= 7
                        int t_, c_, w_, m0s_, from_;
                        // (the step counter j lives in M0 inside the loop: v_writelane takes its lane select from M0, because a
                        //  second scalar register next to the data operand would exceed the constant-bus limit.  M0 is saved and
                        //  restored around the block -- two scalar moves per trace group -- instead of being declared clobbered:
                        //  the compiler treats it as reserved and makes no promise about a clobbered reserved register)
                        asm volatile("s_mov_b32 %[m0s], m0\n\t"
                                     "s_mov_b32 m0, %[j]\n\t"
                                     "1:\n\t"
                                     "s_and_b32 %[c], %[k], 1\n\t"                   // odd slot: bits 3..2 of the nibble
                                     "s_lshl1_add_u32 %[c], %[c], 0x20000\n\t"        // field width 2 | 2 * bit
                                     "s_lshl2_add_u32 %[c], %[nib], %[c]\n\t"         // + 4 * nibble
                                     "s_lshr_b32 %[t], %[k], 1\n\t"                   // the slot's lane (the lane select is taken modulo 64)
                                     "v_readlane_b32 %[w], %[wreg], %[t]\n\t"
                                     "s_bfe_u32 %[from], %[w], %[c]\n\t"
                                     "v_writelane_b32 %[vf], %[from], m0\n\t"
                                     "s_add_i32 m0, m0, 1\n\t"
                                     "s_cmp_lg_u32 %[from], 1\n\t"
                                     "s_subb_u32 %[k], %[k], 0\n\t"                   // k -= (from != FROM_U)
                                     "s_cmp_eq_u32 %[from], 0\n\t"
                                     "s_addc_u32 %[nib], %[nib], 1\n\t"               // one band down, two on FROM_D (the only code that moves k AND e)
                                     "s_cmp_le_i32 %[nib], 7\n\t"
                                     "s_cbranch_scc1 1b\n\t"
                                     "s_mov_b32 %[j], m0\n\t"
                                     "s_mov_b32 m0, %[m0s]"
                                     : [k] "+s"(curr_k), [nib] "+s"(nib), [j] "+s"(j), [vf] "+v"(vfrom),
                                       [from] "=&s"(from_), [t] "=&s"(t_), [c] "=&s"(c_), [w] "=&s"(w_), [m0s] "=&s"(m0s_)
                                     : [wreg] "v"(tq[pos])
                                     : "scc");
                        curr_e = 8 * cg + 5 - nib - curr_k;                                // band = k + e + 2
                        done = (curr_k | curr_e) >> 31;                                 // -1 once either index is negative
                        if (j > 56 || done) flush();
                        if (done) break;
                        nib -= 8;
                        NP_BT_LOAD(tq[pos], cg - NP_BT_DEPTH);
                        cg -= 1;
                    }
                }
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
                __builtin_amdgcn_s_setprio(0);
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
// (workgroup-private counts in LDS first: a batch of equally long reads hits ONE bucket, and 100 000 atomics on one address
//  take most of a millisecond)
__global__ void __launch_bounds__(256) np_align_hist_kernel(int n_reads, const np_read_dev* __restrict__ reads, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[NP_ORDER_BUCKETS];
    for (int i = threadIdx.x; i < NP_ORDER_BUCKETS; i += 256) h[i] = 0;
    __syncthreads();
    for (int r = blockIdx.x * 1024 + threadIdx.x; r < n_reads && r < (blockIdx.x + 1) * 1024; r += 256) atomicAdd(&h[order_bucket(reads[r])], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < NP_ORDER_BUCKETS; i += 256) if (h[i]) atomicAdd(&hist[i], h[i]);
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
    __shared__ uint32_t h[NP_ORDER_BUCKETS];       // workgroup-private counts, then the workgroup's base in every bucket
    for (int i = threadIdx.x; i < NP_ORDER_BUCKETS; i += 256) h[i] = 0;
    __syncthreads();
    int bk[4]; uint32_t local[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = blockIdx.x * 1024 + t * 256 + threadIdx.x;
        bk[t] = -1; local[t] = 0;
        if (r < n_reads) { bk[t] = order_bucket(reads[r]); local[t] = atomicAdd(&h[bk[t]], 1u); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NP_ORDER_BUCKETS; i += 256) { const uint32_t n = h[i]; if (n) h[i] = atomicAdd(&cursor[i], n); }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (bk[t] >= 0) order[h[bk[t]] + local[t]] = (uint32_t)(blockIdx.x * 1024 + t * 256 + threadIdx.x);
}

} // namespace

int np_align_block_threads(void) { return NP_ALIGN_BLOCK; }

// scratch: 2 * NP_ORDER_BUCKETS counters (zeroed here) followed by n_reads order entries
hipError_t np_launch_align_order(int n_reads, const np_read_dev* reads, uint32_t* scratch, hipStream_t s)
{
    uint32_t* hist = scratch; uint32_t* cursor = scratch + NP_ORDER_BUCKETS; uint32_t* order = scratch + 2 * NP_ORDER_BUCKETS;
    hipError_t e = hipMemsetAsync(hist, 0, NP_ORDER_BUCKETS * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const int nb = (n_reads + 1023) / 1024;
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
