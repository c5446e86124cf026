// np_device.h -- shared device-side definitions for the gfx950 kernels.
//
// Numerics contract: every expression below keeps the reference's
// evaluation order and rounding points; the translation units are compiled with -ffp-contract=off so
// hipcc never fuses a*b+c (the reference x86-64 build has no FMA, Makefile:12-13 of the reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/np_hmm.h"

#define NP_LOGSUM_TBL 16000
#define NP_NEG_INF (-__builtin_inff())

// Pore-model state on the device: the three doubles the path reads, padded to 32 B so one state is
// two aligned 16-byte loads (PoreModelStateParams, src/pore_model/nanopolish_poremodel.h:20-36).
struct __attribute__((aligned(32))) np_state_dev {
    double level_mean, level_stdv, level_log_stdv, pad;
};

// p7_FLogsum (src/common/logsum.h:55-66) on an LDS-resident copy of flogsum_lookup.
//   reference: (min == -inf || max - min >= 15.7f) ? max : max + tbl[(int)((max - min) * 1000.f)]
// Branch-free and select-free form.  max - min == |a - b| exactly (a subtraction and its mirror round alike), so no
// minimum is formed.  The LDS copy of the table is ZERO from entry NP_LOGSUM_CUT = 15700 on (np_lse_load_table), and the
// index saturates at the last entry:
//   * |a-b| < 15.7f   <=>  trunc(|a-b| * 1000.f) <= 15699 (15.7f * 1000.f rounds to 15700.0f, its predecessor to 15699.99..),
//                          so exactly the reference's table cases read a non-zero entry, the same one;
//   * |a-b| >= 15.7f, or +inf (one operand -inf): a zero entry, max + 0.0f == max (scores are never -0.0f);
//   * NaN (both -inf): the conversion gives index 0, and -inf + tbl[0] == -inf.
// Two thirds of the log-sums of a forward pass have a -inf or far-away operand (counted on the oracle); those lanes all
// read the same last entry (one broadcast access).  What bounds the kernel is the gather of the remaining third: PMC
// shows the LDS pipe busy for the whole kernel at 4.2 cycles per ds_read, half of them bank-conflict cycles.
#define NP_LOGSUM_CUT 15700
#ifndef NP_LSE_FORM
#define NP_LSE_FORM 1
#endif
// float -> unsigned as the hardware instruction defines it: truncation, saturation (+inf -> 0xffffffff), NaN -> 0
__device__ __forceinline__ uint32_t np_cvt_u32_sat(float v)
{
    uint32_t r;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ float np_lse(float a, float b, const float* __restrict__ tbl)
{
    const float mx = __builtin_fmaxf(a, b);            // inputs are never NaN (finite or -inf): v_max is exact
    const float d = __builtin_fabsf(a - b);
    // byte offset of the entry without a shift: RN(d * 4000.f) == 4 * RN(d * 1000.f) (scaling by 4 is exact) and
    // trunc(4y) & ~3 == 4 * trunc(y); v_cvt_u32_f32 truncates, saturates (+inf -> 0xffffffff) and maps NaN to 0
    uint32_t off = np_cvt_u32_sat(d * 4000.f) & ~3u;
    off = off < 4u * (NP_LOGSUM_TBL - 1) ? off : 4u * (NP_LOGSUM_TBL - 1);
    return mx + *(const float*)((const char*)tbl + off);
}
// The same without the saturating minimum: the table is the kernel's ONLY LDS allocation (64000 bytes, a whole number of
// allocation granules), and an LDS read at or beyond the allocated size returns 0 (writes are dropped) -- the hardware's
// out-of-range rule for DS instructions.  So every offset past the table -- |a-b| >= 16 nats, or +inf when one operand is -inf
// (the conversion saturates to 0xffffffff) -- reads as the zero the saturated form fetched from the table's zeroed tail.
// One vector instruction less per log-sum (6 instead of 7); tbl_lds: the table's LDS address.
// Nothing here is left to the language's undefined corners: the float -> unsigned conversion is the hardware instruction itself
// (v_cvt_u32_f32 truncates, saturates and maps NaN to 0; a C cast of an out-of-range float is undefined), and the address is
// formed as an integer (no in-bounds pointer arithmetic the compiler could reason from).  The hardware side -- one LDS object of
// exactly NP_LOGSUM_TBL * 4 bytes per workgroup, reads past it returning 0 -- is checked at np_create (np_capi.hip:probe_hardware:
// hipFuncGetAttributes on every forward kernel and a probe kernel over dirtied LDS); if either check fails the context scores
// with the clamped np_lse instead.
static_assert((NP_LOGSUM_TBL * 4) % 1280 == 0 && (NP_LOGSUM_TBL * 4) % 512 == 0, "the log-sum table must be a whole number of LDS allocation granules");
// byte offset of the table entry of a log-sum: 4 * trunc(|a - b| * 1000.f) (see np_lse)
__device__ __forceinline__ uint32_t np_lse_offset(float a, float b)
{
#if NP_LSE_FORM == 0
    const float d = __builtin_fabsf(a - b);
    return np_cvt_u32_sat(d * 4000.f) & ~3u;
#else
    // |a - b| * 4000 == |(a - b) * 4000| (rounding to nearest is symmetric), and the conversion takes the absolute value as an operand
    // modifier: the multiplication is then a plain two-operand instruction with the constant as a literal (form 1) or in a vector
    // register (form 2).  With the modifier on the multiplication the compiler has to use the three-operand encoding, which takes no
    // literal on gfx9: it parks 4000.f in a SCALAR register, and a vector instruction with a scalar-register source issues in the slow
    // class (4.6 cycles against 2.6, tools/valu_rates l: "v_mul_f32 |v|, s" / "v_mul_f32 literal") -- once per log-sum, ~8 times per cell.
#if NP_LSE_FORM == 2
    float c4000; asm("v_mov_b32 %0, 0x457a0000" : "=v"(c4000));
    const float t = (a - b) * c4000;
#else
    const float t = (a - b) * 4000.f;
#endif
    uint32_t off;
    asm("v_cvt_u32_f32_e64 %0, |%1|" : "=v"(off) : "v"(t));
    return off & ~3u;
#endif
}
__device__ __forceinline__ float np_lse_oor(float a, float b, const __attribute__((address_space(3))) char* tbl_lds)
{
    const float mx = __builtin_fmaxf(a, b);
    const uint32_t addr = (uint32_t)(uintptr_t)tbl_lds + np_lse_offset(a, b);
    return mx + *(const __attribute__((address_space(3))) float*)(uintptr_t)addr;
}
__device__ __forceinline__ float np_lse_table_entry(const float* __restrict__ logsum, int i) { return i < NP_LOGSUM_CUT ? logsum[i] : 0.0f; }

// get_scaled_gaussian_from_pore_model_state (src/nanopolish_squiggle_read.h:217-226): double math, float store.
// Returns (mean, stdv, log_inv_sqrt_2pi - log_stdv): the last is the left-associated prefix of
// log_normal_pdf (src/hmm/nanopolish_emissions.h:51-55).
struct np_gauss { float mean, stdv, cl, rinv; };

__device__ __forceinline__ np_gauss np_make_gauss(double lm, double ls, double ll, double scale, double shift, double var, double log_var)
{
    np_gauss g;
    g.mean = (float)(scale * lm + shift);
    g.stdv = (float)(ls * var);
    const float log_stdv = (float)(ll + log_var);
    const float log_inv_sqrt_2pi = -0.918938518f;   // (float)log(0.3989422804014327), emissions.h:43; checked on the host at np_create
    g.cl = log_inv_sqrt_2pi - log_stdv;
    g.rinv = (float)(1.0 / (double)g.stdv);        // correctly rounded reciprocal, once per k-mer (see np_div_exact)
    return g;
}

__device__ __forceinline__ np_gauss np_scale_state(const np_state_dev* __restrict__ model, uint32_t rank,
                                                   double scale, double shift, double var, double log_var)
{
    return np_make_gauss(model[rank].level_mean, model[rank].level_stdv, model[rank].level_log_stdv, scale, shift, var, log_var);
}

// n / d, correctly rounded (== the IEEE fp32 quotient the reference computes), from a correctly rounded
// reciprocal r = RN(1/d): q0 = RN(n r); two Markstein corrections q <- fma(fma(-d, q, n), r, q).  This is the
// body of the hardware division expansion (v_div_scale/v_div_fmas/v_div_fixup only add range scaling, which
// n = x - mean with |n| < 2^15 and d = stdv in [2^-5, 2^6) never need; np_register_model refuses models outside) with the reciprocal hoisted out
// of the per-cell path: 5 VALU ops instead of ~11 incl. a quarter-rate v_rcp_f32.  Equality with `/` is checked
// on the device by np_selftest_division (tests/test_gpu_parity.py) over 2^32 random operand pairs.
__device__ __forceinline__ float np_div_exact(float n, float d, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    return q;
}

// the same with the NEGATED divisor handed over (nd = -d): the two fused corrections then need no operand modifier, and a
// caller that keeps -d instead of d (kernel A's per-k-mer records) never forms the negation
__device__ __forceinline__ float np_div_exact_nd(float n, float nd, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(nd, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(nd, q, n);
    q = __builtin_fmaf(e, r, q);
    return q;
}
// log_probability_match_r9 with the record (mean, -stdv, log-constant, 1/stdv)
__device__ __forceinline__ float np_emission_nd(float x, float mean, float nstdv, float cl, float rinv)
{
    const float a = np_div_exact_nd(x - mean, nstdv, rinv);
    return cl + (-0.5f * a * a);
}

// log_probability_match_r9 (src/hmm/nanopolish_emissions.h:57-68) with drift == 0.
__device__ __forceinline__ float np_emission(float x, const np_gauss& g)
{
    const float a = np_div_exact(x - g.mean, g.stdv, g.rinv);     // == (x - mean) / stdv, IEEE-correct
    return g.cl + (-0.5f * a * a);
}

// Wave-level shift by one lane towards higher lane ids (lane j receives lane j-1); lane 0 receives `fill`.
// DPP wave_shr:1 (0x138) exists on gfx9-family ISAs including gfx950.
__device__ __forceinline__ float np_wave_shr1(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                                                 0x138, 0xf, 0xf, false));
}
// Rotate right by one lane across the whole wave (lane j receives lane j-1, lane 0 receives lane 63): wave_ror:1 (0x13C).
__device__ __forceinline__ float np_wave_ror1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false));
}

// get_next_event / get_closest_event_to, src/nanopolish_squiggle_read.cpp:161-186: the start event of the nearest k-mer at or
// below k_idx that has events (searching back at most 1000 k-mers, end-exclusive), else of the nearest one above.
// ms = base_to_event_map[].start of the read (-1: no events), K = its size.
__device__ __forceinline__ int closest_event(const int32_t* ms, int K, int k_idx)
{
    const int stop_before = 0 > k_idx - 1000 ? 0 : k_idx - 1000;
    const int stop_after = k_idx + 1000 < K - 1 ? k_idx + 1000 : K - 1;
    int event_before = -1, event_after = -1;
    for (int s = k_idx; s != stop_before; s -= 1) { const int ei = ms[s]; if (ei != -1) { event_before = ei; break; } }
    if (event_before != -1) return event_before;
    for (int s = k_idx; s != stop_after; s += 1) { const int ei = ms[s]; if (ei != -1) { event_after = ei; break; } }
    return event_after;
}

// ---- range-checked buffer access (shared by the event aligner and the eventalign chain) --------------------------------
// Range-checked loads through buffer descriptors: an offset outside [0, bytes) -- negative included, it wraps to a
// huge unsigned -- returns 0 instead of faulting, so neither the event-mean prefetch nor the parameter refill needs
// a clamp.  Whatever an out-of-range load returns only ever feeds a masked cell.
// (the operands go through readfirstlane: they are wave-uniform but may sit in VGPRs, and a descriptor the compiler
//  cannot prove scalar costs a waterfall loop per load)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes)
{
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    void* q = (void*)(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
// The byte offset goes to the instruction as ONE register: the compiler would otherwise split `x + c` into a register
// part and the instruction's immediate offset, and the hardware range-checks their sum without 32-bit wrap-around --
// a negative register part with a positive immediate (true offset in range) would then read as 0.
__device__ __forceinline__ int whole_offset(int off) { asm("" : "+v"(off)); return off; }
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, int off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, whole_offset(off), 0, 0));
}
__device__ __forceinline__ float4 buf_f32x4(__amdgpu_buffer_rsrc_t r, int off)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, whole_offset(off), 0, 0));
}

// 16-bit store through a descriptor: an offset outside [0, bytes) is dropped by the hardware
__device__ __forceinline__ void buf_store_u16(__amdgpu_buffer_rsrc_t r, int off, uint32_t v)
{
    __builtin_amdgcn_raw_buffer_store_b16((short)v, r, whole_offset(off), 0, 0);
}
