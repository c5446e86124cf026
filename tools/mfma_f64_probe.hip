// mfma_f64_probe.hip -- can the matrix pipe do kernel A's fp64 additions?  (round 6)
//
// v_mfma_f64_4x4x4_4b_f64 computes, in each of four 4x4 blocks, D = C + A B in IEEE fp64.  With A the identity (one lane in four
// holds 1.0) the product adds exactly one non-zero term per output element, so D[lane] = C[lane] + B[lane]: a per-lane fp64 addition
// that issues on the MATRIX pipe, which the band loop of np_event_align_kernel leaves idle while its vector port is saturated.
// This probe answers, on the device:
//   1. layout   -- which (A lane, B lane) pairs feed which D lane (one-hot sweeps), hence the identity pattern;
//   2. exact    -- D == C + B bit for bit against v_add_f64 (random operands, -inf / huge / tiny in C, finite B);
//   3. rates    -- SIMD cycles per wave-instruction of the MFMA alone, and of a band-like VALU mix with its ten fp64 additions
//                  on the vector port, on the matrix pipe, and removed (the most the move can give), at 8 waves per SIMD.
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o /tmp/mfma_f64_probe && /tmp/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_one(double* o, const double* a, const double* b, const double* c)
{
    const int l = threadIdx.x;
    o[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

// n per-lane additions two ways
__global__ void k_add(double* o_m, double* o_v, const double* ident, const double* b, const double* c, int n)
{
    const int l = threadIdx.x & 63;
    const double A = ident[l];
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const double B = b[(size_t)i * 64 + l], C = c[(size_t)i * 64 + l];
        o_m[(size_t)i * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, C, 0, 0, 0);
        double s;
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(s) : "v"(C), "v"(B));
        o_v[(size_t)i * 64 + l] = s;
    }
}

// MODE 6: as 0, but the four compare masks are stored by two s_store_dwordx4 instead of four v_addc (trace as lane-mask planes)
// MODE 0: ten v_add_f64 + the rest of a band on the vector port; 1: the ten additions as MFMAs; 2: no additions; 3: MFMAs only;
// 4: the vector rest only counted as in 2 but with ten s_nop (issue slots without a pipe); 5: ten v_add_f64 only
#define REP10(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9)
#define TRACE_ADDC "v_addc_co_u32_e64 %[t], s[28:29], %[t], %[t], s[20:21]\n v_addc_co_u32_e64 %[t], s[28:29], %[t], %[t], s[22:23]\n" \
                   "v_addc_co_u32_e64 %[t], s[28:29], %[t], %[t], s[24:25]\n v_addc_co_u32_e64 %[t], s[28:29], %[t], %[t], s[26:27]\n"
// the four compare masks leave through the scalar data cache instead (two 16-byte scalar stores per band)
#define TRACE_FLOAT "v_sub_f32 %[x0], %[a6], %[a0]\n v_sub_f32 %[x1], %[a6], %[a1]\n v_sub_f32 %[x2], %[a7], %[a2]\n v_sub_f32 %[x3], %[a7], %[a3]\n" \
                    "v_fma_f32 %[x0], %[x0], %[big], %[one] clamp\n v_fma_f32 %[x1], %[x1], %[big], %[one] clamp\n v_fma_f32 %[x2], %[x2], %[big], %[one] clamp\n v_fma_f32 %[x3], %[x3], %[big], %[one] clamp\n" \
                    "v_fma_f32 %[x0], %[x0], %[two], %[x1]\n v_fma_f32 %[x2], %[x2], %[two], %[x3]\n v_fma_f32 %[f0], %[f0], %[four], %[x0]\n v_fma_f32 %[f1], %[f1], %[four], %[x2]\n"
#define TRACE_SSTORE "s_store_dwordx4 s[20:23], %[sp], 0x0\n s_store_dwordx4 s[24:27], %[sp], 0x10\n"
template <int MODE> __global__ void __launch_bounds__(256, 8) k_mix(unsigned long long* out, double* sink, const double* ident, int iters, uint64_t* splane = nullptr)
{
    const int l = threadIdx.x & 63;
    const double A = ident[l];
    double B = 1.0 + l * 1e-3;
    double d0 = l, d1 = l + 1, d2 = l + 2, d3 = l + 3, d4 = l + 4, d5 = l + 5, d6 = l + 6, d7 = l + 7, d8 = l + 8, d9 = l + 9;
    float a0 = l, a1 = 1.5f, a2 = 2.5f, a3 = 3.5f, a4 = 4.5f, a5 = 5.5f, a6 = 6.5f, a7 = 7.5f;
    double e0 = 1.0, e1 = 2.0, e2 = 3.0, e3 = 4.0, e4 = 5.0;
    uint32_t t = 0;
    float f0 = 0.0f, f1 = 0.0f;
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define VADD(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##n) : "v"(B));
#define MADD(n) d##n = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, d##n, 0, 0, 0);
        if (MODE == 0 || MODE == 5 || MODE == 6 || MODE == 8 || MODE == 9) { REP10(VADD) }
        if (MODE == 1 || MODE == 3) { REP10(MADD) }
        if (MODE == 4) { asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0"); }
        if (MODE != 3 && MODE != 5) {
            // the band's other 44 vector instructions by class: 18 fast fp32, 5 + 6 conversions, 4 compares to scalar pairs + 4 carry adds,
            // 2 max3, 2 selects, 1 DPP rotate, 2 lane reads
            if (MODE == 6 || MODE == 7) {
                const uint64_t* sp = splane + ((size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4096 + (size_t)(i & 1023) * 4);      // 32 B per band, a 32 KB window per wave
                sp = (const uint64_t*)(((uint64_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)sp >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint64_t)sp));
            asm volatile(
                "v_fma_f32 %[a0], %[a0], %[a1], %[a2]\n v_fma_f32 %[a1], %[a1], %[a2], %[a3]\n v_fma_f32 %[a2], %[a2], %[a3], %[a4]\n"
                "v_fma_f32 %[a3], %[a3], %[a4], %[a5]\n v_fma_f32 %[a4], %[a4], %[a5], %[a6]\n v_fma_f32 %[a5], %[a5], %[a6], %[a7]\n"
                "v_mul_f32 %[a6], %[a6], %[a7]\n v_mul_f32 %[a7], %[a7], %[a0]\n v_sub_f32 %[a0], %[a0], %[a1]\n"
                "v_fma_f32 %[a0], %[a0], %[a1], %[a2]\n v_fma_f32 %[a1], %[a1], %[a2], %[a3]\n v_fma_f32 %[a2], %[a2], %[a3], %[a4]\n"
                "v_fma_f32 %[a3], %[a3], %[a4], %[a5]\n v_fma_f32 %[a4], %[a4], %[a5], %[a6]\n v_fma_f32 %[a5], %[a5], %[a6], %[a7]\n"
                "v_mul_f32 %[a6], %[a6], %[a7]\n v_mul_f32 %[a7], %[a7], %[a0]\n v_sub_f32 %[a0], %[a0], %[a1]\n"
                "v_cvt_f64_f32 %[e0], %[a0]\n v_cvt_f64_f32 %[e1], %[a1]\n v_cvt_f64_f32 %[e2], %[a2]\n v_cvt_f64_f32 %[e3], %[a3]\n v_cvt_f64_f32 %[e4], %[a4]\n"
                "v_cvt_f32_f64 %[a0], %[e0]\n v_cvt_f32_f64 %[a1], %[e1]\n v_cvt_f32_f64 %[a2], %[e2]\n v_cvt_f32_f64 %[a3], %[e3]\n v_cvt_f32_f64 %[a4], %[e4]\n v_cvt_f32_f64 %[a5], %[e0]\n"
                "v_cmp_eq_f32_e64 s[20:21], %[a0], %[a1]\n v_cmp_eq_f32_e64 s[22:23], %[a1], %[a2]\n v_cmp_eq_f32_e64 s[24:25], %[a2], %[a3]\n v_cmp_eq_f32_e64 s[26:27], %[a3], %[a4]\n"
                TRACE_SSTORE
                "v_max3_f32 %[a6], %[a6], %[a0], %[a1]\n v_max3_f32 %[a7], %[a7], %[a2], %[a3]\n"
                "v_cndmask_b32_e64 %[a6], %[a6], %[a4], s[20:21]\n v_cndmask_b32_e64 %[a7], %[a7], %[a5], s[22:23]\n"
                "v_mov_b32_dpp %[a5], %[a7] wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                "v_readlane_b32 s30, %[a6], 5\n v_readlane_b32 s31, %[a7], 9\n"
                : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7),
                  [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [e3] "+v"(e3), [e4] "+v"(e4), [t] "+v"(t)
                : [sp] "s"(sp) : "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
            } else if (MODE == 8 || MODE == 9) {
            // MODE 8 (round 6, late): the trace WITHOUT compares and carry adds -- per candidate a float difference, a clamped fma that turns
            // "difference == 0" into 1.0 / 0.0, and per cell two fmas that shift the two bits into a float accumulator: 12 fast-class
            // instructions for the 8 slow ones.  MODE 9: no trace instructions at all (the bound).
#define BAND_HEAD \
                "v_fma_f32 %[a0], %[a0], %[a1], %[a2]\n v_fma_f32 %[a1], %[a1], %[a2], %[a3]\n v_fma_f32 %[a2], %[a2], %[a3], %[a4]\n" \
                "v_fma_f32 %[a3], %[a3], %[a4], %[a5]\n v_fma_f32 %[a4], %[a4], %[a5], %[a6]\n v_fma_f32 %[a5], %[a5], %[a6], %[a7]\n" \
                "v_mul_f32 %[a6], %[a6], %[a7]\n v_mul_f32 %[a7], %[a7], %[a0]\n v_sub_f32 %[a0], %[a0], %[a1]\n" \
                "v_fma_f32 %[a0], %[a0], %[a1], %[a2]\n v_fma_f32 %[a1], %[a1], %[a2], %[a3]\n v_fma_f32 %[a2], %[a2], %[a3], %[a4]\n" \
                "v_fma_f32 %[a3], %[a3], %[a4], %[a5]\n v_fma_f32 %[a4], %[a4], %[a5], %[a6]\n v_fma_f32 %[a5], %[a5], %[a6], %[a7]\n" \
                "v_mul_f32 %[a6], %[a6], %[a7]\n v_mul_f32 %[a7], %[a7], %[a0]\n v_sub_f32 %[a0], %[a0], %[a1]\n" \
                "v_cvt_f64_f32 %[e0], %[a0]\n v_cvt_f64_f32 %[e1], %[a1]\n v_cvt_f64_f32 %[e2], %[a2]\n v_cvt_f64_f32 %[e3], %[a3]\n v_cvt_f64_f32 %[e4], %[a4]\n" \
                "v_cvt_f32_f64 %[a0], %[e0]\n v_cvt_f32_f64 %[a1], %[e1]\n v_cvt_f32_f64 %[a2], %[e2]\n v_cvt_f32_f64 %[a3], %[e3]\n v_cvt_f32_f64 %[a4], %[e4]\n v_cvt_f32_f64 %[a5], %[e0]\n" \
                "v_max3_f32 %[a6], %[a6], %[a0], %[a1]\n v_max3_f32 %[a7], %[a7], %[a2], %[a3]\n"
#define BAND_TAIL \
                "v_cndmask_b32_e64 %[a6], %[a6], %[a4], s[20:21]\n v_cndmask_b32_e64 %[a7], %[a7], %[a5], s[22:23]\n" \
                "v_mov_b32_dpp %[a5], %[a7] wave_ror:1 row_mask:0xf bank_mask:0xf\n" \
                "v_readlane_b32 s30, %[a6], 5\n v_readlane_b32 s31, %[a7], 9\n"
#define BAND_OPS \
                : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7), \
                  [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [e3] "+v"(e3), [e4] "+v"(e4), [t] "+v"(t), [f0] "+v"(f0), [f1] "+v"(f1), \
                  [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3) \
                : [one] "v"(1.0f), [big] "v"(-1.2676506e30f), [two] "v"(2.0f), [four] "v"(4.0f) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31"
            float x0, x1, x2, x3;
            if (MODE == 8) asm volatile(BAND_HEAD TRACE_FLOAT BAND_TAIL BAND_OPS);
            else asm volatile(BAND_HEAD BAND_TAIL BAND_OPS);
            if ((i & 7) == 7) { f0 = 0.0f; f1 = 0.0f; }       // (the accumulators start over every eight bands; the real kernel converts and stores them there)
            } else {
            asm volatile(
                "v_fma_f32 %[a0], %[a0], %[a1], %[a2]\n v_fma_f32 %[a1], %[a1], %[a2], %[a3]\n v_fma_f32 %[a2], %[a2], %[a3], %[a4]\n"
                "v_fma_f32 %[a3], %[a3], %[a4], %[a5]\n v_fma_f32 %[a4], %[a4], %[a5], %[a6]\n v_fma_f32 %[a5], %[a5], %[a6], %[a7]\n"
                "v_mul_f32 %[a6], %[a6], %[a7]\n v_mul_f32 %[a7], %[a7], %[a0]\n v_sub_f32 %[a0], %[a0], %[a1]\n"
                "v_fma_f32 %[a0], %[a0], %[a1], %[a2]\n v_fma_f32 %[a1], %[a1], %[a2], %[a3]\n v_fma_f32 %[a2], %[a2], %[a3], %[a4]\n"
                "v_fma_f32 %[a3], %[a3], %[a4], %[a5]\n v_fma_f32 %[a4], %[a4], %[a5], %[a6]\n v_fma_f32 %[a5], %[a5], %[a6], %[a7]\n"
                "v_mul_f32 %[a6], %[a6], %[a7]\n v_mul_f32 %[a7], %[a7], %[a0]\n v_sub_f32 %[a0], %[a0], %[a1]\n"
                "v_cvt_f64_f32 %[e0], %[a0]\n v_cvt_f64_f32 %[e1], %[a1]\n v_cvt_f64_f32 %[e2], %[a2]\n v_cvt_f64_f32 %[e3], %[a3]\n v_cvt_f64_f32 %[e4], %[a4]\n"
                "v_cvt_f32_f64 %[a0], %[e0]\n v_cvt_f32_f64 %[a1], %[e1]\n v_cvt_f32_f64 %[a2], %[e2]\n v_cvt_f32_f64 %[a3], %[e3]\n v_cvt_f32_f64 %[a4], %[e4]\n v_cvt_f32_f64 %[a5], %[e0]\n"
                "v_cmp_eq_f32_e64 s[20:21], %[a0], %[a1]\n v_cmp_eq_f32_e64 s[22:23], %[a1], %[a2]\n v_cmp_eq_f32_e64 s[24:25], %[a2], %[a3]\n v_cmp_eq_f32_e64 s[26:27], %[a3], %[a4]\n"
                TRACE_ADDC
                "v_max3_f32 %[a6], %[a6], %[a0], %[a1]\n v_max3_f32 %[a7], %[a7], %[a2], %[a3]\n"
                "v_cndmask_b32_e64 %[a6], %[a6], %[a4], s[20:21]\n v_cndmask_b32_e64 %[a7], %[a7], %[a5], s[22:23]\n"
                "v_mov_b32_dpp %[a5], %[a7] wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                "v_readlane_b32 s30, %[a6], 5\n v_readlane_b32 s31, %[a7], 9\n"
                : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7),
                  [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [e3] "+v"(e3), [e4] "+v"(e4), [t] "+v"(t)
                : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
            }

        }
    }
    if (MODE == 6 || MODE == 7) asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = t1 - t0; }
    sink[blockIdx.x * 256 + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + d8 + d9 + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + e0 + e1 + e2 + e3 + e4 + t + f0 + f1;
}

// ---- 4. the multi-read ring as an instruction-stream mock: one band of NS ring slots per lane, every instruction of the
// real band step with its real dependences (emission: sub, mul, four corrections, three for cl - a^2/2; candidates: 2 conversions in, 5 fp64
// additions with scalar constants, 3 conversions out; max3; 2 compares to scalar pairs + 2 carry adds; window select), plus per band the
// left-neighbour transfer (NROT DPP rotates + one conversion) and NRD lane reads with scalar selects (two band ends per read).
//   NS = 2, NROT = 1, NRD = 2 : the shipped layout (one read per wave, 128-slot ring, 100 live)          -> per read: this
//   NS = 5, NROT = 3, NRD = 6 : three reads per wave on 21 lanes x 5 slots each (105-slot rings, 100 live) -> per read: this / 3
// The mock leaves out what the three-read form would ADD (per-read move decisions as vector code or three scalar chains, per-read window
// masks, per-read re-targets, ragged band counts): it is the most the layout can give.
template <int NS, int NROT, int NRD, int WAVES> __global__ void __launch_bounds__(256, WAVES) k_ring(unsigned long long* out, float* sink, int iters)
{
    const int l = threadIdx.x & 63;
    float x[NS], g0[NS], g1[NS], g2[NS], g3[NS], p[NS];
    double d[NS];
    uint32_t t = l;
#pragma unroll
    for (int i = 0; i < NS; ++i) { x[i] = 80.f + l + i; g0[i] = 79.f + i; g1[i] = -1.5f - 0.01f * l; g2[i] = -1.3f; g3[i] = 1.f / 1.5f; p[i] = -100.f - l; d[i] = -101.0 - i; }
    double L = -99.0;
    float lft = -98.f;
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < NROT; ++r) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(lft) : "v"(p[NS - 1]));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(L) : "v"(lft));
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            float n_, q_, e_, em, sd, su, sl, m;
            double E, P, D, U, Lq;
            asm volatile("v_sub_f32 %[n], %[x], %[g0]\n v_mul_f32 %[q], %[n], %[g3]\n v_fma_f32 %[e], %[g1], %[q], %[n]\n v_fma_f32 %[q], %[e], %[g3], %[q]\n"
                         "v_fma_f32 %[e], %[g1], %[q], %[n]\n v_fma_f32 %[q], %[e], %[g3], %[q]\n v_mul_f32 %[e], 0.5, %[q]\n v_mul_f32 %[e], %[e], %[q]\n"
                         "v_sub_f32 %[em], %[g2], %[e]\n"
                         "v_cvt_f64_f32 %[E], %[em]\n v_cvt_f64_f32 %[P], %[p]\n"
                         "v_add_f64 %[D], %[d], s[20:21]\n v_add_f64 %[D], %[D], %[E]\n v_add_f64 %[U], %[P], s[22:23]\n v_add_f64 %[U], %[U], %[E]\n"
                         "v_add_f64 %[Lq], %[L], s[24:25]\n"
                         "v_cvt_f32_f64 %[sd], %[D]\n v_cvt_f32_f64 %[su], %[U]\n v_cvt_f32_f64 %[sl], %[Lq]\n"
                         "v_max3_f32 %[m], %[sd], %[su], %[sl]\n"
                         "v_cmp_eq_f32_e64 s[26:27], %[m], %[sl]\n v_cmp_eq_f32_e64 s[28:29], %[m], %[su]\n"
                         "v_addc_co_u32_e64 %[t], s[30:31], %[t], %[t], s[26:27]\n v_addc_co_u32_e64 %[t], s[30:31], %[t], %[t], s[28:29]\n"
                         "v_cndmask_b32_e64 %[p], %[g2], %[m], s[18:19]\n"
                         : [n] "=&v"(n_), [q] "=&v"(q_), [e] "=&v"(e_), [em] "=&v"(em), [sd] "=&v"(sd), [su] "=&v"(su), [sl] "=&v"(sl), [m] "=&v"(m),
                           [E] "=&v"(E), [P] "=&v"(P), [D] "=&v"(D), [U] "=&v"(U), [Lq] "=&v"(Lq), [p] "+v"(p[i]), [t] "+v"(t)
                         : [x] "v"(x[i]), [g0] "v"(g0[i]), [g1] "v"(g1[i]), [g2] "v"(g2[i]), [g3] "v"(g3[i]), [d] "v"(d[i]), [L] "v"(L)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s18", "s19");
            d[i] = L; L = P;                       // the next slot's left neighbour is this slot's cell; this slot's next diagonal its left (register renames in the real loop)
        }
#pragma unroll
        for (int r = 0; r < NRD; ++r) asm volatile("v_readlane_b32 s14, %0, s15" : : "v"(p[r % NS]) : "s14");
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float acc = (float)L + lft + t;
#pragma unroll
    for (int i = 0; i < NS; ++i) acc += p[i] + (float)d[i];
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int NS, int NROT, int NRD, int WAVES> static int run_ring(const char* name, int reads_per_wave)
{
    int dev = 0; hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
    const int blocks = pr.multiProcessorCount * WAVES, iters = 20000;
    unsigned long long* out; float* sink;
    CK(hipMalloc(&out, blocks * 8)); CK(hipMalloc(&sink, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_ring<NS, NROT, NRD, WAVES>), dim3(blocks), dim3(256), 0, 0, out, sink, 200);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_ring<NS, NROT, NRD, WAVES>), dim3(blocks), dim3(256), 0, 0, out, sink, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns_band = ms * 1e6 / ((double)iters * WAVES);           // per SIMD: WAVES waves side by side
    printf("  %-44s waves/SIMD %d: %8.2f ms  %7.1f SIMD-cycles (2.4 GHz) per wave-band = %6.1f per READ-band\n", name, WAVES, ms, ns_band * 2.4, ns_band * 2.4 / reads_per_wave);
    CK(hipFree(out)); CK(hipFree(sink));
    return 0;
}

// ---- 5. the log-sum table gather of kernel B: ds_read_b32 against ds_read_b64 banking ---------------------------------------
// A 64 000-byte table in LDS, 512 threads per workgroup, two workgroups per CU (four waves per SIMD, kernel B's shape); every thread runs eight
// independent chains of look-ups at pseudo-random entries (the index of the next look-up depends on the value read, as a log-sum's does on the
// previous sum).  B64 = 0: ds_read_b32 at 4 i (bank (a / 4) mod 32 per 32-lane group); 1: ds_read_b64 at the aligned pair 8 (i / 2) and a select of
// the wanted half (bank PAIR (a / 8) mod 32: 32 lanes onto 32 pairs -- the same balls-in-bins as 32 lanes onto 32 banks -- plus one more vector
// instruction per look-up).
template <int B64> __global__ void __launch_bounds__(512, 4) k_gather(float* sink, int iters)
{
    __shared__ float tbl[16000];
    for (int i = threadIdx.x; i < 16000; i += 512) tbl[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-4f;
    __syncthreads();
    uint32_t x[8]; float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { x[c] = threadIdx.x * 7919u + c * 104729u + blockIdx.x; acc[c] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            x[c] = x[c] * 1664525u + 1013904223u + (uint32_t)__builtin_bit_cast(int, acc[c]);
            const uint32_t i = (x[c] >> 8) % 15700u;
            float v;
            if (B64) {
                const float2 pr = *reinterpret_cast<const float2*>(&tbl[i & ~1u]);
                v = (i & 1u) ? pr.y : pr.x;
            } else v = tbl[i];
            acc[c] += v;
        }
    }
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) a += acc[c];
    sink[blockIdx.x * 512 + threadIdx.x] = a;
}
template <int B64> static int run_gather(const char* name)
{
    int dev = 0; hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
    const int blocks = pr.multiProcessorCount * 2, iters = 4000;
    float* sink; CK(hipMalloc(&sink, (size_t)blocks * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<B64>, dim3(blocks), dim3(512), 0, 0, sink, 50);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<B64>, dim3(blocks), dim3(512), 0, 0, sink, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_gathers_per_cu = 16.0 * 8.0 * iters;        // 16 waves per CU, 8 look-ups per iteration
    printf("  %-40s %8.3f ms  %6.2f CU-cycles (2.4 GHz) per wave-instruction of look-ups\n", name, ms, ms * 1e-3 * 2.4e9 / wave_gathers_per_cu);
    CK(hipFree(sink));
    return 0;
}

template <int MODE> static int run_mix(const char* name, const double* ident_d, int waves_per_simd)
{
    int dev = 0; hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;          // 256-thread blocks: 4 waves, one per SIMD
    const int iters = 20000;
    unsigned long long* out; double* sink; uint64_t* splane = nullptr;
    CK(hipMalloc(&out, blocks * 16)); CK(hipMalloc(&sink, (size_t)blocks * 256 * 8));
    if (MODE == 6 || MODE == 7) CK(hipMalloc(&splane, (size_t)blocks * 4 * 4096 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mix<MODE>, dim3(blocks), dim3(256), 0, 0, out, sink, ident_d, 200, splane);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mix<MODE>, dim3(blocks), dim3(256), 0, 0, out, sink, ident_d, iters, splane);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 2); CK(hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost));
    double cyc = 0, wc = 0; for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; wc += h[2 * i + 1]; }
    cyc /= blocks; wc /= blocks;
    // per SIMD: waves_per_simd waves, each `iters` iterations, in ms
    const double ns_per_iter_simd = ms * 1e6 / ((double)iters * waves_per_simd);
    printf("  %-34s waves/SIMD %d: %8.2f ms  %7.1f ns per iteration and SIMD-wave (= %6.1f cycles at 2.4 GHz)  s_memtime/iter %.1f  wall_clock/iter %.2f\n",
           name, waves_per_simd, ms, ns_per_iter_simd, ns_per_iter_simd * 2.4, cyc / iters, wc / iters);
    CK(hipFree(out)); CK(hipFree(sink)); if (splane) CK(hipFree(splane));
    return 0;
}

int main()
{
    double *a, *b, *c, *o;
    CK(hipMalloc(&a, 512)); CK(hipMalloc(&b, 512)); CK(hipMalloc(&c, 512)); CK(hipMalloc(&o, 512));
    double ha[64], hb[64], hc[64], ho[64];
    // ---- 1. layout: A one-hot at lane la, B[l] = l + 1, C = 0 -> D[L] = B value of the lane that pairs with la for output L
    int pairB[64][64]; memset(pairB, -1, sizeof pairB);          // pairB[la][L] = lb
    for (int la = 0; la < 64; ++la) {
        for (int l = 0; l < 64; ++l) { ha[l] = l == la ? 1.0 : 0.0; hb[l] = l + 1; hc[l] = 0.0; }
        CK(hipMemcpy(a, ha, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(c, hc, 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, o, a, b, c);
        CK(hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost));
        for (int L = 0; L < 64; ++L) if (ho[L] != 0.0) pairB[la][L] = (int)ho[L] - 1;
    }
    printf("layout: A lane -> (D lane <- B lane)\n");
    for (int la = 0; la < 64; ++la) {
        printf("  A %2d:", la);
        for (int L = 0; L < 64; ++L) if (pairB[la][L] >= 0) printf(" (%d<-%d)", L, pairB[la][L]);
        printf("\n");
    }
    // identity pattern with B as the addend: A lanes whose every output reads ITS OWN lane of B
    double identB[64]; int coverB[64] = {0};
    for (int la = 0; la < 64; ++la) {
        bool own = true, any = false;
        for (int L = 0; L < 64; ++L) if (pairB[la][L] >= 0) { any = true; if (pairB[la][L] != L) own = false; }
        identB[la] = (own && any) ? 1.0 : 0.0;
        if (own && any) for (int L = 0; L < 64; ++L) if (pairB[la][L] >= 0) coverB[L]++;
    }
    bool okB = true; for (int L = 0; L < 64; ++L) if (coverB[L] != 1) okB = false;
    printf("identity-in-A pattern (D[l] = C[l] + B[l]): %s; lanes with 1.0:", okB ? "EXISTS" : "does not exist");
    for (int l = 0; l < 64; ++l) if (identB[l] != 0.0) printf(" %d", l);
    printf("\n");
    if (!okB) { printf("RESULT: no per-lane addition through A = identity\n"); return 2; }

    double* ident_d; CK(hipMalloc(&ident_d, 512)); CK(hipMemcpy(ident_d, identB, 512, hipMemcpyHostToDevice));

    // ---- 2. exactness
    {
        const int n = 1 << 16;
        std::vector<double> hB((size_t)n * 64), hC((size_t)n * 64), oM((size_t)n * 64), oV((size_t)n * 64);
        std::mt19937_64 rng(12345);
        std::uniform_real_distribution<double> u(-1.0, 1.0);
        for (size_t i = 0; i < hB.size(); ++i) {
            const int kind = (int)(rng() % 16);
            double C = u(rng) * std::ldexp(1.0, (int)(rng() % 40) - 10), B = u(rng) * std::ldexp(1.0, (int)(rng() % 30) - 20);
            if (kind == 0) C = -INFINITY;
            if (kind == 1) C = (double)(float)C;                       // a float-valued cell
            if (kind == 2) { C = (double)(float)C; B = (double)(float)B; }
            if (kind == 3) C = u(rng) * 1e300;
            if (kind == 4) C = u(rng) * 1e-300;
            if (kind == 5) B = u(rng) * 1e-310;                        // subnormal addend
            if (kind == 6) { C = 0.0; }
            if (kind == 7) { C = -B; }                                  // exact cancellation
            if (kind == 8) { C = std::nextafter(-B, 0.0); }
            hB[i] = B; hC[i] = C;
        }
        double *dB, *dC, *dM, *dV;
        CK(hipMalloc(&dB, hB.size() * 8)); CK(hipMalloc(&dC, hB.size() * 8)); CK(hipMalloc(&dM, hB.size() * 8)); CK(hipMalloc(&dV, hB.size() * 8));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, hC.data(), hB.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_add, dim3(1024), dim3(64), 0, 0, dM, dV, ident_d, dB, dC, n);
        CK(hipMemcpy(oM.data(), dM, hB.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(oV.data(), dV, hB.size() * 8, hipMemcpyDeviceToHost));
        size_t bad = 0, bad_host = 0;
        for (size_t i = 0; i < hB.size(); ++i) {
            uint64_t x, y, z; const double hs = hC[i] + hB[i];
            memcpy(&x, &oM[i], 8); memcpy(&y, &oV[i], 8); memcpy(&z, &hs, 8);
            if (x != y) { if (bad < 8) printf("  MISMATCH C=%a B=%a mfma=%a valu=%a\n", hC[i], hB[i], oM[i], oV[i]); ++bad; }
            if (y != z && !(hs != hs)) ++bad_host;
        }
        printf("exactness: %zu additions, %zu differ between the matrix pipe and v_add_f64 (%zu v_add_f64 results differ from the host's)\n", hB.size(), bad, bad_host);
        CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dM)); CK(hipFree(dV));
    }

    // ---- 3. rates
    printf("rates (per loop iteration of one wave; a SIMD runs waves/SIMD of them side by side):\n");
    for (int w : {1, 4, 8}) {
        if (run_mix<3>("10 MFMA f64 4x4x4 only", ident_d, w)) return 1;
        if (run_mix<5>("10 v_add_f64 only", ident_d, w)) return 1;
        if (run_mix<2>("band rest (44 vector), no adds", ident_d, w)) return 1;
        if (run_mix<4>("band rest + 10 s_nop", ident_d, w)) return 1;
        if (run_mix<0>("band rest + 10 v_add_f64", ident_d, w)) return 1;
        if (run_mix<1>("band rest + 10 MFMA", ident_d, w)) return 1;
        if (run_mix<8>("band, trace as 12 fast fp32 ops", ident_d, w)) return 1;
        if (run_mix<9>("band, no trace at all", ident_d, w)) return 1;
        if (run_mix<0>("band rest + 10 v_add_f64 (again)", ident_d, w)) return 1;
        if (run_mix<6>("band (adds) - 4 addc + 2 s_store_x4", ident_d, w)) return 1;
    }
    printf("ring layouts (instruction-stream mock of the FAST band loop, no walk, no loads):\n");
    if (run_ring<2, 1, 2, 8>("2 slots/lane, 1 read/wave (shipped)", 1)) return 1;
    if (run_ring<2, 1, 2, 6>("2 slots/lane, 1 read/wave", 1)) return 1;
    if (run_ring<2, 1, 2, 4>("2 slots/lane, 1 read/wave", 1)) return 1;
    if (run_ring<5, 3, 6, 4>("5 slots/lane, 3 reads/wave (21 lanes each)", 3)) return 1;
    if (run_ring<5, 3, 6, 3>("5 slots/lane, 3 reads/wave (21 lanes each)", 3)) return 1;
    if (run_ring<5, 3, 6, 5>("5 slots/lane, 3 reads/wave (needs <= 96 registers)", 3)) return 1;
    if (run_ring<7, 1, 8, 4>("7 slots/lane, 4 reads/wave (16-lane DPP rows)", 4)) return 1;
    if (run_ring<7, 1, 8, 3>("7 slots/lane, 4 reads/wave (16-lane DPP rows)", 4)) return 1;
    printf("log-sum table gather (16 000 floats in LDS, random entries, four waves per SIMD):\n");
    if (run_gather<0>("ds_read_b32 (shipped)")) return 1;
    if (run_gather<1>("ds_read_b64 pair + select")) return 1;
    if (run_gather<0>("ds_read_b32 (shipped)")) return 1;
    if (run_gather<1>("ds_read_b64 pair + select")) return 1;
    return 0;
}
