// np_logf.h -- bit-exact restatement of glibc 2.35's logf (sysdeps/ieee754/flt-32/e_logf.c, the
// ARM optimized-routines algorithm) usable from both host and device code.
//
// Why: calculate_transitions (src/hmm/nanopolish_profile_hmm_r9.inl:61-72) calls the float `log` overload,
// i.e. glibc's logf, on per-read values that only exist on the device in the fused pipeline
// (events_per_base comes out of the event aligner).  ocml's logf is not bit-identical to glibc's, and a
// 1-ulp difference in lp_mm_self can flip a p7_FLogsum table index.  The algorithm below reproduces the host
// libm's result exactly: table and coefficients were read out of this image's libm.so.6 and the function is
// checked against host logf on a 7M-point sweep in tests/test_host_logic.py (CPU) -- both the FMA and non-FMA
// evaluation orders give identical floats on that sweep, we use the non-fused one (-ffp-contract=off).
// PROVENANCE: third-party algorithm and constants, NOT part of the nanopolish reference: glibc 2.35 (sysdeps/ieee754/flt-32/e_logf.c), itself the ARM
// optimized-routines implementation (Szabolcs Nagy, MIT licence; glibc's copy LGPL-2.1-or-later).  Restated here, not copied: the control flow is
// this file's own, the coefficients and the table are the published ones (they ARE the function).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define NP_HD __host__ __device__
#else
#define NP_HD
#endif

NP_HD static inline float np_logf_glibc(float x)
{
    const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;

    uint32_t ix;
    memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        // x < 0x1p-126 or inf or nan
        if (ix * 2u == 0u) return -__builtin_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
        // subnormal: normalise
        float xs = x * 0x1p23f;
        memcpy(&ix, &xs, 4);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    float zf;
    memcpy(&zf, &iz, 4);
    const double invc = T[i][0], logc = T[i][1];
    const double z = (double)zf;
    const double r = z * invc - 1;
    const double y0 = logc + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

// calculate_transitions (src/hmm/nanopolish_profile_hmm_r9.inl:17-76).  Output order = BlockTransitions
// (src/hmm/nanopolish_profile_hmm_r9.h:75-95): mm_self, mb, mk, mm_next, bb, bk, bm_next, bm_self, kk, km.
NP_HD static inline void np_transitions(double events_per_base, double indel_bias, float out[10])
{
    double read_events_per_base = events_per_base;
    read_events_per_base *= indel_bias;
    read_events_per_base = read_events_per_base > 1.25 ? read_events_per_base : 1.25;   // std::max(1.25, x)
    const float p_stay = (float)(1 - (1 / read_events_per_base));
    const float p_skip = 0.0025f;
    const float p_bad = 0.001f;
    const float p_bad_self = p_bad;
    const float p_skip_self = 0.3f;
    const float p_mk = p_skip, p_mb = p_bad, p_mm_self = p_stay;
    const float p_mm_next = 1.0f - p_mm_self - p_mk - p_mb;
    const float p_bb = p_bad_self;
    const float p_bk = (1.0f - p_bb) / 3;
    const float p_bm_next = p_bk, p_bm_self = p_bk;
    const float p_kk = p_skip_self;
    const float p_km = 1.0f - p_kk;
    out[0] = np_logf_glibc(p_mm_self);
    out[1] = np_logf_glibc(p_mb);
    out[2] = np_logf_glibc(p_mk);
    out[3] = np_logf_glibc(p_mm_next);
    out[4] = np_logf_glibc(p_bb);
    out[5] = np_logf_glibc(p_bk);
    out[6] = np_logf_glibc(p_bm_next);
    out[7] = np_logf_glibc(p_bm_self);
    out[8] = np_logf_glibc(p_kk);
    out[9] = np_logf_glibc(p_km);
}
