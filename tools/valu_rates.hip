// valu_rates.hip -- issue cost (cycles per wave-instruction) of the VALU ops the two kernels are made of, measured on
// the device with s_memtime around long independent instruction streams.  One wave per SIMD (no contention), 8
// independent destination registers per op so that latency does not limit issue.  Build + run:
//   hipcc -O2 --offload-arch=gfx950 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ void k(unsigned long long* out, float* sink, int iters)
{
    float a0 = threadIdx.x, a1 = 1.5f, a2 = 2.5f, a3 = 3.5f, a4 = 4.5f, a5 = 5.5f, a6 = 6.5f, a7 = 7.5f;
    float b0 = 0.5f, b1 = 1.25f, b2 = 2.25f, b3 = 3.25f, b4 = 4.25f, b5 = 5.25f, b6 = 6.25f, b7 = 7.25f;
    double d0 = 1.0, d1 = 2.0, d2 = 3.0, d3 = 4.0, d4 = 5.0, d5 = 6.0, d6 = 7.0, d7 = 8.0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {1.f, 2.f}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
    __shared__ float lds[512]; lds[threadIdx.x] = 1.0f; __syncthreads();   // 2 KB: 32 waves per CU co-reside (16 KB here capped a CU at 10 waves: the first r04 calibration)
    int l0 = threadIdx.x * 4, l1 = l0 + 256, l2 = l0 + 512, l3 = l0 + 768, l4 = l0 + 1024, l5 = l0 + 1280, l6 = l0 + 1536, l7 = l0 + 1792;
    int s0 = 0; int q0 = 0, q1 = 1, q2 = 2, q3 = 3, q4 = 4, q5 = 5, q6 = 6, q7 = 7;
    asm volatile("v_cmp_gt_f32 vcc, 1.0, %0\n s_mov_b64 s[10:11], vcc" : : "v"(a0) : "vcc", "s10", "s11");
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define CVT64(n) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##n) : "v"(a##n));
#define CVT32(n) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a##n) : "v"(d##n));
#define ADD64(n) asm volatile("v_add_f64 %0, %1, %1" : "=v"(d##n) : "v"(d##n));
#define FMA32(n) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(a##n) : "v"(a##n));
#define PKFMA(n) asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(p##n) : "v"(p##n));
#define PKMUL(n) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(p##n) : "v"(p##n));
#define MAX3(n) asm volatile("v_max3_f32 %0, %1, %1, %1" : "=v"(a##n) : "v"(a##n));
#define CNDM(n) asm volatile("v_cndmask_b32 %0, %1, %1, vcc" : "=v"(a##n) : "v"(a##n));
#define DPPM(n) asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(a##n) : "v"(a##n));
#define RDLN(n) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(a##n));
#define CMPF(n) asm volatile("v_cmp_eq_f32 vcc, %0, %0" : : "v"(a##n) : "vcc");
#define FMA64(n) asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(d##n) : "v"(d##n));
#define ADDF(n) asm volatile("v_add_f32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define CND64(n) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[10:11]" : "=v"(a##n) : "v"(a##n), "v"(a0) : "s10", "s11");
#define CNDV2(n) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a##n) : "v"(a##n), "v"(a0));
#define SUBU(n) asm volatile("v_sub_u32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define MAXF(n) asm volatile("v_max_f32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define PKADD(n) asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(p##n) : "v"(p##n));
#define MOVB(n) asm volatile("v_mov_b32 %0, %1" : "=v"(a##n) : "v"(a##n));
#define FMA3(n) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a##n) : "v"(a##n), "v"(a0), "v"(a1));
#define MAX3D(n) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a##n) : "v"(a##n), "v"(a0), "v"(a1));
#define CMPS(n) asm volatile("v_cmp_eq_f32_e64 s[10:11], %0, %0" : : "v"(a##n) : "s10", "s11");
#define CNDE64V(n) asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc" : "=v"(a##n) : "v"(a##n), "v"(a0));
#define ADDC(n) asm volatile("v_addc_co_u32 %0, vcc, %1, %1, vcc" : "=v"(a##n) : "v"(a##n) : "vcc");
#define CNDS0(n) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[0:1]" : "=v"(a##n) : "v"(a##n), "v"(a0));
#define CNDMIX(n) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a##n) : "v"(a##n), "v"(a0) : "vcc");
#define CNDMIXS(n) asm volatile("v_cmp_gt_f32_e64 s[10:11], %1, %2\n v_cndmask_b32_e64 %0, %1, %2, s[10:11]" : "=v"(a##n) : "v"(a##n), "v"(a0) : "s10", "s11");
#define C1CND4 asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc" : "+v"(a1), "+v"(a0), "+v"(a2), "+v"(a3), "+v"(a4) : : "vcc");
#define C1CND4S asm volatile("v_cmp_gt_f32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %2, %2, %1, s[10:11]\n v_cndmask_b32_e64 %3, %3, %1, s[10:11]\n v_cndmask_b32_e64 %4, %4, %1, s[10:11]" : "+v"(a1), "+v"(a0), "+v"(a2), "+v"(a3), "+v"(a4) : : "s10", "s11");
#define C1CND4E asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %2, %2, %1, vcc\n v_cndmask_b32_e64 %3, %3, %1, vcc\n v_cndmask_b32_e64 %4, %4, %1, vcc" : "+v"(a1), "+v"(a0), "+v"(a2), "+v"(a3), "+v"(a4) : : "vcc");
#define C1X4CND asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n v_add_f32 %4, %4, %4\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1), "+v"(a0), "+v"(a2), "+v"(a3), "+v"(a4) : : "vcc");
#define SADD(n) asm volatile("s_add_i32 %0, %0, 1" : "+s"(q##n) : : "scc");
#define SMIX(n) asm volatile("s_add_i32 %0, %0, 1\n v_add_f32 %1, %1, %1" : "+s"(q##n), "+v"(a##n) : : "scc");
#define SCMP(n) asm volatile("s_cmp_lt_i32 %0, 5\n s_cselect_b32 %0, %0, 3" : "+s"(q##n) : : "scc");
#define CVTU(n) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(a##n) : "v"(a##n));
#define CVTI32(n) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a##n) : "v"(a##n));
#define MINU(n) asm volatile("v_min_u32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define MINF(n) asm volatile("v_min_f32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define MULF(n) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define MULFA(n) asm volatile("v_mul_f32 %0, |%1|, %1" : "=v"(a##n) : "v"(a##n));
#define LSHLR(n) asm volatile("v_lshlrev_b32 %0, 2, %1" : "=v"(a##n) : "v"(a##n));
#define ANDB(n) asm volatile("v_and_b32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define DSRD(n) asm volatile("ds_read_b32 %0, %1" : "=v"(a##n) : "v"(l##n));
#define MED3(n) asm volatile("v_med3_i32 %0, %1, %1, %1" : "=v"(a##n) : "v"(a##n));
#define MAXI(n) asm volatile("v_max_i32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define SUBF(n) asm volatile("v_sub_f32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define ADDU(n) asm volatile("v_add_u32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define LSHLADD(n) asm volatile("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(a##n) : "v"(a##n));
#define CMPU(n) asm volatile("v_cmp_gt_u32 vcc, %0, %0" : : "v"(a##n) : "vcc");
#define LSHL(n) asm volatile("v_lshl_or_b32 %0, %1, 2, %1" : "=v"(a##n) : "v"(a##n));
#define MUL64(n) asm volatile("v_mul_f64 %0, %1, %1" : "=v"(d##n) : "v"(d##n));
#define CVTI(n) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d##n) : "v"(a##n));
// ---- round 2: candidates for the interleaved band layout / carry-chain trace packing ----
#define SUBCO32(n) asm volatile("v_sub_co_u32_e32 %0, vcc, %1, %1" : "=v"(a##n) : "v"(a##n) : "vcc");
#define SUBCO64(n) asm volatile("v_sub_co_u32_e64 %0, s[10:11], %1, %1" : "=v"(a##n) : "v"(a##n) : "s10", "s11");
#define ADDC64(n) asm volatile("v_addc_co_u32_e64 %0, s[12:13], %1, %1, s[10:11]" : "=v"(a##n) : "v"(a##n) : "s12", "s13");
#define TRACE8 asm volatile("v_sub_co_u32_e64 %1, s[10:11], %2, %3\n v_sub_co_u32_e64 %1, s[12:13], %2, %4\n v_sub_co_u32_e64 %1, s[14:15], %3, %4\n v_sub_co_u32_e64 %1, s[16:17], %4, %2\n" \
                            "v_addc_co_u32_e64 %0, s[18:19], %0, %0, s[10:11]\n v_addc_co_u32_e64 %0, s[18:19], %0, %0, s[12:13]\n v_addc_co_u32_e64 %0, s[18:19], %0, %0, s[14:15]\n v_addc_co_u32_e64 %0, s[18:19], %0, %0, s[16:17]" \
                            : "+v"(a0), "=&v"(a1) : "v"(a2), "v"(a3), "v"(a4) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19");
#define RDLNS(n) asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s0) : "v"(a##n), "s"(q##n));
#define SLSHL64(n) asm volatile("s_lshl_b64 s[10:11], s[10:11], 1" ::: "s10", "s11", "scc");
#define SOR64(n) asm volatile("s_or_b64 s[10:11], s[10:11], s[12:13]" ::: "s10", "s11", "scc");
#define CMPX(n) asm volatile("s_mov_b64 s[12:13], exec\n v_cmpx_eq_u32_e32 %0, %1\n s_mov_b64 exec, s[12:13]" : : "s"(q##n), "v"(l##n) : "s12", "s13", "vcc");
#define MIXSF(n) asm volatile("v_cvt_f64_f32 %0, %2\n v_add_f32 %1, %2, %2" : "=v"(d##n), "=v"(a##n) : "v"(a##n));
#define ALIGNB(n) asm volatile("v_alignbit_b32 %0, %1, %1, 4" : "=v"(a##n) : "v"(a##n));
#define ORB(n) asm volatile("v_or_b32 %0, %1, %1" : "=v"(a##n) : "v"(a##n));
#define MAX64(n) asm volatile("v_max_f64 %0, %1, %1" : "=v"(d##n) : "v"(d##n));
#define SCSEL64(n) asm volatile("s_cselect_b64 s[10:11], s[10:11], s[12:13]" ::: "s10", "s11");
#define SBITC(n) asm volatile("s_bitcmp1_b32 %0, 0\n s_cselect_b32 %0, %0, 3" : "+s"(q##n) : : "scc");
#define MOVS(n) asm volatile("v_mov_b32 %0, %1" : "=v"(a##n) : "s"(q##n));
#define ADDFS(n) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a##n) : "s"(q##n), "v"(a##n));
#define SNOP(n) asm volatile("s_nop 0");
#define SWAIT(n) asm volatile("s_waitcnt vmcnt(0)");
#define SBR(n) asm volatile("s_cmp_eq_u32 %0, 77\n s_cbranch_scc1 1f\n 1:" : : "s"(q##n) : "scc");
#define SBRT(n) asm volatile("s_branch 1f\n s_nop 0\n 1:");
#define SUBCOADD(n) asm volatile("v_sub_co_u32_e32 %0, vcc, %1, %1\n v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "=&v"(a##n) : "v"(a0) : "vcc");
#define ADD64S(n) asm volatile("v_add_f64 %0, s[10:11], %1" : "=v"(d##n) : "v"(d##n));
#define MULLIT(n) asm volatile("v_mul_f32_e32 %0, 0x457a0000, %1" : "=v"(a##n) : "v"(a##n));
#define MULSABS(n) asm volatile("v_mul_f32_e64 %0, |%1|, s10" : "=v"(a##n) : "v"(a##n) : "s10");
#define CVTUABS(n) asm volatile("v_cvt_u32_f32_e64 %0, |%1|" : "=v"(a##n) : "v"(a##n));
#define ADDLIT(n) asm volatile("v_add_f32_e32 %0, 0x457a0000, %1" : "=v"(a##n) : "v"(a##n));
#define ANDLIT(n) asm volatile("v_and_b32_e32 %0, 0xfffffffc, %1" : "=v"(a##n) : "v"(a##n));
#define ANDINL(n) asm volatile("v_and_b32_e32 %0, -4, %1" : "=v"(a##n) : "v"(a##n));
#define RDLNM0(n) asm volatile("v_readlane_b32 %0, %1, m0" : "=s"(s0) : "v"(a##n) : "m0");
#define RDLNM0S(n) asm volatile("s_mov_b32 m0, %2\n v_readlane_b32 %0, %1, m0" : "=s"(s0) : "v"(a##n), "s"(q##n) : "m0");
#define RDFIRST(n) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(a##n));
#define MAX3F(n) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a##n) : "v"(a##n), "v"(a0), "v"(a1));
// ---- round 6: a slow-class op and a fast-class op ALTERNATING (do the classes share one port, or does the short op issue in the long one's shadow?)
#define FB(n) "v_fma_f32 %1, %1, %1, %1\n"
#define PAIR(NAME, SLOW) asm volatile(SLOW "\n v_fma_f32 %1, %1, %1, %1" : "+v"(a##NAME), "+v"(b##NAME));
#define PX_MAX3(n) asm volatile("v_max3_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(a##n), "+v"(b##n));
#define PX_CMP64(n) asm volatile("v_cmp_eq_f32_e64 s[10:11], %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(a##n), "+v"(b##n) : : "s10", "s11");
#define PX_CMP32(n) asm volatile("v_cmp_eq_f32_e32 vcc, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(a##n), "+v"(b##n) : : "vcc");
#define PX_ADDC64(n) asm volatile("v_addc_co_u32_e64 %0, s[12:13], %0, %0, s[10:11]\n v_fma_f32 %1, %1, %1, %1" : "+v"(a##n), "+v"(b##n) : : "s12", "s13");
#define PX_CND64(n) asm volatile("v_cndmask_b32_e64 %0, %0, %2, s[10:11]\n v_fma_f32 %1, %1, %1, %1" : "+v"(a##n), "+v"(b##n) : "v"(a0));
#define PX_DPP(n) asm volatile("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_fma_f32 %1, %1, %1, %1" : "+v"(a##n), "+v"(b##n));
#define PX_ADD64(n) asm volatile("v_add_f64 %0, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(d##n), "+v"(b##n));
#define PX_CVT32(n) asm volatile("v_cvt_f32_f64 %0, %2\n v_fma_f32 %1, %1, %1, %1" : "=v"(a##n), "+v"(b##n) : "v"(d##n));
#define PX_CVT64(n) asm volatile("v_cvt_f64_f32 %0, %2\n v_fma_f32 %1, %1, %1, %1" : "=v"(d##n), "+v"(b##n) : "v"(a##n));
#define PX_RDLNS(n) asm volatile("v_readlane_b32 %0, %2, %3\n v_fma_f32 %1, %1, %1, %1" : "=s"(s0), "+v"(b##n) : "v"(a##n), "s"(q##n));
// one slow op, TWO fast ones
#define P2_MAX3(n) asm volatile("v_max3_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_add_f32 %2, %2, %2" : "+v"(a##n), "+v"(b##n), "+v"(p##n.x));
#define P2_ADD64(n) asm volatile("v_add_f64 %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_add_f32 %2, %2, %2" : "+v"(d##n), "+v"(b##n), "+v"(a##n));
#define P2_CMP64(n) asm volatile("v_cmp_eq_f32_e64 s[10:11], %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_add_f32 %2, %2, %2" : "+v"(a##n), "+v"(b##n), "+v"(p##n.x) : : "s10", "s11");
// two DIFFERENT slow ops alternating: the reference's fp64 sums beside the compare / select / carry kind
#define SS_ADD64_CMP64(n) asm volatile("v_add_f64 %0, %0, %0\n v_cmp_eq_f32_e64 s[10:11], %1, %1" : "+v"(d##n) : "v"(a##n) : "s10", "s11");
#define SS_ADD64_MAX3(n) asm volatile("v_add_f64 %0, %0, %0\n v_max3_f32 %1, %1, %1, %1" : "+v"(d##n), "+v"(a##n));
#define SS_ADD64_ADDC(n) asm volatile("v_add_f64 %0, %0, %0\n v_addc_co_u32_e64 %1, s[12:13], %1, %1, s[10:11]" : "+v"(d##n), "+v"(a##n) : : "s12", "s13");
#define SS_ADD64_CND(n) asm volatile("v_add_f64 %0, %0, %0\n v_cndmask_b32_e64 %1, %1, %2, s[10:11]" : "+v"(d##n), "+v"(a##n) : "v"(b0));
#define SS_ADD64_DPP(n) asm volatile("v_add_f64 %0, %0, %0\n v_mov_b32_dpp %1, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(d##n), "+v"(a##n));
#define SS_CVT64_CMP64(n) asm volatile("v_cvt_f64_f32 %0, %1\n v_cmp_eq_f32_e64 s[10:11], %2, %2" : "=v"(d##n) : "v"(b##n), "v"(a##n) : "s10", "s11");
#define SS_CVT32_ADDC(n) asm volatile("v_cvt_f32_f64 %0, %2\n v_addc_co_u32_e64 %1, s[12:13], %1, %1, s[10:11]" : "=v"(b##n), "+v"(a##n) : "v"(d##n) : "s12", "s13");
#define SS_CMP64_ADDC(n) asm volatile("v_cmp_eq_f32_e64 s[14:15], %0, %0\n v_addc_co_u32_e64 %1, s[12:13], %1, %1, s[10:11]" : : "v"(b##n), "v"(a##n) : "s12", "s13", "s14", "s15");
#define SS_MAX3_CMP64(n) asm volatile("v_max3_f32 %0, %0, %0, %0\n v_cmp_eq_f32_e64 s[14:15], %1, %1" : "+v"(a##n) : "v"(b##n) : "s14", "s15");
        if (OP == 0) { REP8(CVT64) REP8(CVT64) }
        if (OP == 1) { REP8(CVT32) REP8(CVT32) }
        if (OP == 2) { REP8(ADD64) REP8(ADD64) }
        if (OP == 3) { REP8(FMA32) REP8(FMA32) }
        if (OP == 4) { REP8(PKFMA) REP8(PKFMA) }
        if (OP == 5) { REP8(PKMUL) REP8(PKMUL) }
        if (OP == 6) { REP8(MAX3) REP8(MAX3) }
        if (OP == 7) { REP8(CNDM) REP8(CNDM) }
        if (OP == 8) { REP8(DPPM) REP8(DPPM) }
        if (OP == 9) { REP8(RDLN) REP8(RDLN) }
        if (OP == 10) { REP8(CMPF) REP8(CMPF) }
        if (OP == 11) { REP8(FMA64) REP8(FMA64) }
        if (OP == 12) { REP8(ADDF) REP8(ADDF) }
        if (OP == 13) { REP8(CND64) REP8(CND64) }
        if (OP == 26) { REP8(CNDE64V) REP8(CNDE64V) }
        if (OP == 27) { REP8(ADDC) REP8(ADDC) }
        if (OP == 28) { asm volatile("s_mov_b64 vcc, s[10:11]" ::: "vcc"); REP8(CNDM) REP8(CNDM) }
        if (OP == 29) { REP8(CNDMIX) }
        if (OP == 30) { REP8(CNDMIXS) }
        if (OP == 31) { C1CND4 C1CND4 C1CND4 }
        if (OP == 32) { C1CND4S C1CND4S C1CND4S }
        if (OP == 33) { C1CND4E C1CND4E C1CND4E }
        if (OP == 34) { C1X4CND C1X4CND C1X4CND }
        if (OP == 35) { REP8(SADD) REP8(SADD) }
        if (OP == 36) { REP8(SMIX) }
        if (OP == 37) { REP8(SCMP) }
        if (OP == 40) { REP8(CVTU) REP8(CVTU) }
        if (OP == 41) { REP8(CVTI32) REP8(CVTI32) }
        if (OP == 42) { REP8(MINU) REP8(MINU) }
        if (OP == 43) { REP8(MINF) REP8(MINF) }
        if (OP == 44) { REP8(MULF) REP8(MULF) }
        if (OP == 45) { REP8(MULFA) REP8(MULFA) }
        if (OP == 46) { REP8(LSHLR) REP8(LSHLR) }
        if (OP == 47) { REP8(ANDB) REP8(ANDB) }
        if (OP == 48) { REP8(DSRD) REP8(DSRD) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 49) { REP8(MED3) REP8(MED3) }
        if (OP == 50) { REP8(MAXI) REP8(MAXI) }
        if (OP == 51) { REP8(SUBF) REP8(SUBF) }
        if (OP == 52) { REP8(ADDU) REP8(ADDU) }
        if (OP == 53) { REP8(LSHLADD) REP8(LSHLADD) }
        if (OP == 60) { REP8(SUBCO32) REP8(SUBCO32) }
        if (OP == 61) { REP8(SUBCO64) REP8(SUBCO64) }
        if (OP == 62) { REP8(ADDC64) REP8(ADDC64) }
        if (OP == 63) { TRACE8 TRACE8 }
        if (OP == 64) { REP8(RDLNS) REP8(RDLNS) }
        if (OP == 65) { REP8(SLSHL64) REP8(SLSHL64) }
        if (OP == 66) { REP8(SOR64) REP8(SOR64) }
        if (OP == 67) { REP8(CMPX) REP8(CMPX) }
        if (OP == 68) { REP8(MIXSF) }
        if (OP == 69) { REP8(ALIGNB) REP8(ALIGNB) }
        if (OP == 70) { REP8(ORB) REP8(ORB) }
        if (OP == 71) { REP8(MAX64) REP8(MAX64) }
        if (OP == 73) { REP8(SCSEL64) REP8(SCSEL64) }
        if (OP == 74) { REP8(SBITC) }
        if (OP == 75) { REP8(MOVS) REP8(MOVS) }
        if (OP == 76) { REP8(ADDFS) REP8(ADDFS) }
        if (OP == 77) { REP8(SNOP) REP8(SNOP) }
        if (OP == 78) { REP8(SWAIT) REP8(SWAIT) }
        if (OP == 79) { REP8(SBR) }
        if (OP == 80) { REP8(SBRT) }
        if (OP == 81) { REP8(SUBCOADD) }
        if (OP == 82) { REP8(ADD64S) REP8(ADD64S) }
        if (OP == 17) { REP8(CNDV2) REP8(CNDV2) }
        if (OP == 18) { REP8(SUBU) REP8(SUBU) }
        if (OP == 19) { REP8(MAXF) REP8(MAXF) }
        if (OP == 20) { REP8(PKADD) REP8(PKADD) }
        if (OP == 21) { REP8(MOVB) REP8(MOVB) }
        if (OP == 22) { REP8(FMA3) REP8(FMA3) }
        if (OP == 23) { REP8(MAX3D) REP8(MAX3D) }
        if (OP == 24) { REP8(CMPS) REP8(CMPS) }
        if (OP == 25) { REP8(CMPU) REP8(CMPU) }
        if (OP == 14) { REP8(LSHL) REP8(LSHL) }
        if (OP == 15) { REP8(MUL64) REP8(MUL64) }
        if (OP == 90) { REP8(MULLIT) REP8(MULLIT) }
        if (OP == 91) { REP8(MULSABS) REP8(MULSABS) }
        if (OP == 92) { REP8(CVTUABS) REP8(CVTUABS) }
        if (OP == 93) { REP8(ADDLIT) REP8(ADDLIT) }
        if (OP == 94) { REP8(ANDLIT) REP8(ANDLIT) }
        if (OP == 95) { REP8(ANDINL) REP8(ANDINL) }
        if (OP == 96) { REP8(RDLNM0) REP8(RDLNM0) }
        if (OP == 97) { REP8(RDLNM0S) REP8(RDLNM0S) }
        if (OP == 98) { REP8(RDFIRST) REP8(RDFIRST) }
        if (OP == 16) { REP8(CVTI) REP8(CVTI) }
        if (OP == 100) { REP8(PX_MAX3) }
        if (OP == 101) { REP8(PX_CMP64) }
        if (OP == 102) { REP8(PX_CMP32) }
        if (OP == 103) { REP8(PX_ADDC64) }
        if (OP == 104) { REP8(PX_CND64) }
        if (OP == 105) { REP8(PX_DPP) }
        if (OP == 106) { REP8(PX_ADD64) }
        if (OP == 107) { REP8(PX_CVT32) }
        if (OP == 108) { REP8(PX_CVT64) }
        if (OP == 109) { REP8(PX_RDLNS) }
        if (OP == 110) { REP8(SS_ADD64_CMP64) }
        if (OP == 111) { REP8(SS_ADD64_MAX3) }
        if (OP == 112) { REP8(SS_ADD64_ADDC) }
        if (OP == 113) { REP8(SS_ADD64_CND) }
        if (OP == 114) { REP8(SS_ADD64_DPP) }
        if (OP == 115) { REP8(SS_CVT64_CMP64) }
        if (OP == 116) { REP8(SS_CVT32_ADDC) }
        if (OP == 117) { REP8(SS_CMP64_ADDC) }
        if (OP == 118) { REP8(SS_MAX3_CMP64) }
        if (OP == 120) { P2_MAX3(0) P2_MAX3(1) P2_MAX3(2) P2_MAX3(3) P2_MAX3(4) P2_ADD64(5) }
        if (OP == 121) { P2_ADD64(0) P2_ADD64(1) P2_ADD64(2) P2_ADD64(3) P2_ADD64(4) P2_MAX3(5) }
        if (OP == 122) { P2_CMP64(0) P2_CMP64(1) P2_CMP64(2) P2_CMP64(3) P2_CMP64(4) P2_MAX3(5) }

    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) +
                                          p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + s0 + q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
}

template <int OP> double run(const char* name, int waves_per_simd)
{
    const int blocks = 1024 * waves_per_simd, iters = 16000;
    unsigned long long* d; float* s;
    hipMalloc(&d, blocks * 8); hipMalloc(&s, blocks * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, s, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, s, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    // wall-clock based: instructions per SIMD = waves_per_simd * iters * 16
    const double inst = (double)waves_per_simd * iters * 16;
    printf("%-14s waves/simd=%d  %.3f ms  -> %.2f ns/inst/SIMD  (s_memtime ticks/inst/wave %.2f)\n", name, waves_per_simd, ms,
           ms * 1e6 / inst, avg / (iters * 16.0));
    hipFree(d); hipFree(s);
    return ms;
}

int main(int argc, char** argv)
{
    if (argc > 1 && argv[1][0] == 'l') {          // "l": the literal / modifier forms of round 4 only
        const int w = 8;
        run<44>("v_mul_f32", w); run<90>("v_mul_f32 literal", w); run<91>("v_mul_f32 |v|, s", w); run<40>("v_cvt_u32_f32", w); run<92>("v_cvt_u32_f32 |v| e64", w);
        run<9>("v_readlane const", w); run<64>("v_readlane sgpr sel", w); run<96>("v_readlane m0", w); run<97>("s_mov m0 + v_readlane m0", w); run<98>("v_readfirstlane", w);
        run<93>("v_add_f32 literal", w); run<47>("v_and_b32", w); run<94>("v_and_b32 literal", w); run<95>("v_and_b32 inline -4", w);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'm') {          // "m": round 6, slow-class ops alternating with fast ones and with each other (16 instructions per iteration)
        const int w = 8;
        run<3>("v_fma_f32", w); run<2>("v_add_f64", w); run<6>("v_max3_f32", w); run<24>("v_cmp_e64 sgpr", w); run<62>("v_addc_co e64 sgpr", w);
        run<100>("max3 | fma", w); run<101>("cmp_e64 | fma", w); run<102>("cmp_e32 vcc | fma", w); run<103>("addc_e64 | fma", w); run<104>("cnd_e64 | fma", w);
        run<105>("dpp ror | fma", w); run<106>("add_f64 | fma", w); run<107>("cvt_f32_f64 | fma", w); run<108>("cvt_f64_f32 | fma", w); run<109>("readlane s | fma", w);
        run<110>("add_f64 | cmp_e64", w); run<111>("add_f64 | max3", w); run<112>("add_f64 | addc_e64", w); run<113>("add_f64 | cnd_e64", w); run<114>("add_f64 | dpp", w);
        run<115>("cvt_f64 | cmp_e64", w); run<116>("cvt_f32 | addc_e64", w); run<117>("cmp_e64 | addc_e64", w); run<118>("max3 | cmp_e64", w);
        run<120>("5x(max3,fma,add)+(add64,fma,add) /18", w); run<121>("5x(add64,fma,add)+(max3,..) /18", w); run<122>("5x(cmp64,fma,add)+(max3,..) /18", w);
        return 0;
    }
    for (int w : {8, 4, 2, 1}) {
        if (argc > 1) break;          // any argument: only the round-2 table
        run<12>("v_add_f32", w); run<3>("v_fma_f32", w); run<4>("v_pk_fma_f32", w); run<5>("v_pk_mul_f32", w);
        run<0>("v_cvt_f64_f32", w); run<1>("v_cvt_f32_f64", w); run<16>("v_cvt_f64_u32", w); run<2>("v_add_f64", w);
        run<15>("v_mul_f64", w); run<11>("v_fma_f64", w);
        run<6>("v_max3_f32", w); run<7>("v_cndmask_b32", w); run<8>("v_mov_dpp ror", w); run<9>("v_readlane", w);
        run<10>("v_cmp_eq_f32", w); run<24>("v_cmp_e64 sgpr", w); run<25>("v_cmp_gt_u32", w); run<14>("v_lshl_or_b32", w);
        run<13>("cndmask e64 s", w); run<17>("cndmask 2src vcc", w); run<18>("v_sub_u32", w); run<19>("v_max_f32", w);
        run<26>("cndmask e64 vcc", w); run<27>("v_addc vcc", w); run<28>("cndmask vcc<-salu", w); run<29>("cmp+cnd vcc (x2)", w); run<30>("cmp+cnd sgpr (x2)", w);
        run<31>("1cmp+4cnd e32 (15/16)", w); run<32>("1cmp+4cnd sgpr (15/16)", w); run<33>("1cmp+4cnd e64 vcc", w); run<34>("cmp,3add,cnd e32", w);
        run<35>("s_add_i32", w); run<36>("s_add+v_add (x2)", w); run<37>("s_cmp+s_cselect (x2)", w);
        run<40>("v_cvt_u32_f32", w); run<41>("v_cvt_i32_f32", w); run<42>("v_min_u32", w); run<43>("v_min_f32", w); run<44>("v_mul_f32", w);
        run<45>("v_mul_f32 |abs|", w); run<46>("v_lshlrev_b32", w); run<47>("v_and_b32", w); run<48>("ds_read_b32", w); run<49>("v_med3_i32", w);
        run<50>("v_max_i32", w); run<51>("v_sub_f32", w); run<52>("v_add_u32", w); run<53>("v_lshl_add_u32", w);
        run<20>("v_pk_add_f32", w); run<21>("v_mov_b32", w); run<22>("v_fma_f32 3src", w); run<23>("v_max3 3src", w);
    }
    // round 2 table (per 16 instructions of the named kind; pairs / sequences are marked)
    for (int w : {8}) {
        run<60>("v_sub_co e32 vcc", w); run<61>("v_sub_co e64 sgpr", w); run<62>("v_addc_co e64 sgpr", w); run<63>("4 sub_co + 4 addc (16)", w);
        run<81>("sub_co,addc vcc (x2)", w);
        run<64>("v_readlane sgpr sel", w); run<9>("v_readlane const", w); run<65>("s_lshl_b64", w); run<66>("s_or_b64", w); run<73>("s_cselect_b64", w);
        run<74>("s_bitcmp1+s_cselect (x2)", w); run<67>("s_mov,v_cmpx,s_mov (x3)", w); run<68>("cvt_f64 + add_f32 (x2)", w);
        run<69>("v_alignbit_b32", w); run<70>("v_or_b32", w); run<71>("v_max_f64", w); run<75>("v_mov_b32 v,s", w); run<76>("v_add_f32 s,v", w);
        run<82>("v_add_f64 s,v", w); run<77>("s_nop 0", w); run<78>("s_waitcnt (idle)", w); run<79>("s_cmp+cbranch nt (x2)", w); run<80>("s_branch taken (8)", w);
        run<12>("v_add_f32", w); run<0>("v_cvt_f64_f32", w); run<35>("s_add_i32", w);
    }
    return 0;
}
