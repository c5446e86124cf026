// np_events_kernels.hip -- the stage in front of the event aligner, for gfx950:
//   * scrappie event detection (src/thirdparty/scrappie/event_detection.c:268-319) as SquiggleRead::load_from_raw runs it
//     (src/nanopolish_squiggle_read.cpp:229-236): on the WHOLE raw table -- trim_and_segment_raw's result is discarded
//     there -- with event_detection_defaults;
//   * estimate_scalings_using_mom (src/nanopolish_raw_loader.cpp:30-75) and the aligner's per-read constants
//     (:99-108) on the device, so that raw signal -> events -> alignment -> calibration -> scoring needs no host step.
//
// Parity argument for the detector.  The reference accumulates double-precision prefix sums of the samples and of
// their fp32 squares serially and only ever uses DIFFERENCES of two prefix values (windows of 3 and 6 samples, event
// spans).  Samples are fp32, so all of them are multiples of one power of two g; as long as n * max|x| < 2^53 g every
// partial sum of any subset is representable, every one of the reference's additions is exact, and a difference of
// two prefix values IS the exact sum of the samples in between -- in any order.  np_ed_check_kernel proves that bound
// per read (for the samples and for their squares); reads that pass are segmented from window sums computed directly
// (no prefix arrays, no serial scan), and the result is bit-identical to the reference.  A read that fails the bound
// (samples below ~4 pA in a 130k-sample read would do it) takes the SERIAL path (round 3): its prefix sums are accumulated
// front to back by one lane, exactly the reference's additions with the reference's roundings (np_ed_serial_tstat_kernel,
// np_ed_serial_events_kernel) -- slower, but a fallback for the odd read, bit-identical like the rest.  Only a read with a
// non-finite sample still reports NP_ED_INEXACT; there is no approximate path.
//
// The t-statistics are embarrassingly parallel (np_ed_tstat_kernel: one thread per sample, neighbours through LDS).  The
// short/long peak picker is a sequential state machine per read: short reads take one lane per read (64 reads per wave,
// t-statistics streamed with a four-sample register prefetch); reads of NP_ED_PAR_MIN samples and more are walked as 64
// segments per read, one per lane, with a checked warm-up (np_ed_peaks_par_kernel) -- and with the DNA windows (3 and 6
// samples) each lane computes the t-statistics of its segment on the fly from the raw samples, so that no t-statistic array
// is written or read for them (round 2).  Event means/stdv are one thread per event.
#include <algorithm>
#include "np_kernels.h"
#include "np_log.h"

#define NP_ED_TILE 256
#define NP_ED_HALO 16          // >= the largest window (event_detection_rna: 14)
#define NP_ED_WARMUP 64        // samples a segment of the parallel peak walk starts early (see np_ed_peaks_par_kernel): results never depend on
                               // it; with the t-statistics computed inside the walk a warm-up sample costs as much as a real one, and the rare
                               // repair rounds of a short warm-up cost less than a long one (256 / 128 / 64 / 32: 17.2 / 16.3 / 15.8 / 15.6 ms)
#define NP_ED_PAR_MIN 2048     // reads shorter than this take the lane-per-read walk
#define NP_ED_SERIAL 1         // status of a read whose prefix sums are not provably exact: serial path
#ifndef NP_ED_FUSED
#define NP_ED_FUSED 1          // long reads: t-statistics computed inside the peak walk (DNA windows), no t-statistic array
#endif

namespace {

__device__ __forceinline__ double dbl_readlane(double v, int l)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// ---------------------------------------------------------------------------------------------------------------
// ADC counts -> pA, the conversion the signal loaders apply before SquiggleRead sees the raw table:
//     float signal = rec->raw_signal[i]; rawptr[i] = (signal + offset) * raw_unit;     raw_unit = range / digitisation
// (src/io/nanopolish_fast5_loader.cpp:96-103 for slow5, src/io/nanopolish_fast5_io.cpp:163-165 for fast5: the same fp32
// expression).  Half the bytes of a host-fed batch: int16 up, fp32 on the device.  Block per (read, 1024-sample tile).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) np_adc_to_pa_kernel(int n_reads, const int16_t* __restrict__ adc, const int64_t* __restrict__ raw_off,
                                                            const float* __restrict__ offset, const float* __restrict__ raw_unit,
                                                            float* __restrict__ raw_pa)
{
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const int64_t n = raw_off[r + 1] - raw_off[r];
    const int64_t base = (int64_t)blockIdx.y * 1024;
    if (base >= n) return;
    const float off = offset[r], unit = raw_unit[r];
    const int16_t* in = adc + raw_off[r];
    float* out = raw_pa + raw_off[r];
    // (round 5: four samples per thread -- one 8-byte load, one 16-byte store -- from the first sample of the read whose position in the batch
    //  arrays is a multiple of four: both accesses are then aligned; 2-byte loads and 4-byte stores ran at 3.3 TB/s)
    const int head = (int)((4 - (raw_off[r] & 3)) & 3);                      // samples before that position
    const int64_t tile_lo = base, tile_hi = base + 1024 < n ? base + 1024 : n;
    // the tile [tile_lo, tile_hi) of the read, cut at the aligned positions head + 4 j
    const int64_t j_lo = tile_lo <= head ? 0 : (tile_lo - head + 3) / 4, j_hi = tile_hi <= head ? 0 : (tile_hi - head) / 4;     // whole groups inside the tile
    for (int64_t j = j_lo + threadIdx.x; j < j_hi; j += 256) {
        const int64_t i = head + 4 * j;
        const short4 v = *reinterpret_cast<const short4*>(in + i);
        float4 o;
        o.x = ((float)v.x + off) * unit; o.y = ((float)v.y + off) * unit; o.z = ((float)v.z + off) * unit; o.w = ((float)v.w + off) * unit;
        *reinterpret_cast<float4*>(out + i) = o;
    }
    // what the groups do not cover: before the first group of the tile and after its last
    const int64_t g_begin = j_lo < j_hi ? head + 4 * j_lo : tile_hi, g_end = j_lo < j_hi ? head + 4 * j_hi : tile_hi;
    for (int64_t i = tile_lo + threadIdx.x; i < g_begin; i += 256) out[i] = ((float)in[i] + off) * unit;
    for (int64_t i = g_end + threadIdx.x; i < tile_hi; i += 256) out[i] = ((float)in[i] + off) * unit;
}

// ---------------------------------------------------------------------------------------------------------------
// exactness bound of the prefix sums, one block per read
// ---------------------------------------------------------------------------------------------------------------
// bit patterns of non-negative floats order like the floats; inf/nan patterns (>= 0x7f800000) end up in the maximum
struct ed_range {
    uint32_t amax = 0u, amin = 0xffffffffu, qmax = 0u, qmin = 0xffffffffu;      // |x| and x^2 (fp32): largest, smallest non-zero
    __device__ __forceinline__ void take(float v)
    {
        const float q = v * v;
        const uint32_t a = __builtin_bit_cast(uint32_t, v) & 0x7fffffffu, b = __builtin_bit_cast(uint32_t, q);
        amax = a > amax ? a : amax; qmax = b > qmax ? b : qmax;
        if (a != 0u) amin = a < amin ? a : amin;
        if (b != 0u) qmin = b < qmin ? b : qmin;
    }
};
// the block's ranges combined, and read r's verdict (256 threads)
__device__ __forceinline__ void ed_check_finish(const ed_range& R, const int64_t n, const int r, int32_t* __restrict__ status)
{
    __shared__ uint32_t red[4][256];
    red[0][threadIdx.x] = R.amax; red[1][threadIdx.x] = R.amin; red[2][threadIdx.x] = R.qmax; red[3][threadIdx.x] = R.qmin;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = max(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = min(red[1][threadIdx.x], red[1][threadIdx.x + s]);
            red[2][threadIdx.x] = max(red[2][threadIdx.x], red[2][threadIdx.x + s]);
            red[3][threadIdx.x] = min(red[3][threadIdx.x], red[3][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bool ok = true, bad = false;
        for (int w = 0; w < 2; ++w) {
            const uint32_t mx = red[2 * w][0], mn = red[2 * w + 1][0];
            if (mx >= 0x7f800000u) { bad = true; continue; }               // inf / nan
            if (mn == 0xffffffffu) continue;                                // all zero
            if ((mn >> 23) == 0u) { ok = false; continue; }                 // a denormal term: no bound attempted
            if (w == 0 && mn < 0x30800000u) { ok = false; continue; }       // a non-zero |sample| below 2^-30: the fused walk's two-operation
                                                                            // divisions (div_small_f32) want their dividends 0 or >= 2^-100
            const int g = (int)(mn >> 23) - 127 - 23;                       // every term is a multiple of 2^g
            const double bound = (double)n * (double)__builtin_bit_cast(float, mx);
            ok = ok && bound < ldexp(1.0, 53 + g);
        }
        // 0: every addition of the reference is exact (the parallel path); NP_ED_SERIAL: not provably -- the serial path repeats
        // the reference's additions one by one; NP_ED_INEXACT: a non-finite sample (the reference's own result is undefined)
        status[r] = bad ? NP_ED_INEXACT : (ok ? 0 : NP_ED_SERIAL);
    }
}

__global__ void __launch_bounds__(256) np_ed_check_kernel(int n_reads, const float* __restrict__ raw, const int64_t* __restrict__ raw_off,
                                                           int32_t* __restrict__ status)
{
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const float* x = raw + raw_off[r];
    const int64_t n = raw_off[r + 1] - raw_off[r];
    ed_range R;
    // 16-byte loads from the 16-byte boundary at or below the read's first sample (round 5: 4-byte loads ran at 3.5 TB/s); the up to
    // three samples before the read and after it are skipped, not read
    const int mis = (int)((((uintptr_t)x) & 15u) >> 2);
    for (int64_t g = -(int64_t)mis + 4 * (int64_t)threadIdx.x; g < n; g += 1024) {
        if (g >= 0 && g + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4*>(x + g);
            R.take(v.x); R.take(v.y); R.take(v.z); R.take(v.w);
        } else {
            for (int t = 0; t < 4; ++t) if (g + t >= 0 && g + t < n) R.take(x[g + t]);
        }
    }
    ed_check_finish(R, n, r, status);
}

// ADC counts -> pA AND the exactness bound of the samples written, one block per read (round 5): np_adc_to_pa_kernel followed by
// np_ed_check_kernel moved 6 + 4 bytes per sample, this one moves 6 -- the check looks at the values on their way out.  Groups of four
// samples from the first position of the read that is a multiple of four in the batch arrays (8-byte loads, 16-byte stores, aligned).
// write_below (round 6, np_detect_events_adc_dev): the pA values are stored only for reads SHORTER than this many samples -- the longer ones are
// walked and summed straight from the counts (np_ed_peaks_par_kernel<true, true>, np_ed_events_kernel<true>), and their verdict needs no store.
__global__ void __launch_bounds__(256) np_adc_to_pa_check_kernel(int n_reads, const int16_t* __restrict__ adc, const int64_t* __restrict__ raw_off,
                                                                  const float* __restrict__ offset, const float* __restrict__ raw_unit,
                                                                  float* __restrict__ raw_pa, int32_t* __restrict__ status, int64_t write_below)
{
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const int64_t n = raw_off[r + 1] - raw_off[r];
    const bool wr = n < write_below;
    const float off = offset[r], unit = raw_unit[r];
    const int16_t* in = adc + raw_off[r];
    float* out = raw_pa + raw_off[r];
    ed_range R;
    const int64_t head = (int64_t)((4 - (raw_off[r] & 3)) & 3) < n ? (int64_t)((4 - (raw_off[r] & 3)) & 3) : n;      // samples before the first aligned group
    const int64_t groups = (n - head) / 4;
    for (int64_t j = threadIdx.x; j < groups; j += 256) {
        const int64_t i = head + 4 * j;
        const short4 v = *reinterpret_cast<const short4*>(in + i);
        float4 o;
        o.x = ((float)v.x + off) * unit; o.y = ((float)v.y + off) * unit; o.z = ((float)v.z + off) * unit; o.w = ((float)v.w + off) * unit;
        if (wr) *reinterpret_cast<float4*>(out + i) = o;
        R.take(o.x); R.take(o.y); R.take(o.z); R.take(o.w);
    }
    // what the groups do not cover: up to three samples before the first and after the last
    for (int64_t i = threadIdx.x; i < head; i += 256) { const float o = ((float)in[i] + off) * unit; if (wr) out[i] = o; R.take(o); }
    for (int64_t i = head + 4 * groups + threadIdx.x; i < n; i += 256) { const float o = ((float)in[i] + off) * unit; if (wr) out[i] = o; R.take(o); }
    ed_check_finish(R, n, r, status);
}
// ... and the reads of at least `from` samples that the verdict sends to the SERIAL path get their pA values after all (the serial kernels read them)
__global__ void __launch_bounds__(256) np_adc_to_pa_serial_kernel(int n_reads, const int16_t* __restrict__ adc, const int64_t* __restrict__ raw_off,
                                                                   const float* __restrict__ offset, const float* __restrict__ raw_unit,
                                                                   float* __restrict__ raw_pa, const int32_t* __restrict__ status, int64_t from)
{
    const int r = blockIdx.x;
    if (r >= n_reads || status[r] != NP_ED_SERIAL) return;
    const int64_t n = raw_off[r + 1] - raw_off[r];
    if (n < from) return;
    const float off = offset[r], unit = raw_unit[r];
    const int16_t* in = adc + raw_off[r];
    float* out = raw_pa + raw_off[r];
    for (int64_t i = threadIdx.x; i < n; i += 256) out[i] = ((float)in[i] + off) * unit;
}

// ---------------------------------------------------------------------------------------------------------------
// compute_tstat (event_detection.c:63-119) for both windows, one thread per sample
// ---------------------------------------------------------------------------------------------------------------
// x / d for a divisor whose correctly rounded reciprocal r is known (the window length: two values per launch): the
// Markstein sequence of np_div_exact, in double and in float.  ~5 multiply-adds instead of the ~40-instruction IEEE
// division expansion; tests/test_gpu_events.py compares every t-statistic with the reference's bit for bit.
__device__ __forceinline__ double div_exact_f64(double n, double d, double r)
{
    double q = n * r;
    double e = __builtin_fma(-d, q, n);
    q = __builtin_fma(e, r, q);
    e = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}

// sd / sq: the tile's samples and their fp32 squares, widened to double once per sample; the window sums are then
// plain (exact) double additions.  [lo, hi) relative to the centre c.
__device__ __forceinline__ void window_sums(const double* __restrict__ sd, const double* __restrict__ sq, int c, int lo, int hi,
                                            double& s, double& q)
{
    s = 0.0; q = 0.0;
    for (int j = lo; j < hi; ++j) { s += sd[c + j]; q += sq[c + j]; }
}

__device__ __forceinline__ float tstat_from_sums(double sum1, double sumsq1, double sum2d, double sumsq2d, int64_t i, int64_t n, int w)
{
    if (w < 2 || n < 2 * (int64_t)w || i < w || i > n - w) return 0.0f;        // quick return and fudged boundaries
    const float w_lengthf = (float)w;
    const float sum2 = (float)sum2d, sumsq2 = (float)sumsq2d;
    const double wd = (double)w_lengthf, rwd = 1.0 / wd;                      // (uniform: once per wave)
    const float rwf = (float)rwd;                                             // RN(1/w) in fp32: 1/w is not a rounding tie
    const float mean1 = (float)div_exact_f64(sum1, wd, rwd);
    const float mean2 = np_div_exact(sum2, w_lengthf, rwf);
    float combined_var = (float)(div_exact_f64(sumsq1, wd, rwd) - (double)(mean1 * mean1) + (double)np_div_exact(sumsq2, w_lengthf, rwf) -
                                 (double)(mean2 * mean2));
    combined_var = fmaxf(combined_var, 1.17549435e-38f);                       // FLT_MIN
    const float delta_mean = mean2 - mean1;
    // (a variance clamped to FLT_MIN has a denormal quotient, outside what the correction steps cover: plain division there)
    const float cvw = combined_var < 1e-30f ? combined_var / w_lengthf : np_div_exact(combined_var, w_lengthf, rwf);
    return (float)(fabs((double)delta_mean) / sqrt((double)cvw));
}

__global__ void __launch_bounds__(NP_ED_TILE) np_ed_tstat_kernel(int n_reads, const float* __restrict__ raw, const int64_t* __restrict__ raw_off,
                                                                 const int32_t* __restrict__ status, int w1, int w2, float2* __restrict__ tstat,
                                                                 int64_t skip_from)
{
    const int r = blockIdx.x;                               // (reads on x: the y extent of a grid stops at 65535)
    if (r >= n_reads || status[r] != 0) return;
    const int64_t n = raw_off[r + 1] - raw_off[r];
    if (n >= skip_from) return;                             // served by the fused walk (np_ed_peaks_par_kernel<true>)
    const int64_t base = (int64_t)blockIdx.y * NP_ED_TILE;
    if (base >= n) return;
    const float* x = raw + raw_off[r];
    __shared__ double sd[NP_ED_TILE + 2 * NP_ED_HALO], sq[NP_ED_TILE + 2 * NP_ED_HALO];
    for (int t = threadIdx.x; t < NP_ED_TILE + 2 * NP_ED_HALO; t += NP_ED_TILE) {
        const int64_t i = base - NP_ED_HALO + t;
        const float v = (i >= 0 && i < n) ? x[i] : 0.0f;
        sd[t] = (double)v; sq[t] = (double)(v * v);          // fp32 product, as the reference's sumsq (:47)
    }
    __syncthreads();
    const int64_t i = base + threadIdx.x;
    if (i >= n) return;
    const int c = NP_ED_HALO + (int)threadIdx.x;
    // the inner window's sums are part of the outer window's (every addition here is exact, so regrouping is free)
    const int wa = w1 < w2 ? w1 : w2, wb = w1 < w2 ? w2 : w1;
    double la, lqa, ra, rqa, lx, lqx, rx, rqx;
    window_sums(sd, sq, c, -wa, 0, la, lqa); window_sums(sd, sq, c, 0, wa, ra, rqa);
    window_sums(sd, sq, c, -wb, -wa, lx, lqx); window_sums(sd, sq, c, wa, wb, rx, rqx);
    const float ta = tstat_from_sums(la, lqa, ra, rqa, i, n, wa);
    const float tb = tstat_from_sums(lx + la, lqx + lqa, ra + rx, rqa + rqx, i, n, wb);
    tstat[raw_off[r] + i] = w1 < w2 ? make_float2(ta, tb) : make_float2(tb, ta);
}


// ---------------------------------------------------------------------------------------------------------------
// The serial path of a read that failed the exactness bound: compute_sum_sumsq (event_detection.c:35-49) as the reference runs
// it -- sum[i + 1] = sum[i] + data[i], sumsq[i + 1] = sumsq[i] + data[i] * data[i], front to back in double -- by lane 0, 64
// samples at a time into an LDS ring, and compute_tstat (:63-119) from DIFFERENCES of those prefix values by all 64 lanes.
// One wave per read.  The t-statistics go to the batch's t-statistic array; the peak walk then reads them like any other read's.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) np_ed_serial_tstat_kernel(int n_reads, const float* __restrict__ raw, const int64_t* __restrict__ raw_off,
                                                                const int32_t* __restrict__ status, int w1, int w2, float2* __restrict__ tstat)
{
    const int r = blockIdx.x;
    if (r >= n_reads || status[r] != NP_ED_SERIAL) return;
    const int lane = threadIdx.x;
    const float* x = raw + raw_off[r];
    const int64_t n = raw_off[r + 1] - raw_off[r];
    __shared__ double S[256], Q[256];                 // prefix values sum[j], sumsq[j] at ring slot j & 255
    double s = 0.0, q = 0.0;                          // (lane 0) the running prefix values
    int64_t have = 0;                                 // prefix values sum[0 .. have] are in the ring
    if (lane == 0) { S[0] = 0.0; Q[0] = 0.0; }
    const int wa = w1 < w2 ? w1 : w2, wb = w1 < w2 ? w2 : w1;
    for (int64_t base = 0; base < n; base += 64) {
        // extend the prefix to sum[min(n, base + 64 + wb)]: at most 64 + NP_ED_HALO new values, while sum[base - wb ..] stay in the ring
        const int64_t want = base + 64 + wb < n ? base + 64 + wb : n;
        __syncthreads();
        if (lane == 0) {
            for (int64_t j = have; j < want; ++j) {
                const float v = x[j];
                s = s + (double)v; q = q + (double)(v * v);
                S[(j + 1) & 255] = s; Q[(j + 1) & 255] = q;
            }
        }
        have = want;
        __syncthreads();
        const int64_t i = base + lane;
        if (i < n) {
            float t[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int w = k ? wb : wa;
                float tv = 0.0f;
                if (!(w < 2 || n < 2 * (int64_t)w || i < w || i > n - w)) {
                    double sum1 = S[i & 255], sumsq1 = Q[i & 255];
                    if (i > w) { sum1 -= S[(i - w) & 255]; sumsq1 -= Q[(i - w) & 255]; }
                    const double sum2d = S[(i + w) & 255] - S[i & 255], sumsq2d = Q[(i + w) & 255] - Q[i & 255];
                    tv = tstat_from_sums(sum1, sumsq1, sum2d, sumsq2d, i, n, w);
                }
                t[k] = tv;
            }
            tstat[raw_off[r] + i] = w1 < w2 ? make_float2(t[0], t[1]) : make_float2(t[1], t[0]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// short_long_peak_detector (event_detection.c:126-207), one lane per read
// ---------------------------------------------------------------------------------------------------------------

// One lane per read.  The two detectors' nested branches are flattened into predicates (every lane of the wave is in a
// different state, so a branchy body would execute all of its paths anyway):
//   CASE 1 (no maximum yet):  a deeper minimum, or a rise of more than peak_height that starts a peak;
//   CASE 2 (in a peak):       a higher maximum; [short detector] masking of the long one once it is going to fire;
//                             the fall that validates the peak; emission once the peak is window/2 samples behind.
// ---- round 5: the t-statistic's last two operations, (float)(|dm| / sqrt(cvw)) in double, without the compiler's range scaling ----------
// hipcc expands a double sqrt into the rsq + Goldschmidt sequence below WRAPPED in a scale-by-2^256 for arguments under 2^-767 and a
// class test for 0 / inf, and a double division into div_scale x 2 + rcp + the same Newton steps + div_fmas + div_fixup: ~16 + ~12
// instructions.  The arguments here are a float's worth of range -- cvw = a float >= FLT_MIN / 14 widened to double, dm a float -- so no
// intermediate can leave the normal doubles and the wrappers have nothing to do; the core sequences are the ones whose last step is an
// exact residual correction (correctly rounded: the same values the wrapped expansions produce; tests/test_gpu_events.py compares the
// detected events of the fused walk with the reference's bit for bit, denormal-variance and near-zero-sample reads included).
__device__ __forceinline__ double sqrt_f64_normal(double x)       // x > 0, normal, far from the range's ends
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double div_f64_normal(double a, double b)     // b > 0 normal; a >= 0 (0 allowed); the quotient a normal double or 0
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// x / w for a window length w = 2^a w' (w' odd, small), correctly rounded, in TWO operations (round 6): q = RN(x ch + RN(x cl)) with
// ch = RN(1 / w), cl = RN(1 / w - ch).  Why that is the IEEE quotient: scale x / w so that its unit in the last place is 1; the rounding
// boundaries are the half-odd-integers (2k + 1) / 2, and x / w - (2k + 1) / 2 = (2^(t+1) X - w' (2k + 1)) / (2 w') for integers X, t, k --
// an even number minus an odd one over 2 w' -- so the exact quotient is never closer than 1 / (2 w') of a unit to a boundary (1/6 for the
// DNA windows 3 and 6), while x ch + RN(x cl) differs from it by ~3 * 2^-p units (p = 24 or 53): both lie strictly between the same two
// boundaries and round alike.  (The general Markstein sequence of np_div_exact needs five operations because it has no such gap to rely
// on.)  Holds while RN(x cl) keeps its relative precision: for fp32, |x| >= 2^-100; in exact arithmetic on the host, over 10^6 random and
// boundary-adjacent operands per w in {3, 5, 6, 7, 9, 11, 13, 14}: no difference, the largest failing |x| for w = 3 is 9.4e-38.  On the
// device np_selftest_division_small compares the fp32 form with the IEEE divide for EVERY float and the fp64 form on 2^30 doubles.
// (A zero dividend gives a zero quotient whose SIGN follows the constants' signs, not the dividend's: see the self-test.)
__device__ __forceinline__ float div_small_f32(float x, float ch, float cl) { return __builtin_fmaf(x, ch, x * cl); }
__device__ __forceinline__ double div_small_f64(double x, double ch, double cl) { return __builtin_fma(x, ch, x * cl); }
#define NP_ED_W3_CH 0x1.5555555555555p-2
#define NP_ED_W3_CL 0x1.5555555555555p-56
#define NP_ED_W3_CHF 0x1.555556p-2f
#define NP_ED_W3_CLF (-0x1.555556p-27f)

// tstat_from_sums for the fused walk (windows of 3 and 6 samples: `half` = 1.0 / 0.5 scales the constants of 1/3 exactly): the same arithmetic
// with (1) the two wrappers above, (2) the divisions by the window length in their two-operation form -- the float dividends sum2, sumsq2 are
// 0 or at least 2^-53 for every read the walk sees (ed_check_finish sends reads with a non-zero sample below 2^-30 to the serial path),
// (3) the plain division of a variance clamped to FLT_MIN (a denormal quotient) taken only when SOME lane of the wave has such a sample
// -- a wave-uniform branch instead of an IEEE division expanded next to every exact one.
// ---- round 6, late: the t-statistic's last step, (float)(|dm| / sqrt(cvw)) evaluated in double (event_detection.c:111), FILTERED --------------
// The reference's value is T = RN32(Qd), Qd = RN64(A / RN64(sqrt(c))) with A = |dm| and c = cvw widened to double: Qd is within 2^-52 (relative)
// of the real R = A / sqrt(c).  q below -- one v_rsq_f64 (good to 2^-23 by the ISA's accuracy statement), ONE Newton step, one product -- is within
// E < 2^-44 of R.  A float rounding boundary is a double whose low 29 significand bits read 0x10000000; if q's low 29 bits are farther than
// 2^14 from that pattern, q is at least 2^14 * 2^-53 = 2^-39 (relative) from every boundary, R and Qd lie on q's side of it, and RN32(q) = T.
// Otherwise (`near`: 2^-14 of the values; also when q is below the normal floats, where the boundaries sit elsewhere) the caller evaluates the
// exact sequence -- for the whole wave, behind a wave-uniform branch.  6 fp64 instructions instead of 18 per window.  np_selftest_tstat_ratio
// compares the two on 2^32 operand pairs and reports how far from a boundary the farthest disagreement of the UNFILTERED q sat (the margin
// the accuracy assumption leaves).
#define NP_ED_RATIO_BAND 16384u
__device__ __forceinline__ float ed_ratio_exact(float dm, float cvw) { return (float)div_f64_normal(fabs((double)dm), sqrt_f64_normal((double)cvw)); }
__device__ __forceinline__ double ed_ratio_approx(float dm, float cvw)
{
    const double A = fabs((double)dm), c = (double)cvw;
    double y = __builtin_amdgcn_rsq(c);
    const double cy = c * y;
    const double r = __builtin_fma(-cy, y, 1.0);          // 1 - c y^2
    y = __builtin_fma(0.5 * y, r, y);                     // y (1 + r / 2)
    return A * y;
}
__device__ __forceinline__ bool ed_ratio_near(double q)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, q);
    const uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
    const bool near_mid = ((lo & 0x1fffffffu) - (0x10000000u - NP_ED_RATIO_BAND)) <= 2u * NP_ED_RATIO_BAND;
    const bool below_normal = (hi - 0x00100000u) < (0x38100000u - 0x00100000u);          // 0 < q < 2^-126 (q >= 0; q == 0 is exact)
    return near_mid || below_normal;
}
__device__ __forceinline__ float ed_ratio_filtered(float dm, float cvw, bool& near)
{
#if defined(NP_ED_RATIO_EXACT) && NP_ED_RATIO_EXACT        // A/B builds: the exact sequence for every value
    near = false;
    return ed_ratio_exact(dm, cvw);
#endif
    const double q = ed_ratio_approx(dm, cvw);
    near = ed_ratio_near(q);
    return (float)q;
}

__device__ __forceinline__ float tstat_from_sums_fast(double sum1, double sumsq1, double sum2d, double sumsq2d, int i, int n, int w, float w_lengthf,
                                                      const double half, const float halff, float& dm_out, float& cvw_out, bool& near)
{
    const double chd = NP_ED_W3_CH * half, cld = NP_ED_W3_CL * half;
    const float chf = NP_ED_W3_CHF * halff, clf = NP_ED_W3_CLF * halff;
    const float sum2 = (float)sum2d, sumsq2 = (float)sumsq2d;
    const float mean1 = (float)div_small_f64(sum1, chd, cld);
    const float mean2 = div_small_f32(sum2, chf, clf);
    float combined_var = (float)(div_small_f64(sumsq1, chd, cld) - (double)(mean1 * mean1) + (double)div_small_f32(sumsq2, chf, clf) -
                                 (double)(mean2 * mean2));
    combined_var = fmaxf(combined_var, 1.17549435e-38f);                       // FLT_MIN
    const float delta_mean = mean2 - mean1;
    float cvw = div_small_f32(combined_var, chf, clf);
    if (__builtin_amdgcn_ballot_w64(combined_var < 1e-30f) != 0ull) cvw = combined_var < 1e-30f ? combined_var / w_lengthf : cvw;
    dm_out = delta_mean; cvw_out = cvw;
    return ed_ratio_filtered(delta_mean, cvw, near);      // (the caller applies compute_tstat's boundary rule and, where `near`, ed_ratio_exact)
}

#if defined(NP_ED_ABL) && (NP_ED_ABL & 4)
#define NP_ED_ABL_NOSTORE && false      /* timing ablation: the per-lane event list is not written */
#else
#define NP_ED_ABL_NOSTORE
#endif
struct detector { int masked_to, peak_pos; float peak_value; int valid_peak; };

template <int K>
__device__ __forceinline__ bool detector_step(detector& dt, detector& other, int i, float current_value, float peak_height, float threshold,
                                              int window_length, int& emitted_pos)
{
    const bool act = !(dt.masked_to >= i);
    const bool inpk = dt.peak_pos != -1;
    const bool c1 = act && !inpk, c2 = act && inpk;
    const bool deeper = c1 && current_value < dt.peak_value;
    const bool rise = c1 && !(current_value < dt.peak_value) && (current_value - dt.peak_value > peak_height);
    const bool higher = c2 && current_value > dt.peak_value;
    dt.peak_value = (deeper || rise || higher) ? current_value : dt.peak_value;
    dt.peak_pos = (rise || higher) ? i : dt.peak_pos;
    const bool over = dt.peak_value > threshold;
    if (K == 0) {
        // the short detector dominates the long one if it is going to fire
        const bool dom = c2 && over;
        other.masked_to = dom ? dt.peak_pos + window_length : other.masked_to;
        other.peak_pos = dom ? -1 : other.peak_pos;
        other.peak_value = dom ? 3.40282347e+38f : other.peak_value;
        other.valid_peak = dom ? 0 : other.valid_peak;
    }
    dt.valid_peak = (c2 && over && (dt.peak_value - current_value > peak_height)) ? 1 : dt.valid_peak;
    const bool emit = c2 && dt.valid_peak != 0 && (i - dt.peak_pos) > window_length / 2;
    emitted_pos = dt.peak_pos;
    dt.peak_pos = emit ? -1 : dt.peak_pos;
    dt.peak_value = emit ? current_value : dt.peak_value;
    dt.valid_peak = emit ? 0 : dt.valid_peak;
    return emit;
}

__global__ void __launch_bounds__(64) np_ed_peaks_kernel(int n_reads, const int64_t* __restrict__ raw_off, const float2* __restrict__ tstat,
                                                          const int32_t* __restrict__ status, np_detector_param p,
                                                          const int64_t* __restrict__ event_off, uint32_t* __restrict__ event_start,
                                                          int32_t* __restrict__ n_events)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    const bool mine = r < n_reads && (raw_off[r + 1] - raw_off[r] < NP_ED_PAR_MIN || status[r] != 0);   // long reads: np_ed_peaks_par_kernel
    const bool live = mine && status[r] >= 0;                   // (NP_ED_SERIAL: t-statistics from the serial kernel)
    const int n = live ? (int)(raw_off[r + 1] - raw_off[r]) : 0;
    const float2* ts = tstat + (live ? raw_off[r] : 0);
    uint32_t* es = event_start + (live ? event_off[r] : 0);
    const int cap = live ? (int)(event_off[r + 1] - event_off[r]) : 0;
    detector d0 = {0, -1, 3.40282347e+38f, 0}, d1 = {0, -1, 3.40282347e+38f, 0};       // DEF_PEAK_POS, DEF_PEAK_VAL = FLT_MAX
    const int w1 = (int)p.window_length1, w2 = (int)p.window_length2;
    int peak_count = 0;
    bool overflow = false;
    if (live && cap > 0) es[0] = 0u;

    // wave-uniform trip count; every lane walks its own read, four samples per prefetched group
    int nmax = n;
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(nmax, o, 64); nmax = v > nmax ? v : nmax; }
    float2 cur[4], nxt[4];
    for (int q = 0; q < 4; ++q) nxt[q] = (q < n) ? ts[q] : make_float2(0.f, 0.f);
    for (int i0 = 0; i0 < nmax; i0 += 4) {
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        for (int q = 0; q < 4; ++q) { const int j = i0 + 4 + q; nxt[q] = (j < n) ? ts[j] : make_float2(0.f, 0.f); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q;
            const bool in = i < n;
            int pos;
            // (a lane past the end of its read is masked by making both detectors inactive: masked_to >= i)
            if (detector_step<0>(d0, d1, in ? i : -1, cur[q].x, p.peak_height, p.threshold1, w1, pos)) {
                peak_count++;
                if (peak_count < cap) es[peak_count] = (uint32_t)pos; else overflow = true;
            }
            if (detector_step<1>(d1, d0, in ? i : -1, cur[q].y, p.peak_height, p.threshold2, w2, pos)) {
                peak_count++;
                if (peak_count < cap) es[peak_count] = (uint32_t)pos; else overflow = true;
            }
        }
    }
    if (mine) {
        // create_events (:243-266): one event more than there are peaks; no peak at all is undefined in the reference
        n_events[r] = !live ? status[r] : (overflow ? NP_ED_OVERFLOW : (peak_count > 0 ? (int32_t)(peak_count + 1) : 0));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same walk, 64 segments of one read at a time (one wave per read).  The detectors forget their past quickly (every
// emission resets one, the short one keeps resetting the long one), so a lane that starts NP_ED_WARMUP samples before
// its segment from the initial state almost always reaches the segment in the true state.  "Almost" is not good enough
// for bit-exact output, so it is checked: the state with which lane j enters its segment must equal the state with
// which lane j-1 left the previous one.  Lane 0 starts from the true initial state, hence by induction every lane whose
// check passes walked its segment exactly as the serial detector does.  A lane whose check fails walks its segment
// again from its left neighbour's exit state (no warm-up), and the check is repeated until nothing changes -- at
// worst 63 rounds, i.e. the serial walk; in practice none.  Emissions are buffered per lane (in the event arrays that
// create_event fills later) and concatenated in lane order, which is emission order.
// ---------------------------------------------------------------------------------------------------------------

struct walk_state { detector d0, d1; };

__device__ __forceinline__ bool same_state(const walk_state& a, const walk_state& b)
{
    return a.d0.masked_to == b.d0.masked_to && a.d0.peak_pos == b.d0.peak_pos && a.d0.valid_peak == b.d0.valid_peak &&
           __builtin_bit_cast(uint32_t, a.d0.peak_value) == __builtin_bit_cast(uint32_t, b.d0.peak_value) &&
           a.d1.masked_to == b.d1.masked_to && a.d1.peak_pos == b.d1.peak_pos && a.d1.valid_peak == b.d1.valid_peak &&
           __builtin_bit_cast(uint32_t, a.d1.peak_value) == __builtin_bit_cast(uint32_t, b.d1.peak_value);
}
__device__ __forceinline__ walk_state shfl_up_state(const walk_state& s)
{
    walk_state o;
    o.d0.masked_to = __shfl_up(s.d0.masked_to, 1, 64); o.d0.peak_pos = __shfl_up(s.d0.peak_pos, 1, 64);
    o.d0.peak_value = __shfl_up(s.d0.peak_value, 1, 64); o.d0.valid_peak = __shfl_up(s.d0.valid_peak, 1, 64);
    o.d1.masked_to = __shfl_up(s.d1.masked_to, 1, 64); o.d1.peak_pos = __shfl_up(s.d1.peak_pos, 1, 64);
    o.d1.peak_value = __shfl_up(s.d1.peak_value, 1, 64); o.d1.valid_peak = __shfl_up(s.d1.valid_peak, 1, 64);
    return o;
}

// walks samples [begin, end) of this lane's read (`trip` = wave-uniform maximum of end - begin); with RECORD, emissions go to
// tmp.  Every lane streams its own segment, so the t-statistics are fetched in whole 64-byte lines: blocks of eight
// samples aligned in the batch-wide array (ts_all, 64-byte aligned; base = the read's first sample in it), four 16-byte
// loads per block, one block prefetched.  (Private streams of 64 lanes x 8 waves do not fit L1: a line must be consumed
// by the loads that miss on it.)
template <bool RECORD>
__device__ __forceinline__ void walk_segment(const float2* __restrict__ ts_all, int64_t base, int begin, int end, int trip, bool lane_active,
                                             walk_state& st, const np_detector_param& p, uint32_t* tmp, int tmp_cap, int& cnt)
{
    const int w1 = (int)p.window_length1, w2 = (int)p.window_length2;
    const int64_t a_begin = base + begin, a_end = base + end;
    int64_t blk = a_begin >> 3;                               // this lane's first block
    const int n_blk = (trip + 7) / 8 + 1;                     // a misaligned range of `trip` samples touches at most this many
    const float4* lines = (const float4*)ts_all;
    float4 cur[4], nxt[4];
    {
        const bool need = lane_active && (blk << 3) < a_end;
        for (int q = 0; q < 4; ++q) nxt[q] = need ? lines[blk * 4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t = 0; t < n_blk; ++t) {
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        {
            const bool need = lane_active && ((blk + 1) << 3) < a_end;
            for (int q = 0; q < 4; ++q) nxt[q] = need ? lines[(blk + 1) * 4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t a = (blk << 3) + q;
            const bool in = lane_active && a >= a_begin && a < a_end;
            const int i = (int)(a - base);
            const float t1 = (q & 1) ? cur[q >> 1].z : cur[q >> 1].x, t2 = (q & 1) ? cur[q >> 1].w : cur[q >> 1].y;
            int pos;
            if (detector_step<0>(st.d0, st.d1, in ? i : -1, t1, p.peak_height, p.threshold1, w1, pos) && RECORD) {
                if (cnt < tmp_cap) tmp[cnt] = (uint32_t)pos;
                cnt++;
            }
            if (detector_step<1>(st.d1, st.d0, in ? i : -1, t2, p.peak_height, p.threshold2, w2, pos) && RECORD) {
                if (cnt < tmp_cap) tmp[cnt] = (uint32_t)pos;
                cnt++;
            }
        }
        blk += 1;
    }
}

// The same walk with the t-statistics computed on the fly from the raw samples (windows 3 and 6, the DNA defaults): no
// t-statistic array is written or read for these reads (8 bytes per sample each way, 115 GB per 100 000 reads).  Each lane
// streams ITS segment of the read in blocks of 8 samples through a range-checked descriptor over the read (samples before the
// first or after the last read as 0, which is what compute_tstat's window sums see there), keeps three blocks in registers
// and slides the eight window sums -- sample and fp32-square sums of the 3- and 6-sample windows left and right of the
// position -- by one sample per step: np_ed_check_kernel has proved every such addition exact, so the slid sums ARE the
// window sums, and tstat_from_sums is the arithmetic of np_ed_tstat_kernel.
#ifndef NP_ED_UNROLL2
#define NP_ED_UNROLL2 0     // 1: two blocks per iteration, ring-indexed sums (round 6 experiment)
#endif
// ADC (round 6): xr describes the read's int16 COUNTS from the 4-byte boundary at or below its first sample (`lead` = 1 when the first sample is the
// upper half of that word); a block of eight samples is four words (five when lead), converted as the signal loaders do: (count + off) * unit.
// Words past the read come back 0 and convert to off * unit where the float form reads 0.0f: both only ever feed t-statistics of positions
// within a window of the read's ends, which compute_tstat fudges to 0.
template <bool RECORD, bool ADC = false>
__device__ __forceinline__ void walk_segment_raw(__amdgpu_buffer_rsrc_t xr, int n, int begin, int end, int trip, bool lane_active,
                                                 walk_state& st, const np_detector_param& p, uint32_t* tmp, int tmp_cap, int& cnt,
                                                 const int lead = 0, const float adc_off = 0.0f, const float adc_unit = 1.0f, const bool ratio_exact = false)
{
    constexpr int WA = 3, WB = 6;
    int blk = begin >> 3;                                     // this lane's first block of 8 samples (read-relative)
    const int n_blk = (trip + 7) / 8 + 1;                     // a misaligned range of `trip` samples touches at most this many
    float W[24];                                              // blocks blk-1, blk, blk+1
    auto load_block = [&](int b, float* dst) {
#ifdef NP_ED_ABL      // timing ablation (results wrong): bit 1 = no sample loads at all, bit 2 = every lane reads the SAME segment (coalesced, cached)
        const bool need = !(NP_ED_ABL & 1) && lane_active && b >= 0 && b * 8 < n;
        if (NP_ED_ABL & 2) b = b % 96;
#else
        const bool need = lane_active && b >= 0 && b * 8 < n;
#endif
        if (ADC) {
            // samples [8 b, 8 b + 8) = bytes [16 b + 2 lead, + 16) of the word-aligned stream
            uint32_t d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u, d4 = 0u;
            if (need) {
                const float4 w = buf_f32x4(xr, 16 * b);
                d0 = __builtin_bit_cast(uint32_t, w.x); d1 = __builtin_bit_cast(uint32_t, w.y); d2 = __builtin_bit_cast(uint32_t, w.z); d3 = __builtin_bit_cast(uint32_t, w.w);
                if (lead) d4 = __builtin_bit_cast(uint32_t, buf_f32(xr, 16 * b + 16));        // (wave-uniform: one read per wave)
            }
            if (lead) {
                d0 = __builtin_amdgcn_alignbit(d1, d0, 16); d1 = __builtin_amdgcn_alignbit(d2, d1, 16);
                d2 = __builtin_amdgcn_alignbit(d3, d2, 16); d3 = __builtin_amdgcn_alignbit(d4, d3, 16);
            }
            const uint32_t d[4] = {d0, d1, d2, d3};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = (float)(int16_t)(d[k] & 0xffffu), c = (float)((int32_t)d[k] >> 16);
                dst[2 * k] = need ? (a + adc_off) * adc_unit : 0.0f;
                dst[2 * k + 1] = need ? (c + adc_off) * adc_unit : 0.0f;
            }
            return;
        }
        const float4 lo = need ? buf_f32x4(xr, 32 * b) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 hi = need ? buf_f32x4(xr, 32 * b + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        dst[0] = lo.x; dst[1] = lo.y; dst[2] = lo.z; dst[3] = lo.w; dst[4] = hi.x; dst[5] = hi.y; dst[6] = hi.z; dst[7] = hi.w;
    };
    load_block(blk - 1, W); load_block(blk, W + 8); load_block(blk + 1, W + 16);
    // Window sums from THREE-SAMPLE sums (round 5).  S(j) = x[j] + x[j+1] + x[j+2] (and Q(j) of the fp32 squares): every one of the eight
    // window sums of sample i is one of them or the sum of two --
    //     left [i-3, i) = S(i-3)     left [i-6, i) = S(i-6) + S(i-3)     right [i, i+3) = S(i)     right [i, i+6) = S(i) + S(i+3)
    // -- and every addition involved is exact (np_ed_check_kernel's bound covers any sum of the read's samples), so the values are the
    // reference's prefix-sum differences whatever the order.  Per sample ONE new S and one new Q (S(j+1) = S(j) - x[j] + x[j+3]) and four
    // additions, instead of eight window sums slid by two operations each.  S[k] / Q[k]: position 8 blk - 6 + k; ten entries are live.
    const int w1 = (int)p.window_length1, w2 = (int)p.window_length2;       // (3, 6) or (6, 3): which statistic feeds which detector
    double S[18], Q[18];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float a = W[2 + k], b = W[3 + k], d = W[4 + k];
        S[k] = (double)a + (double)b + (double)d;
        Q[k] = (double)(a * a) + (double)(b * b) + (double)(d * d);
    }
#if NP_ED_UNROLL2
    // Round 6: two blocks per loop iteration, the three-sample sums in a RING of sixteen (S(j) at index j & 15 relative to the iteration's first
    // block): ten entries are live at any step and the one being formed overwrites one that died six steps earlier, so the sums are never moved --
    // the single-block loop below copies twenty doubles (and the compiler a few more) from S[k + 8] to S[k] after every eight samples, 20 of its
    // ~240 vector instructions per sample are moves.  The raw samples: the steps of a block read x[i + 3] and x[i + 6] only, i.e. the current and
    // the next block; the block after that is in flight.  (Same sums, same order of operations per sum: events bit-identical.)
    double RS[16], RQ[16];
#pragma unroll
    for (int k = 0; k < 10; ++k) { RS[k] = S[k]; RQ[k] = Q[k]; }
    float A[8], B[8], C[8], D[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { A[j] = W[8 + j]; B[j] = W[16 + j]; }
    for (int t = 0; t < n_blk; t += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float* cur = h ? B : A; float* nx = h ? C : B;
            if (h == 0) load_block(blk + 2, C); else load_block(blk + 2, D);          // one block ahead of the window
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int o = 8 * h + q;
                const int i = blk * 8 + q;
                const bool in = lane_active && i >= begin && i < end;
                float dma, cva, dmb, cvb; bool na, nb;
                float ta = tstat_from_sums_fast(RS[(o + 3) & 15], RQ[(o + 3) & 15], RS[(o + 6) & 15], RQ[(o + 6) & 15], i, n, WA, 3.0f, 1.0, 1.0f, dma, cva, na);
                float tb = tstat_from_sums_fast(RS[o & 15] + RS[(o + 3) & 15], RQ[o & 15] + RQ[(o + 3) & 15], RS[(o + 6) & 15] + RS[(o + 9) & 15],
                                                RQ[(o + 6) & 15] + RQ[(o + 9) & 15], i, n, WB, 6.0f, 0.5, 0.5f, dmb, cvb, nb);
                if (ratio_exact || __builtin_amdgcn_ballot_w64(na || nb) != 0ull) { ta = ed_ratio_exact(dma, cva); tb = ed_ratio_exact(dmb, cvb); }
                ta = (n < 2 * WA || i < WA || i > n - WA) ? 0.0f : ta; tb = (n < 2 * WB || i < WB || i > n - WB) ? 0.0f : tb;      // quick return and fudged boundaries
                const float t1 = w1 < w2 ? ta : tb, t2 = w1 < w2 ? tb : ta;
                int pos;
                if (detector_step<0>(st.d0, st.d1, in ? i : -1, t1, p.peak_height, p.threshold1, w1, pos) && RECORD) {
                    if (cnt < tmp_cap NP_ED_ABL_NOSTORE) tmp[cnt] = (uint32_t)pos;
                    cnt++;
                }
                if (detector_step<1>(st.d1, st.d0, in ? i : -1, t2, p.peak_height, p.threshold2, w2, pos) && RECORD) {
                    if (cnt < tmp_cap NP_ED_ABL_NOSTORE) tmp[cnt] = (uint32_t)pos;
                    cnt++;
                }
                // S(i + 4) = S(i + 3) - x[i+3] + x[i+6]
                const float xo = q + 3 < 8 ? cur[q + 3] : nx[q + 3 - 8], xi = q + 6 < 8 ? cur[q + 6] : nx[q + 6 - 8];
                RS[(o + 10) & 15] = RS[(o + 9) & 15] - (double)xo + (double)xi;
                RQ[(o + 10) & 15] = RQ[(o + 9) & 15] - (double)(xo * xo) + (double)(xi * xi);
            }
            blk += 1;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { A[j] = C[j]; B[j] = D[j]; }
    }
#else
    float nxt[8];
    for (int t = 0; t < n_blk; ++t) {
        load_block(blk + 2, nxt);                             // one block ahead of the window
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = blk * 8 + q;
            const bool in = lane_active && i >= begin && i < end;
            float dma, cva, dmb, cvb; bool na, nb;
            float ta = tstat_from_sums_fast(S[q + 3], Q[q + 3], S[q + 6], Q[q + 6], i, n, WA, 3.0f, 1.0, 1.0f, dma, cva, na);
            float tb = tstat_from_sums_fast(S[q] + S[q + 3], Q[q] + Q[q + 3], S[q + 6] + S[q + 9], Q[q + 6] + Q[q + 9], i, n, WB, 6.0f, 0.5, 0.5f, dmb, cvb, nb);
            // (where the filter cannot vouch for a lane -- 2^-14 of the values -- the whole wave takes the exact sequence: it agrees with the
            //  filtered one wherever that was trusted)
            if (ratio_exact || __builtin_amdgcn_ballot_w64(na || nb) != 0ull) { ta = ed_ratio_exact(dma, cva); tb = ed_ratio_exact(dmb, cvb); }
            ta = (n < 2 * WA || i < WA || i > n - WA) ? 0.0f : ta; tb = (n < 2 * WB || i < WB || i > n - WB) ? 0.0f : tb;          // quick return and fudged boundaries
            const float t1 = w1 < w2 ? ta : tb, t2 = w1 < w2 ? tb : ta;
            int pos;
            if (detector_step<0>(st.d0, st.d1, in ? i : -1, t1, p.peak_height, p.threshold1, w1, pos) && RECORD) {
                if (cnt < tmp_cap NP_ED_ABL_NOSTORE) tmp[cnt] = (uint32_t)pos;
                cnt++;
            }
            if (detector_step<1>(st.d1, st.d0, in ? i : -1, t2, p.peak_height, p.threshold2, w2, pos) && RECORD) {
                if (cnt < tmp_cap NP_ED_ABL_NOSTORE) tmp[cnt] = (uint32_t)pos;
                cnt++;
            }
            // S(i + 4) = S(i + 3) - x[i+3] + x[i+6]
            const float xo = W[8 + q + 3], xi = W[8 + q + 6];
            S[q + 10] = S[q + 9] - (double)xo + (double)xi;
            Q[q + 10] = Q[q + 9] - (double)(xo * xo) + (double)(xi * xi);
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) { S[k] = S[k + 8]; Q[k] = Q[k + 8]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) W[j] = W[j + 8];
#pragma unroll
        for (int j = 0; j < 8; ++j) W[16 + j] = nxt[j];
        blk += 1;
    }
#endif
}

template <bool FUSED, bool ADC = false>
__global__ void __launch_bounds__(64, 3) np_ed_peaks_par_kernel(int n_reads, const int64_t* __restrict__ raw_off, const float2* __restrict__ tstat,
                                                              const float* __restrict__ raw, const int16_t* __restrict__ adc,
                                                              const float* __restrict__ adc_offset, const float* __restrict__ adc_unit,
                                                              const int32_t* __restrict__ status, np_detector_param p,
                                                              const int64_t* __restrict__ event_off, uint32_t* __restrict__ event_start,
                                                              uint32_t* __restrict__ scratch_a, uint32_t* __restrict__ scratch_b,
                                                              uint32_t* __restrict__ scratch_c, int32_t* __restrict__ n_events, int warmup, int ratio_exact_mode)
{
    const int r = blockIdx.x;
    if (r >= n_reads) return;
    const int64_t n64 = raw_off[r + 1] - raw_off[r];
    if (n64 < NP_ED_PAR_MIN || status[r] != 0) return;            // short reads / declined reads: the other kernel
    // (ratio_exact_mode: the t-statistic's last step by the exact sequence for every value -- a context whose probe of ed_ratio_filtered
    //  failed at np_create, or the option "ed_ratio_exact")
    const bool ratio_exact = ratio_exact_mode != 0;
    const int n = (int)n64;
    const int lane = threadIdx.x;
    const int64_t base = raw_off[r];
    const int64_t eo = event_off[r];
    const int cap = (int)(event_off[r + 1] - eo);
    uint32_t* es = event_start + eo;
    // per-lane emission buffers: three arrays of `cap` entries, 22 / 21 / 21 lanes each
    const int grp = lane < 22 ? 0 : (lane < 43 ? 1 : 2), idx = lane - (grp == 0 ? 0 : (grp == 1 ? 22 : 43));
    const int tmp_cap = cap / 22;
    uint32_t* tmp = (grp == 0 ? scratch_a : (grp == 1 ? scratch_b : scratch_c)) + eo + (int64_t)idx * tmp_cap;

    const int S = (n + 63) / 64;
    const int start = lane * S < n ? lane * S : n;
    const int end = start + S < n ? start + S : n;
    const int begin = start - warmup > 0 ? start - warmup : 0;
    const detector fresh = {0, -1, 3.40282347e+38f, 0};            // DEF_PEAK_POS, DEF_PEAK_VAL = FLT_MAX
    walk_state st = {fresh, fresh};
    int cnt = 0;
    static_assert(!ADC || FUSED, "the walk reads counts only in its fused form");
    // ADC: the descriptor starts at the 4-byte boundary at or below the read's first count (the batch's count array is 4-byte aligned) and ends
    // at the one at or above its last: every load is word-aligned and none leaves the word that holds the read's last count
    const uintptr_t a0 = ADC ? (uintptr_t)(adc + base) : 0;
    const int lead = ADC ? (int)((a0 >> 1) & 1) : 0;
    const float a_off = ADC ? adc_offset[r] : 0.0f, a_unit = ADC ? adc_unit[r] : 1.0f;
    const __amdgpu_buffer_rsrc_t xr = ADC ? make_rsrc((const void*)(a0 & ~(uintptr_t)3), (uint32_t)((2 * (n + lead) + 3) & ~3))
                                          : make_rsrc(raw + base, (uint32_t)n * 4u);
    if (FUSED) walk_segment_raw<false, ADC>(xr, n, begin, start, warmup, start < end, st, p, tmp, tmp_cap, cnt, lead, a_off, a_unit, ratio_exact);
    else walk_segment<false>(tstat, base, begin, start, warmup, start < end, st, p, tmp, tmp_cap, cnt);   // warm-up, nothing recorded
    walk_state entry = st;                                                                           // state at the segment's first sample
    if (FUSED) walk_segment_raw<true, ADC>(xr, n, start, end, S, start < end, st, p, tmp, tmp_cap, cnt, lead, a_off, a_unit, ratio_exact);
    else walk_segment<true>(tstat, base, start, end, S, start < end, st, p, tmp, tmp_cap, cnt);

    // verification / repair rounds
    for (int round = 0; round < 64; ++round) {
        walk_state left = shfl_up_state(st);
        // an empty segment's exit state is its left neighbour's; make that transitive for the check below
        if (lane > 0 && start >= end) { st = left; entry = left; }
        left = shfl_up_state(st);
        const bool bad = lane > 0 && start < end && !same_state(entry, left);
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) break;
        walk_state redo = left;
        int c2 = 0;
        if (FUSED) walk_segment_raw<true, ADC>(xr, n, start, end, S, bad, redo, p, tmp, tmp_cap, c2, lead, a_off, a_unit, ratio_exact);
        else walk_segment<true>(tstat, base, start, end, S, bad, redo, p, tmp, tmp_cap, c2);
        if (bad) { st = redo; entry = left; cnt = c2; }
    }

    // concatenate in lane order
    int off = cnt;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(off, o, 64); if (lane >= o) off += v; }
    const int total = __shfl(off, 63, 64);
    off -= cnt;
    const bool over = __builtin_amdgcn_ballot_w64(cnt > tmp_cap) != 0ull || total + 1 > cap;
    if (!over) {
        if (lane == 0) es[0] = 0u;
        for (int q = 0; q < cnt; ++q) es[1 + off + q] = tmp[q];
    }
    if (lane == 0) n_events[r] = over ? NP_ED_OVERFLOW : (total > 0 ? total + 1 : 0);
}

// ---------------------------------------------------------------------------------------------------------------
// create_event (event_detection.c:223-241), one thread per event
// ---------------------------------------------------------------------------------------------------------------
// Round 5: the samples reach the events through LDS.  (One thread per event summing its ~5 samples straight from memory -- 4-byte loads, a
// dependent double accumulation per load -- ran at 1.3 TB/s: 14.6 ms per 100 000 reads for 19 GB.)  The workgroup stages a window of
// NP_EV_WIN samples with 16-byte loads, finds how many of the next events lie inside it (events come almost sorted: each thread probes its
// own until one does not fit, the block's minimum is the bound), and every thread sums its events out of LDS; an event that fits no
// window -- longer than the window, or one of the rare inverted pairs of peaks -- is summed from memory by one thread, as before.
// Same additions (every one exact: np_ed_check_kernel), same roundings: event tables bit-identical (tests/test_gpu_events.py).
#ifndef NP_EV_WIN
#define NP_EV_WIN 6144
#endif
__device__ __forceinline__ void ed_event_finish(double s, double q, int64_t start, int64_t end, float* __restrict__ event_length,
                                                float* __restrict__ event_mean, float* __restrict__ event_stdv, int64_t slot)
{
    if (end < start) { s = -s; q = -q; }
    const float length = (float)((uint64_t)end - (uint64_t)start);
    const float mean = (float)s / length;
    const float deltasqr = (float)q;
    const float var = deltasqr / length - mean * mean;
    event_length[slot] = length;
    event_mean[slot] = mean;
    event_stdv[slot] = sqrtf(fmaxf(var, 0.0f));
}
// ADC (round 6): the window is staged from the int16 counts, four per 8-byte load, converted on the way into LDS as the signal loaders do.
template <bool ADC>
__global__ void __launch_bounds__(256) np_ed_events_kernel(int n_reads, const float* __restrict__ raw, const int16_t* __restrict__ adc,
                                                            const float* __restrict__ adc_offset, const float* __restrict__ adc_unit,
                                                            const int64_t* __restrict__ raw_off,
                                                            const int64_t* __restrict__ event_off, const uint32_t* __restrict__ event_start,
                                                            const int32_t* __restrict__ n_events, const int32_t* __restrict__ status,
                                                            float* __restrict__ event_length, float* __restrict__ event_mean, float* __restrict__ event_stdv)
{
    const int r = blockIdx.x;
    if (r >= n_reads || status[r] != 0) return;              // (serial reads: np_ed_serial_events_kernel)
    const int n_ev = n_events[r];
    const float* x = ADC ? nullptr : raw + raw_off[r];
    const int16_t* ax = ADC ? adc + raw_off[r] : nullptr;
    const float a_off = ADC ? adc_offset[r] : 0.0f, a_unit = ADC ? adc_unit[r] : 1.0f;
    auto sample = [&](int64_t i) -> float { return ADC ? ((float)ax[i] + a_off) * a_unit : x[i]; };
    const int64_t n = raw_off[r + 1] - raw_off[r];
    const int64_t eo = event_off[r];
    const uint32_t* es = event_start + eo;
    __shared__ float win[NP_EV_WIN + 4];
    __shared__ int s_bound;
    // (peaks come in the order the two detectors emit them; should a later one lie before an earlier one the reference's unsigned
    //  arithmetic wraps -- sums[end] - sums[start] is then minus the sum in between: lo / hi below, the sign in ed_event_finish)
    auto bounds = [&](int e, int64_t& start, int64_t& end) {
        start = es[e];
        end = e + 1 < n_ev ? (int64_t)es[e + 1] : n;
    };
    // (round 5, second pass: a thread's events of the round -- e0 + thread + 256 k, k < NP_EV_PER -- have their bounds requested BEFORE the
    //  window is staged and kept in registers for the probe and for the sums: one round trip to memory per round instead of three; a round
    //  takes at most 256 NP_EV_PER events)
    constexpr int NP_EV_PER = 6;
    int e0 = 0;
    while (e0 < n_ev) {
        int64_t st0, en0;
        bounds(e0, st0, en0);
        uint32_t b_st[NP_EV_PER], b_en[NP_EV_PER];
#pragma unroll
        for (int k = 0; k < NP_EV_PER; ++k) {
            const int e = e0 + (int)threadIdx.x + 256 * k;
            b_st[k] = e < n_ev ? es[e] : 0u;
            b_en[k] = e + 1 < n_ev ? es[e + 1] : (uint32_t)n;
        }
        const int64_t w0 = st0 < en0 ? st0 : en0;                       // the window starts at the first unprocessed event ...
        const int mis = ADC ? (int)((((uintptr_t)(ax + w0)) & 7u) >> 1)  // ... moved down to an 8-byte boundary of the batch's count array
                            : (int)((((uintptr_t)(x + w0)) & 15u) >> 2);    // ... resp. to a 16-byte boundary of its sample array
        const int64_t wa = w0 - mis, wend = wa + NP_EV_WIN;             // window = samples [wa, wend) of the read (wa may be -1 .. -3 for its first event)
        if (threadIdx.x == 0) s_bound = n_ev < e0 + 256 * NP_EV_PER ? n_ev : e0 + 256 * NP_EV_PER;
        for (int i4 = threadIdx.x; i4 < NP_EV_WIN / 4; i4 += 256) {
            const int64_t g = wa + 4 * (int64_t)i4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g >= 0 && g + 4 <= n) {
                if (ADC) {
                    const short4 c = *reinterpret_cast<const short4*>(ax + g);
                    v = make_float4(((float)c.x + a_off) * a_unit, ((float)c.y + a_off) * a_unit, ((float)c.z + a_off) * a_unit, ((float)c.w + a_off) * a_unit);
                } else v = *reinterpret_cast<const float4*>(x + g);
            } else {
                if (g >= 0 && g < n) v.x = sample(g);
                if (g + 1 >= 0 && g + 1 < n) v.y = sample(g + 1);
                if (g + 2 >= 0 && g + 2 < n) v.z = sample(g + 2);
                if (g + 3 >= 0 && g + 3 < n) v.w = sample(g + 3);
            }
            *reinterpret_cast<float4*>(&win[4 * i4]) = v;
        }
        __syncthreads();
        // how many events from e0 on lie inside the window: every thread probes its own events until one does not fit
#pragma unroll
        for (int k = 0; k < NP_EV_PER; ++k) {
            const int e = e0 + (int)threadIdx.x + 256 * k;
            if (e >= n_ev) break;
            const int64_t st = b_st[k], en = b_en[k];
            const int64_t lo = st < en ? st : en, hi = st < en ? en : st;
            if (lo < wa || hi > wend) { atomicMin(&s_bound, e); break; }
        }
        __syncthreads();
        const int e1 = s_bound;
#pragma unroll
        for (int k = 0; k < NP_EV_PER; ++k) {
            const int e = e0 + (int)threadIdx.x + 256 * k;
            if (e >= e1) break;
            const int64_t st = b_st[k], en = b_en[k];
            const int64_t lo = st < en ? st : en, hi = st < en ? en : st;
            double s = 0.0, q = 0.0;
            for (int i = (int)(lo - wa); i < (int)(hi - wa); ++i) { const float v = win[i]; s += (double)v; q += (double)(v * v); }
            ed_event_finish(s, q, st, en, event_length, event_mean, event_stdv, eo + e);
        }
        if (e1 == e0) {
            // the first event does not fit the window it opens: summed from memory
            if (threadIdx.x == 0) {
                const int64_t lo = st0 < en0 ? st0 : en0, hi = st0 < en0 ? en0 : st0;
                double s = 0.0, q = 0.0;
                for (int64_t i = lo; i < hi; ++i) { const float v = sample(i); s += (double)v; q += (double)(v * v); }
                ed_event_finish(s, q, st0, en0, event_length, event_mean, event_stdv, eo + e0);
            }
            e0 += 1;
        } else e0 = e1;
        __syncthreads();                                               // (the window and the bound are rewritten by the next round)
    }
}


// create_event for a read on the serial path: the prefix sums are accumulated front to back as the reference does, and every
// event takes the difference of the prefix values at its two ends (event_detection.c:223-241).  One lane per read; the events'
// start positions are visited in order, and the scan restarts from the first sample should a start lie before its predecessor
// (the two detectors emit in time order, so that is a corner of a corner).
__global__ void __launch_bounds__(64) np_ed_serial_events_kernel(int n_reads, const float* __restrict__ raw, const int64_t* __restrict__ raw_off,
                                                                 const int64_t* __restrict__ event_off, const uint32_t* __restrict__ event_start,
                                                                 const int32_t* __restrict__ n_events, const int32_t* __restrict__ status,
                                                                 float* __restrict__ event_length, float* __restrict__ event_mean, float* __restrict__ event_stdv)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_reads || status[r] != NP_ED_SERIAL) return;
    const int n_ev = n_events[r];
    const float* x = raw + raw_off[r];
    const int64_t n = raw_off[r + 1] - raw_off[r];
    const int64_t eo = event_off[r];
    double s = 0.0, q = 0.0;          // sum[pos], sumsq[pos]
    int64_t pos = 0;
    auto seek = [&](int64_t to) {     // the prefix values at `to`
        if (to < pos) { s = 0.0; q = 0.0; pos = 0; }
        for (; pos < to; ++pos) { const float v = x[pos]; s = s + (double)v; q = q + (double)(v * v); }
    };
    for (int e = 0; e < n_ev; ++e) {
        const int64_t start = event_start[eo + e];
        const int64_t end = e + 1 < n_ev ? (int64_t)event_start[eo + e + 1] : n;
        seek(start);
        const double s0 = s, q0 = q;
        seek(end);
        const double ds = s - s0, dq = q - q0;
        const float length = (float)((uint64_t)end - (uint64_t)start);
        const float mean = (float)ds / length;
        const float deltasqr = (float)dq;
        const float var = deltasqr / length - mean * mean;
        event_length[eo + e] = length;
        event_mean[eo + e] = mean;
        event_stdv[eo + e] = sqrtf(fmaxf(var, 0.0f));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// estimate_scalings_using_mom (raw_loader.cpp:30-75) + the aligner's per-read constants (:99-108) + set4.
// One wave per read; every sum is accumulated in the reference's order (terms staged 64 at a time in LDS, one lane
// per sum adds its row front to back).
// ---------------------------------------------------------------------------------------------------------------
// Round 5: NP_MOM_R reads per wave.  The three ordered sums are serial chains of one double addition per term -- but a chain needs one
// LANE, and with one read per wave (rounds 2-4) the 64-step serial phase of every chunk ran with one or two of 64 lanes busy and was 95 %
// of the kernel's instructions (8.5 ms per 100 000 reads of the from-raw step).  Now the wave stages a 64-term chunk of EACH of its reads
// (lanes = terms, one LDS tile per read and sum; row stride 65: the serial readers sit on distinct banks) and lane (r, c) adds the 64 terms
// of sum c of read r: one serial phase per NP_MOM_R reads.  A read shorter than its group's longest contributes zero terms past its end (a
// zero term leaves a non-negative-zero sum unchanged: every sum here starts at +0 and adds non-negative or finite terms).  Same terms, same
// order: shift / scale bit-identical (tests/test_gpu_events.py::test_pass_from_raw_signal_matches_oracle).
// (That alone made the kernel SLOWER, 8.5 -> 9.5 ms, as it had the recalibration: the second pass looks level_mean up per k-mer, 64 random
//  128-byte lines per wave instruction out of a table that does not fit the L1.  So the workgroup -- NP_MOM_W waves -- keeps the base
//  model's level_mean in LDS: 32 KB for the 4 096 states of a 6-mer model; a larger model is read from memory.)
#ifndef NP_MOM_R
#define NP_MOM_R 4
#endif
#ifndef NP_MOM_W
#define NP_MOM_W 4
#endif
#define NP_MOM_STATES 4096
#ifndef NP_MOM_D
#define NP_MOM_D 4                       // chunks of 64 terms requested ahead
#endif
__global__ void __launch_bounds__(64 * NP_MOM_W) np_mom_fill_kernel(int n_reads, np_read_dev* __restrict__ reads, np_read_dev* __restrict__ reads_b,
                                                                    const float* __restrict__ event_mean, const int32_t* __restrict__ n_events,
                                                                    const uint16_t* __restrict__ ranks, const np_state_dev* __restrict__ model, int n_states)
{
    constexpr int R = NP_MOM_R;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ double terms_all[NP_MOM_W][R][2][65];
    __shared__ double level[NP_MOM_STATES];
    const bool table = n_states <= NP_MOM_STATES;
    if (table) {
        for (int q = threadIdx.x; q < n_states; q += 64 * NP_MOM_W) level[q] = model[q].level_mean;
        __syncthreads();
    }
    double (*terms)[2][65] = terms_all[wave];
    int ne[R], K[R];
    const float* ev[R]; const uint16_t* rk[R];
    int max_ne = 0, max_K = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int r = (blockIdx.x * NP_MOM_W + wave) * R + q;
        ne[q] = 0; K[q] = 0; ev[q] = event_mean; rk[q] = ranks;
        if (r < n_reads) {
            const int v = __builtin_amdgcn_readfirstlane(n_events[r]);
            ne[q] = v > 0 ? v : 0;
            K[q] = __builtin_amdgcn_readfirstlane((int)reads[r].n_kmers);
            ev[q] = event_mean + reads[r].event_off; rk[q] = ranks + reads[r].rank_off;
        }
        max_ne = ne[q] > max_ne ? ne[q] : max_ne; max_K = K[q] > max_K ? K[q] : max_K;
    }
    // serial-phase roles: one sum per read (passes 1 and 3): lane q < R owns read q; two sums per read (pass 2): lane 2 q + c
    const int r1 = lane < R ? lane : 0, r2 = lane < 2 * R ? lane >> 1 : 0, c2 = lane & 1;
    auto fence = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    // (round 5, second pass: the terms of the NEXT NP_MOM_D chunks are requested before the current ones are summed.  Chunk by chunk -- load,
    //  wait, stage, sum -- every one of a read's ~335 chunks exposed a full round trip to memory with four or eight lanes busy: 81 % of the
    //  wave-cycles waiting, 7.4 ms per 100 000 reads for 0.3 ms of additions.)
    constexpr int D = NP_MOM_D;
    auto load_events = [&](const int base, float (&dst)[D][R]) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = base + 64 * d + lane;
#pragma unroll
            for (int q = 0; q < R; ++q) dst[d][q] = i < ne[q] ? ev[q][i] : 0.0f;
        }
    };
    float cur[D][R], nxt[D][R];
    double acc = 0.0;
    // pass 1: event_level_sum
    load_events(0, cur);
    for (int base = 0; base < max_ne; base += 64 * D) {
        load_events(base + 64 * D, nxt);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (base + 64 * d >= max_ne) break;
#pragma unroll
            for (int q = 0; q < R; ++q) terms[q][0][lane] = (double)cur[d][q];         // (past a read's end: 0.0f, the zero term)
            fence();
            if (lane < R) { const double* row = terms[r1][0];
#pragma unroll 16
                for (int t = 0; t < 64; ++t) acc += row[t]; }
            fence();
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int q = 0; q < R; ++q) cur[d][q] = nxt[d][q];
    }
    const double event_level_sum = acc;                 // lane q < R: read q's
    // pass 2: kmer_level_sum, kmer_level_sq_sum (pow(l, 2) == l * l)
    acc = 0.0;
    auto load_ranks = [&](const int base, uint32_t (&dst)[D][R]) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = base + 64 * d + lane;
#pragma unroll
            for (int q = 0; q < R; ++q) dst[d][q] = i < K[q] ? (uint32_t)rk[q][i] : 0xffffffffu;
        }
    };
    {
        uint32_t rc[D][R], rn[D][R];
        load_ranks(0, rc);
        for (int base = 0; base < max_K; base += 64 * D) {
            load_ranks(base + 64 * D, rn);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (base + 64 * d >= max_K) break;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const uint32_t rank = rc[d][q];
                    const double l = rank != 0xffffffffu ? (table ? level[rank] : model[rank].level_mean) : 0.0;
                    terms[q][0][lane] = l; terms[q][1][lane] = l * l;
                }
                fence();
                if (lane < 2 * R) { const double* row = terms[r2][c2];
#pragma unroll 16
                    for (int t = 0; t < 64; ++t) acc += row[t]; }
                fence();
            }
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int q = 0; q < R; ++q) rc[d][q] = rn[d][q];
        }
    }
    // read q's two sums sit in lanes 2 q and 2 q + 1: bring them to lane q
    const double kmer_level_sum = __shfl(acc, 2 * r1, 64), kmer_level_sq_sum = __shfl(acc, 2 * r1 + 1, 64);
    const int my = (blockIdx.x * NP_MOM_W + wave) * R + r1;
    const int my_ne = lane < R && my < n_reads ? (n_events[my] > 0 ? n_events[my] : 0) : 0, my_K = lane < R && my < n_reads ? (int)reads[my].n_kmers : 1;
    const double shift = event_level_sum / (double)(uint32_t)my_ne - kmer_level_sum / (double)(uint32_t)my_K;
    // pass 3: event_level_sq_sum (every lane needs the shift of the read whose terms it forms)
    double shift_q[R];
#pragma unroll
    for (int q = 0; q < R; ++q) shift_q[q] = dbl_readlane(shift, q);
    acc = 0.0;
    load_events(0, cur);
    for (int base = 0; base < max_ne; base += 64 * D) {
        load_events(base + 64 * D, nxt);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (base + 64 * d >= max_ne) break;
            const int i = base + 64 * d + lane;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                double t = 0.0;
                if (i < ne[q]) { const double dlt = (double)cur[d][q] - shift_q[q]; t = dlt * dlt; }
                terms[q][0][lane] = t;
            }
            fence();
            if (lane < R) { const double* row = terms[r1][0];
#pragma unroll 16
                for (int t = 0; t < 64; ++t) acc += row[t]; }
            fence();
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int q = 0; q < R; ++q) cur[d][q] = nxt[d][q];
    }
    const double event_level_sq_sum = acc;
    if (lane < R && my < n_reads) {
        np_read_dev* rd = reads + my;
        const double scale = (event_level_sq_sum / (double)(uint32_t)my_ne) / (kmer_level_sq_sum / (double)(uint32_t)my_K);
        rd->n_events = (uint32_t)my_ne;
        rd->scale = scale; rd->shift = shift; rd->var = 1.0; rd->log_var = 0.0;      // set4(shift, scale, 0, 1): log(1) == 0
        // raw_loader.cpp:99-108, with glibc's log / exp restated (np_log.h)
        const double events_per_kmer = (double)(uint32_t)my_ne / (double)(uint32_t)my_K;
        const double p_stay = 1 - (1 / (events_per_kmer + 1));
        const double epsilon = 1e-10;
        rd->lp_skip = np_log_glibc(epsilon);
        rd->lp_stay = np_log_glibc(p_stay);
        rd->lp_step = np_log_glibc(1.0 - np_exp_glibc(rd->lp_skip) - np_exp_glibc(rd->lp_stay));
        rd->lp_trim = np_log_glibc(0.01);
        if (reads_b) reads_b[my].n_events = (uint32_t)my_ne;
    }
}

} // namespace

hipError_t np_launch_adc_to_pa(int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples, const float* offset,
                               const float* raw_unit, float* raw_pa, int32_t* status /* not NULL: also the detector's exactness verdict per read */, hipStream_t s)
{
    if (n_reads <= 0 || max_samples <= 0) return hipSuccess;
    if (status) {
        hipLaunchKernelGGL(np_adc_to_pa_check_kernel, dim3(n_reads), dim3(256), 0, s, n_reads, adc, raw_off, offset, raw_unit, raw_pa, status, INT64_MAX);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(np_adc_to_pa_kernel, dim3(n_reads, (unsigned)((max_samples + 1023) / 1024)), dim3(256), 0, s, n_reads, adc, raw_off,
                       offset, raw_unit, raw_pa);
    return hipGetLastError();
}

hipError_t np_launch_detect_events(int n_reads, const float* raw, const int64_t* raw_off, int64_t max_samples, const np_detector_param& p,
                                   float2* tstat, int32_t* status, const int64_t* event_off, int64_t max_events, uint32_t* event_start,
                                   float* event_length, float* event_mean, float* event_stdv, int32_t* n_events, int warmup, bool checked, hipStream_t s, int ratio_exact)
{
    if (n_reads <= 0) return hipSuccess;
    if (!checked) hipLaunchKernelGGL(np_ed_check_kernel, dim3(n_reads), dim3(256), 0, s, n_reads, raw, raw_off, status);
    const unsigned tiles = (unsigned)((max_samples + NP_ED_TILE - 1) / NP_ED_TILE);
    // reads of NP_ED_PAR_MIN samples and more compute their t-statistics inside the peak walk when the windows are the DNA
    // defaults (3 and 6 samples); other window lengths (RNA: 7 and 14) and short reads go through the t-statistic array
    const bool fused = NP_ED_FUSED && ((p.window_length1 == 3 && p.window_length2 == 6) || (p.window_length1 == 6 && p.window_length2 == 3));
    // (fused: only reads shorter than NP_ED_PAR_MIN still need the array -- a grid of eight tiles per read instead of one per 256
    //  samples of the longest read)
    const unsigned t_tiles = fused ? std::min<unsigned>(tiles, (NP_ED_PAR_MIN + NP_ED_TILE - 1) / NP_ED_TILE) : tiles;
    if (t_tiles > 0)
        hipLaunchKernelGGL(np_ed_tstat_kernel, dim3(n_reads, t_tiles), dim3(NP_ED_TILE), 0, s, n_reads, raw, raw_off, status,
                           (int)p.window_length1, (int)p.window_length2, tstat, fused ? (int64_t)NP_ED_PAR_MIN : INT64_MAX);
    // reads on the serial path (rare): their t-statistics, whatever their length, into the same array
    hipLaunchKernelGGL(np_ed_serial_tstat_kernel, dim3(n_reads), dim3(64), 0, s, n_reads, raw, raw_off, status, (int)p.window_length1,
                       (int)p.window_length2, tstat);
    hipLaunchKernelGGL(np_ed_peaks_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, s, n_reads, raw_off, tstat, status, p, event_off,
                       event_start, n_events);
    if (max_samples >= NP_ED_PAR_MIN) {
        if (fused)
            hipLaunchKernelGGL((np_ed_peaks_par_kernel<true, false>), dim3(n_reads), dim3(64), 0, s, n_reads, raw_off, tstat, raw, (const int16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, status, p, event_off, event_start,
                               (uint32_t*)event_length, (uint32_t*)event_mean, (uint32_t*)event_stdv, n_events, warmup < 0 ? NP_ED_WARMUP : warmup, ratio_exact);
        else
            hipLaunchKernelGGL((np_ed_peaks_par_kernel<false, false>), dim3(n_reads), dim3(64), 0, s, n_reads, raw_off, tstat, raw, (const int16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, status, p, event_off, event_start,
                               (uint32_t*)event_length, (uint32_t*)event_mean, (uint32_t*)event_stdv, n_events, warmup < 0 ? NP_ED_WARMUP : warmup, ratio_exact);
    }
    (void)max_events;
    hipLaunchKernelGGL(np_ed_events_kernel<false>, dim3(n_reads), dim3(256), 0, s, n_reads, raw, (const int16_t*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, raw_off, event_off, event_start, n_events, status, event_length, event_mean, event_stdv);
    hipLaunchKernelGGL(np_ed_serial_events_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, s, n_reads, raw, raw_off, event_off, event_start,
                       n_events, status, event_length, event_mean, event_stdv);
    return hipGetLastError();
}

// np_detect_events_adc_dev: int16 counts in, events out.  With the DNA windows the long reads never exist as pA values: the conversion's pass only
// takes the exactness verdict (2 bytes per sample read, nothing written), the walk and the event sums convert the counts they load.  raw_pa is
// written for what still goes through the un-fused kernels: reads shorter than NP_ED_PAR_MIN samples and reads on the serial path.  Other window
// lengths (RNA): the two-call form.
hipError_t np_launch_detect_events_adc(int n_reads, const int16_t* adc, const int64_t* raw_off, int64_t max_samples, const float* offset, const float* raw_unit,
                                       float* raw_pa, const np_detector_param& p, float2* tstat, int32_t* status, const int64_t* event_off, int64_t max_events,
                                       uint32_t* event_start, float* event_length, float* event_mean, float* event_stdv, int32_t* n_events, int warmup, hipStream_t s, int ratio_exact)
{
    if (n_reads <= 0) return hipSuccess;
    const bool fused = NP_ED_FUSED && ((p.window_length1 == 3 && p.window_length2 == 6) || (p.window_length1 == 6 && p.window_length2 == 3));
    if (!fused || max_samples <= 0) {
        const hipError_t e = np_launch_adc_to_pa(n_reads, adc, raw_off, max_samples, offset, raw_unit, raw_pa, status, s);
        if (e != hipSuccess) return e;
        return np_launch_detect_events(n_reads, raw_pa, raw_off, max_samples, p, tstat, status, event_off, max_events, event_start, event_length, event_mean,
                                       event_stdv, n_events, warmup, max_samples > 0, s, ratio_exact);
    }
    hipLaunchKernelGGL(np_adc_to_pa_check_kernel, dim3(n_reads), dim3(256), 0, s, n_reads, adc, raw_off, offset, raw_unit, raw_pa, status, (int64_t)NP_ED_PAR_MIN);
    hipLaunchKernelGGL(np_adc_to_pa_serial_kernel, dim3(n_reads), dim3(256), 0, s, n_reads, adc, raw_off, offset, raw_unit, raw_pa, status, (int64_t)NP_ED_PAR_MIN);
    const unsigned tiles = (unsigned)((max_samples + NP_ED_TILE - 1) / NP_ED_TILE);
    const unsigned t_tiles = std::min<unsigned>(tiles, (NP_ED_PAR_MIN + NP_ED_TILE - 1) / NP_ED_TILE);
    if (t_tiles > 0)
        hipLaunchKernelGGL(np_ed_tstat_kernel, dim3(n_reads, t_tiles), dim3(NP_ED_TILE), 0, s, n_reads, raw_pa, raw_off, status,
                           (int)p.window_length1, (int)p.window_length2, tstat, (int64_t)NP_ED_PAR_MIN);
    hipLaunchKernelGGL(np_ed_serial_tstat_kernel, dim3(n_reads), dim3(64), 0, s, n_reads, raw_pa, raw_off, status, (int)p.window_length1,
                       (int)p.window_length2, tstat);
    hipLaunchKernelGGL(np_ed_peaks_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, s, n_reads, raw_off, tstat, status, p, event_off, event_start, n_events);
    if (max_samples >= NP_ED_PAR_MIN)
        hipLaunchKernelGGL((np_ed_peaks_par_kernel<true, true>), dim3(n_reads), dim3(64), 0, s, n_reads, raw_off, tstat, (const float*)nullptr, adc, offset, raw_unit,
                           status, p, event_off, event_start, (uint32_t*)event_length, (uint32_t*)event_mean, (uint32_t*)event_stdv, n_events,
                           warmup < 0 ? NP_ED_WARMUP : warmup, ratio_exact);
    (void)max_events;
    hipLaunchKernelGGL(np_ed_events_kernel<true>, dim3(n_reads), dim3(256), 0, s, n_reads, (const float*)nullptr, adc, offset, raw_unit, raw_off, event_off,
                       event_start, n_events, status, event_length, event_mean, event_stdv);
    hipLaunchKernelGGL(np_ed_serial_events_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, s, n_reads, raw_pa, raw_off, event_off, event_start,
                       n_events, status, event_length, event_mean, event_stdv);
    return hipGetLastError();
}

// Direct-RNA reads are sequenced 3' -> 5': load_from_raw detects the events, takes the MoM scalings, and then REVERSES the event list
// so that it runs along the basecalled sequence (src/nanopolish_squiggle_read.cpp:260-263) before the event aligner sees it.  One
// workgroup per read swaps event i with event n - 1 - i in up to four arrays (start, length, mean, stdv: whatever the caller keeps).
__global__ void __launch_bounds__(256) np_reverse_events_kernel(const int64_t* __restrict__ event_off, const int32_t* __restrict__ n_events,
                                                                uint32_t* a0, float* a1, float* a2, float* a3)
{
    const int r = blockIdx.x;
    const int n = n_events[r];
    if (n <= 1) return;
    const int64_t o = event_off[r];
    for (int i = threadIdx.x; i < n / 2; i += 256) {
        const int64_t p = o + i, q = o + n - 1 - i;
        if (a0) { const uint32_t t = a0[p]; a0[p] = a0[q]; a0[q] = t; }
        if (a1) { const float t = a1[p]; a1[p] = a1[q]; a1[q] = t; }
        if (a2) { const float t = a2[p]; a2[p] = a2[q]; a2[q] = t; }
        if (a3) { const float t = a3[p]; a3[p] = a3[q]; a3[q] = t; }
    }
}
hipError_t np_launch_reverse_events(int n_reads, const int64_t* event_off, const int32_t* n_events, uint32_t* start, float* length, float* mean,
                                    float* stdv, hipStream_t s)
{
    if (n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(np_reverse_events_kernel, dim3(n_reads), dim3(256), 0, s, event_off, n_events, start, length, mean, stdv);
    return hipGetLastError();
}

hipError_t np_launch_mom_fill(int n_reads, np_read_dev* reads, np_read_dev* reads_b, const float* event_mean, const int32_t* n_events,
                              const uint16_t* ranks, const np_state_dev* model, int n_states, hipStream_t s)
{
    if (n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(np_mom_fill_kernel, dim3((n_reads + NP_MOM_R * NP_MOM_W - 1) / (NP_MOM_R * NP_MOM_W)), dim3(64 * NP_MOM_W), 0, s, n_reads, reads, reads_b, event_mean, n_events, ranks,
                       model, n_states);
    return hipGetLastError();
}

// ---- device self-test of the two-operation divisions by a window length (div_small_f32 / div_small_f64) ---------------------------------
// fp32: EVERY float x with 2^-100 <= |x| < inf, and 0, against the IEEE divide; fp64: n_f64 doubles with random significands and exponents in
// [2^-300, 2^300) (and what a window sum looks like: integers times a power of two).  w == 3 uses the walk's own constants (NP_ED_W3_*).
static __global__ void __launch_bounds__(256) np_selftest_div_small_kernel(int w, double chd, double cld, float chf, float clf, uint64_t n_f64,
                                                                            unsigned long long* out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nth = (uint64_t)gridDim.x * 256;
    const float wf = (float)w; const double wd = (double)w;
    if (w == 3) { chd = NP_ED_W3_CH; cld = NP_ED_W3_CL; chf = NP_ED_W3_CHF; clf = NP_ED_W3_CLF; }
    if (w == 6) { chd = NP_ED_W3_CH * 0.5; cld = NP_ED_W3_CL * 0.5; chf = NP_ED_W3_CHF * 0.5f; clf = NP_ED_W3_CLF * 0.5f; }
    unsigned long long bad32 = 0, bad64 = 0, seen32 = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += nth) {
        const uint32_t bits = (uint32_t)b, mag = bits & 0x7fffffffu;
        if (mag >= 0x7f800000u || (mag != 0u && mag < 0x0d800000u)) continue;            // inf / nan; non-zero below 2^-100
        const float x = __builtin_bit_cast(float, bits);
        const float want = x / wf, got = div_small_f32(x, chf, clf);
        // (x = -0: the quotient is -0 and this form returns the zero whose sign the products' signs make, +0 for the negative cl of 1/3 --
        //  the one value whose bits may differ; the t-statistic squares a mean, takes |delta| and adds the quotients to other terms, and a
        //  sum of fp32 squares is never -0: no zero's sign reaches it)
        bad32 += __builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, got) && !(want == 0.0f && got == 0.0f);
        ++seen32;
    }
    uint64_t sd = 0x9E3779B97F4A7C15ull * (tid + 1) + (uint64_t)w;
    for (uint64_t k = tid; k < n_f64; k += nth) {
        sd += 0x9E3779B97F4A7C15ull;
        uint64_t z = sd; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        uint64_t mant = z & 0xfffffffffffffull;
        if ((z >> 60) < 4) mant &= ~((1ull << (z >> 54 & 31)) - 1ull);                       // a quarter: trailing zeros, like sums of a few floats
        const uint64_t e = 1023 - 300 + ((z >> 52) & 0xff) * 600 / 256;
        const double x = __builtin_bit_cast(double, ((z >> 63) << 63) | (e << 52) | mant);
        const double want = x / wd, got = div_small_f64(x, chd, cld);
        bad64 += __builtin_bit_cast(uint64_t, want) != __builtin_bit_cast(uint64_t, got);
    }
    if (bad32) atomicAdd(out + 0, bad32);
    if (bad64) atomicAdd(out + 1, bad64);
    atomicAdd(out + 2, seen32);
}
hipError_t np_launch_selftest_div_small(int w, double chd, double cld, float chf, float clf, uint64_t n_f64, unsigned long long* d_out, hipStream_t s)
{
    hipLaunchKernelGGL(np_selftest_div_small_kernel, dim3(8192), dim3(256), 0, s, w, chd, cld, chf, clf, n_f64, d_out);
    return hipGetLastError();
}

// ---- device self-test of the filtered t-statistic ratio (ed_ratio_filtered against ed_ratio_exact) ------------------------------------------
// Pseudo-random (|dm|, cvw) pairs over the floats' range as the walk meets them (cvw from FLT_MIN / 6 up, dm from denormals up, zeros now and
// then), plus pairs steered towards rounding boundaries: dm = RN32(m sqrt(cvw)) for a 25-bit midpoint m, so that the quotient lands within
// 2^-24 of a boundary.  out[0]: values the filter TRUSTED whose float differs from the exact sequence's (must be 0); out[1]: values it sent to
// the exact sequence; out[2]: the largest distance (in units of 2^-53) from a boundary pattern at which the UNFILTERED value disagreed -- the
// band is 16384, so this is the margin the accuracy assumption leaves.
static __global__ void __launch_bounds__(256) np_selftest_ratio_kernel(uint64_t n_per_thread, uint64_t seed, unsigned long long* out)
{
    uint64_t sd = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * 256 + threadIdx.x + 1);
    unsigned long long bad = 0, nnear = 0, far = 0;
    for (uint64_t i = 0; i < n_per_thread; ++i) {
        sd += 0x9E3779B97F4A7C15ull;
        uint64_t z = sd; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        // cvw: exponent in [2^-128, 2^40), random significand (the lowest values are denormal floats: FLT_MIN / 6)
        const uint32_t ce = (uint32_t)((z >> 23) % 168u);
        float c = __builtin_bit_cast(float, (ce << 23) | ((uint32_t)z & 0x7fffffu));
        if (c < 1.9e-39f) c = 1.9e-39f;
        float dm;
        if ((z >> 60) < 6) {                 // steered: the quotient within 2^-24 of a float midpoint
            const uint32_t me = 100u + (uint32_t)((z >> 40) % 60u);
            const double m = (double)__builtin_bit_cast(float, (me << 23) | ((uint32_t)(z >> 32) & 0x7fffffu)) * (1.0 + 5.9604644775390625e-08);      // x (1 + 2^-24): a midpoint
            dm = (float)(m * sqrt((double)c));
        } else {
            const uint32_t de = (uint32_t)((z >> 48) % 200u);
            dm = __builtin_bit_cast(float, (de << 23) | ((uint32_t)(z >> 25) & 0x7fffffu));
            if ((z >> 56 & 0xff) == 0) dm = 0.0f;
        }
        const double q = ed_ratio_approx(dm, c);
        const bool near = ed_ratio_near(q);
        const float got = (float)q, want = ed_ratio_exact(dm, c);
        const bool differ = __builtin_bit_cast(uint32_t, got) != __builtin_bit_cast(uint32_t, want);
        bad += differ && !near;
        nnear += near;
        if (differ) {
            const uint32_t L = (uint32_t)__builtin_bit_cast(uint64_t, q) & 0x1fffffffu;
            const uint32_t dist = L > 0x10000000u ? L - 0x10000000u : 0x10000000u - L;
            const bool normal = ((uint32_t)(__builtin_bit_cast(uint64_t, q) >> 32) >= 0x38100000u);
            if (normal) far = far > dist ? far : dist;
        }
    }
    if (bad) atomicAdd(out + 0, bad);
    atomicAdd(out + 1, nnear);
    atomicMax(out + 2, far);
}
hipError_t np_launch_selftest_ratio(uint64_t n_samples, uint64_t seed, unsigned long long* d_out, hipStream_t s)
{
    const unsigned blocks = 8192;
    const uint64_t per_thread = (n_samples + (uint64_t)blocks * 256 - 1) / ((uint64_t)blocks * 256);
    hipLaunchKernelGGL(np_selftest_ratio_kernel, dim3(blocks), dim3(256), 0, s, per_thread, seed, d_out);
    return hipGetLastError();
}
