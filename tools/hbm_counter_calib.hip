// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths kernel A uses
// (/opt/skills/guides/MI355X_MICROARCH.md, "HBM": only wide 16 B/lane reads are calibrated there -- FETCH_SIZE reports 1/2 of them --
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel streams a known number of bytes (1 GiB, far beyond the 256 MiB Infinity Cache) exactly once:
//   calib_read4   coalesced 4 B/lane loads   (kernel A: event means, trace read-back)
//   calib_read16  coalesced 16 B/lane loads  (the guide's calibrated case; kernel A: parameter records)
//   calib_write4  coalesced 4 B/lane stores  (kernel A: packed trace)
//   calib_write8  coalesced 8 B/lane stores  (kernel A: AlignedPair lists)
//   hipcc --offload-arch=gfx950 -O2 tools/hbm_counter_calib.hip -o tools/hbm_counter_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- tools/hbm_counter_calib     (and a second pass with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void calib_read4(const float* __restrict__ p, size_t n, float* sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456f) *sink = acc;
}
__global__ void calib_read16(const float4* __restrict__ p, size_t n, float* sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *sink = acc;
}
__global__ void calib_write4(uint32_t* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void calib_write8(uint2* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint2((uint32_t)i, 7u);
}
int main()
{
    const size_t BYTES = 1ull << 30;
    void* buf; float* sink;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, BYTES);
    hipDeviceSynchronize();
    const dim3 g(256 * 16), b(256);
    hipLaunchKernelGGL(calib_read4, g, b, 0, 0, (const float*)buf, BYTES / 4, sink);
    hipLaunchKernelGGL(calib_read16, g, b, 0, 0, (const float4*)buf, BYTES / 16, sink);
    hipLaunchKernelGGL(calib_write4, g, b, 0, 0, (uint32_t*)buf, BYTES / 4);
    hipLaunchKernelGGL(calib_write8, g, b, 0, 0, (uint2*)buf, BYTES / 8);
    hipDeviceSynchronize();
    printf("each kernel moved %zu bytes once\n", BYTES);
    return 0;
}
