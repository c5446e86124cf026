// Probe: what does a range-checked raw buffer load of 8 bytes return when only ONE of its two dwords is in range?
// (decides whether kernel A may fetch the event means of a lane's two adjacent slots with one buffer_load_dwordx2)
//   hipcc --offload-arch=gfx950 -O2 tools/buffer_oob_probe.hip -o tools/buffer_oob_probe && tools/buffer_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const float* base, uint32_t bytes, const int* offs, float* out, int n)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    int off = offs[i];
    asm("" : "+v"(off));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    // (this compiler lowers __builtin_amdgcn_raw_buffer_load_b64 to a single-dword load + copy, hence the explicit instruction)
    const uint64_t b = (uint64_t)base;
    u4 d; d.x = __builtin_amdgcn_readfirstlane((uint32_t)b); d.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) & 0xffffu;
    d.z = __builtin_amdgcn_readfirstlane(bytes); d.w = 0x00020000u;
    (void)r;
    uint64_t v;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(d) : "memory");
    out[2 * i] = __builtin_bit_cast(float, (uint32_t)v); out[2 * i + 1] = __builtin_bit_cast(float, (uint32_t)(v >> 32));
}
int main()
{
    const int N = 16;
    float h[N + 8]; for (int i = 0; i < N + 8; ++i) h[i] = 100.0f + i;
    float* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    // the buffer is elements 4..4+N of d (so that memory on both sides exists): base = d + 4, N*4 bytes
    int hoffs[6] = {-8, -4, 0, 4 * (N - 2), 4 * (N - 1), 4 * N};
    int* doffs; hipMalloc(&doffs, sizeof(hoffs)); hipMemcpy(doffs, hoffs, sizeof(hoffs), hipMemcpyHostToDevice);
    float* dout; hipMalloc(&dout, 12 * sizeof(float)); float hout[12];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d + 4, (uint32_t)(N * 4), doffs, dout, 6);
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    for (int i = 0; i < 6; ++i) printf("offset %4d -> (%g, %g)   in-range values would be (%g, %g)\n", hoffs[i], hout[2 * i], hout[2 * i + 1],
                                       h[4 + hoffs[i] / 4], h[4 + hoffs[i] / 4 + 1]);
    return 0;
}
