/* A "different libm" for tests/test_gpu_libm.py: log, exp and logf of the C library, with the result moved ONE ULP up for about half
 * of the arguments (chosen by a hash of the argument's bits, so that the function stays a function).  LD_PRELOADed into a child
 * process, it stands for a reference built against another libm version or a build without FMA -- whose constants
 * (src/nanopolish_raw_loader.cpp:99-108, src/nanopolish_squiggle_read.cpp:38-65, src/hmm/nanopolish_profile_hmm_r9.inl:61-72) differ in
 * the last bit from the library's restatement of glibc 2.35.  log(1) stays 0 and exact results (exp(0) = 1) stay exact. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

static int pick64(double x) { uint64_t u; memcpy(&u, &x, 8); u ^= u >> 29; u *= 0x9E3779B97F4A7C15ull; return (int)((u >> 40) & 1u); }
static int pick32(float x) { uint32_t u; memcpy(&u, &x, 4); u ^= u >> 13; u *= 0x9E3779B1u; return (int)((u >> 20) & 1u); }

double log(double x)
{
    static double (*real)(double);
    if (!real) real = (double (*)(double))dlsym(RTLD_NEXT, "log");
    const double r = real(x);
    return (x != 1.0 && isfinite(r) && r != 0.0 && pick64(x)) ? nextafter(r, INFINITY) : r;
}

double exp(double x)
{
    static double (*real)(double);
    if (!real) real = (double (*)(double))dlsym(RTLD_NEXT, "exp");
    const double r = real(x);
    return (x != 0.0 && isfinite(r) && r != 0.0 && pick64(x)) ? nextafter(r, INFINITY) : r;
}

float logf(float x)
{
    static float (*real)(float);
    if (!real) real = (float (*)(float))dlsym(RTLD_NEXT, "logf");
    const float r = real(x);
    return (x != 1.0f && isfinite(r) && r != 0.0f && pick32(x)) ? nextafterf(r, INFINITY) : r;
}
