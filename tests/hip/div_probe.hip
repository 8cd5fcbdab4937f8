// div_probe.hip -- GPU unit test of div_pair (sfm_amd/csrc/fusion_kernels.h): the shared-reciprocal
// pair of divisions must give the bits of two plain IEEE divisions for every operand triple it does
// not flag.  Operands: random sign/exponent/mantissa with exponents inside and around the accepted
// range, plus every combination of special values.  Prints "mismatches <n> flagged <m> of <total>";
// exit code 1 on any mismatch.  Built by build.sh into build/div_probe, run by tests/test_gpu_div_pair.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../sfm_amd/csrc/fusion_kernels.h"

using namespace sfm;

__device__ unsigned long long splitmix(unsigned long long &s)
{
    s += 0x9E3779B97F4A7C15ull;
    unsigned long long z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ double special_value(unsigned i)
{
    const unsigned long long bits[] = {
        0x0000000000000000ull, 0x8000000000000000ull, 0x7FF0000000000000ull, 0xFFF0000000000000ull,
        0x7FF8000000000000ull, 0xFFF8000000000000ull, 0x0000000000000001ull, 0x800FFFFFFFFFFFFFull,
        0x0010000000000000ull, 0x7FEFFFFFFFFFFFFFull, 0x3FF0000000000000ull, 0xBFF0000000000000ull,
        0x26F0000000000000ull /* 2^-400 */, 0x58F0000000000000ull /* 2^400 */,
        0x26EFFFFFFFFFFFFFull /* just below 2^-400 */, 0x58F0000000000001ull /* just above 2^400 */};
    return __longlong_as_double((long long)bits[i % 16]);
}

__device__ double random_value(unsigned long long &s, int wide)
{
    const unsigned long long r = splitmix(s);
    const unsigned long long mant = r & 0xFFFFFFFFFFFFFull;
    const unsigned long long sign = (r >> 63) << 63;
    const int span = wide ? 900 : 780;                       // exponents in +-450 (some flagged) or +-390
    const int e = 1023 + (int)((r >> 52) % (unsigned)span) - span / 2;
    return __longlong_as_double((long long)(sign | ((unsigned long long)e << 52) | mant));
}

__global__ void probe(unsigned long long seed, unsigned rounds, unsigned long long *out)
{
    unsigned long long s = seed + 0x1234567ull * (blockIdx.x * blockDim.x + threadIdx.x);
    unsigned long long mism = 0, flagged = 0, total = 0;
    for (unsigned it = 0; it < rounds; it++) {
        double n0, n1, d;
        const unsigned kind = it & 7u;
        if (kind == 0) {  // specials in every position
            const unsigned long long r = splitmix(s);
            n0 = special_value((unsigned)r);
            n1 = special_value((unsigned)(r >> 8));
            d = special_value((unsigned)(r >> 16));
        } else if (kind == 1) {  // one special, two ordinary
            const unsigned long long r = splitmix(s);
            n0 = random_value(s, 0);
            n1 = random_value(s, 0);
            d = random_value(s, 0);
            if ((r & 3) == 0) n0 = special_value((unsigned)(r >> 8));
            if ((r & 3) == 1) n1 = special_value((unsigned)(r >> 8));
            if ((r & 3) == 2) d = special_value((unsigned)(r >> 8));
        } else {
            n0 = random_value(s, kind == 2);
            n1 = random_value(s, kind == 2);
            d = random_value(s, kind == 2);
            if (kind == 3) n1 = n0 * 3.0;                    // related numerators
            if (kind == 4) n0 = d * (double)(1 + (splitmix(s) & 1023));  // exact quotients
        }
        double q0, q1;
        bool bad = false;
        div_pair<kDivSharedChecked>(n0, n1, d, q0, q1, bad);
        const double p0 = n0 / d, p1 = n1 / d;
        total++;
        if (bad) {
            flagged++;
        } else if (__double_as_longlong(q0) != __double_as_longlong(p0) ||
                   __double_as_longlong(q1) != __double_as_longlong(p1)) {
            // NaN results: payload and sign must agree too (v_div_fixup produces both)
            mism++;
        }
    }
    atomicAdd(&out[0], mism);
    atomicAdd(&out[1], flagged);
    atomicAdd(&out[2], total);
}

int main(int argc, char **argv)
{
    const unsigned rounds = argc > 1 ? (unsigned)std::atoi(argv[1]) : 4096;
    unsigned long long *d_out = nullptr, h[3] = {0, 0, 0};
    if (hipMalloc(&d_out, sizeof h) != hipSuccess) { std::fprintf(stderr, "no HIP device\n"); return 2; }
    (void)hipMemset(d_out, 0, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, 20240315ull, rounds, d_out);
    if (hipDeviceSynchronize() != hipSuccess) { std::fprintf(stderr, "kernel failed\n"); return 2; }
    (void)hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    std::printf("mismatches %llu flagged %llu of %llu\n", h[0], h[1], h[2]);
    return h[0] ? 1 : 0;
}
