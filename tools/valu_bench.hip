// Micro-benchmark (diagnostic): issue rate of the FP64 VALU instructions the fusion kernels are
// made of, per SIMD of a gfx950 CU, with 4 waves per SIMD (the occupancy of fuse_chain_kernel).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(1024) void k(double *out, double a0, double b0, int iters, unsigned long long *cyc)
{
    double x0 = a0 + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const double b = b0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#define STEP(x)                                                                  \
    if (OP == 0) x = __builtin_fma(x, b, b);                                      \
    else if (OP == 1) x = x * b;                                                  \
    else if (OP == 2) x = x + b;                                                  \
    else if (OP == 3) x = __builtin_amdgcn_rcp(x);                                \
    else if (OP == 4) x = x / b;                                                  \
    else if (OP == 5) x = __builtin_amdgcn_div_fixup(x, b, x);                    \
    else if (OP == 6) x = (double)(int)x;                                         \
    else if (OP == 7) x = __builtin_trunc(x) + b;
        STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, int instr_per_step)
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, 1.0, 1.000001, 10, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, 1.0, 1.000001, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    // per SIMD: 4 waves x 8 chains x iters steps
    const double steps = 4.0 * 8 * iters;
    printf("%-12s %8.3f ms  %7.1f ns per step per SIMD  (s_memtime ticks/step %.2f)  ~%.1f cycles at 2.4 GHz per step, %d instr\n", name, ms,
           ms * 1e6 / steps, (double)h[0] / steps, ms * 1e6 / steps * 2.4, instr_per_step);
}

int main()
{
    run<0>("v_fma_f64", 1); run<1>("v_mul_f64", 1); run<2>("v_add_f64", 1); run<3>("v_rcp_f64", 1);
    run<4>("fdiv f64", 12); run<5>("div_fixup", 1); run<6>("cvt i32<->f64", 2); run<7>("trunc+add", 2);
    return 0;
}
