// Micro-benchmark (diagnostic, not part of the library): what one link of the keyframe chain costs when all its
// workgroups sit on ONE XCD and talk through that XCD's L2, against the device-wide form through the fabric.
//   link = every thread stores one byte another workgroup will read | wait for the stores | barrier | load a byte
//          another workgroup stored in this round and check its value
//   scope 0: agent scope (sc1 write-through stores, sc1 loads, agent-scope atomics) -- what the chain kernels use
//   scope 1: L2 scope (sc0 stores and loads, atomics without sc1): only valid when every workgroup shares one L2
// One XCD = workgroups 0, 8, 16, ... of a grid (the others leave at once); the kernel records the XCC_ID of every
// participant so that the placement is measured, not assumed.  Every spin is bounded; mismatches are counted, not assumed away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Bar {
    unsigned *count;   // arrival counter
    unsigned *flag;    // epoch flag
    unsigned *abort_;  // set on timeout
};

__device__ __forceinline__ unsigned xcc_id()
{
    return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xFu;  // HW_REG_XCC_ID[3:0]
}

template <int SCOPE>
__device__ __forceinline__ void store_byte(unsigned char *p, unsigned char v)
{
    if (SCOPE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else asm volatile("global_store_byte %0, %1, off sc0" ::"v"(p), "v"((unsigned)v) : "memory");
}

template <int SCOPE>
__device__ __forceinline__ unsigned load_byte(const unsigned char *p)
{
    if (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v;
    asm volatile("global_load_ubyte %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int SCOPE>
__device__ __forceinline__ unsigned add_count(unsigned *p)
{
    if (SCOPE == 0) return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // executes in this XCD's L2
}

template <int SCOPE>
__device__ __forceinline__ unsigned poll_flag(unsigned *p)
{
    if (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (an sc0 load of a line this CU has read before hits in its L1 and never sees the flag: the poll is a
    // read-modify-write, which always executes in the L2)
    return __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int SCOPE>
__device__ __forceinline__ void set_flag(unsigned *p, unsigned v)
{
    if (SCOPE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ONE_XCD: the grid is 8 x the participants; workgroups are dealt to the XCCs round-robin (workgroup b runs on XCC
// (b + c) mod 8, c differs from launch to launch), so workgroups 0, 8, 16, ... share an XCC: they take part, the
// others leave at once.  The host checks the recorded XCC_IDs.
template <int DATA_SCOPE, int BAR_SCOPE, bool ONE_XCD>
__global__ __launch_bounds__(1024) void link_bench(Bar b, unsigned iters, unsigned char *data, unsigned *xcc_out,
                                                   unsigned *mismatch)
{
    __shared__ unsigned s_ok;
    if (ONE_XCD && (blockIdx.x & 7u) != 0u) return;
    const unsigned G = ONE_XCD ? gridDim.x / 8u : gridDim.x, t = threadIdx.x, wg = ONE_XCD ? blockIdx.x / 8u : blockIdx.x;
    if (t == 0) xcc_out[wg] = xcc_id();
    unsigned bad = 0;
    for (unsigned e = 1; e <= iters; e++) {
        unsigned char *buf = data + (size_t)(e & 1u) * 256 * 1024;  // two buffers: a fast writer's next round leaves this one alone
        store_byte<DATA_SCOPE>(buf + (size_t)wg * 1024 + t, (unsigned char)e);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            unsigned ok = 1;
            const unsigned prev = add_count<BAR_SCOPE>(b.count);
            if (prev == e * G - 1u) {
                set_flag<BAR_SCOPE>(b.flag, e);
            } else {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                while (poll_flag<BAR_SCOPE>(b.flag) < e) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) {  // 20 ms
                        __hip_atomic_store(b.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
            }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) break;
        const unsigned other = (wg + 1u + e) % G;  // cycles through every workgroup: lines come back after G rounds
        const unsigned v = load_byte<DATA_SCOPE>(buf + (size_t)other * 1024 + t);
        bad += v != (e & 0xFFu);
        __syncthreads();
    }
    if (bad) atomicAdd(mismatch, bad);
}

static void run(const char *name, const void *kern, hipStream_t s, unsigned G, unsigned grid, Bar b, unsigned *mem, unsigned char *data,
                unsigned *xcc, unsigned *mism)
{
    const unsigned iters = 2000;
    hipMemsetAsync(mem, 0, 4096 * 4, s);
    hipMemsetAsync(data, 0, 2 * 256 * 1024, s);
    hipMemsetAsync(xcc, 0xFF, 256 * 4, s);
    hipMemsetAsync(mism, 0, 4, s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    void *args[] = {&b, (void *)&iters, &data, &xcc, &mism};
    hipEventRecord(e0, s);
    const hipError_t rc = hipLaunchKernel(kern, dim3(grid), dim3(1024), args, 0, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned ab = 0, mm = 0, hx[256];
    hipMemcpy(&ab, b.abort_, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost);
    hipMemcpy(hx, xcc, G * 4, hipMemcpyDeviceToHost);
    unsigned per[16] = {0};
    for (unsigned i = 0; i < G; i++) per[hx[i] & 15u]++;
    printf("%-46s G %3u rc %d abort %u mismatches %8u  %.3f us per link   workgroups per XCC:", name, G, (int)rc, ab, mm,
           ms * 1000.0f / iters);
    for (int i = 0; i < 8; i++) printf(" %u", per[i]);
    printf("\n");
    fflush(stdout);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned *mem;
    hipMalloc(&mem, 4096 * 4);
    Bar b;
    b.count = mem;
    b.flag = mem + 256;
    b.abort_ = mem + 512;
    unsigned char *data;
    hipMalloc(&data, 2 * 256 * 1024);
    unsigned *xcc, *mism;
    hipMalloc(&xcc, 256 * 4);
    hipMalloc(&mism, 4);
    hipStream_t plain;
    hipStreamCreate(&plain);
    // (a CU-masked stream does not confine workgroups to an XCD: mask bit i is a CU of XCC i mod 8 -- the first 32 bits
    // gave 4 workgroups per XCC -- and creating a stream with every eighth bit set crashed inside the runtime)
    run("agent scope", (const void *)link_bench<0, 0, false>, plain, 256, 256, b, mem, data, xcc, mism);
    run("agent scope", (const void *)link_bench<0, 0, false>, plain, 32, 32, b, mem, data, xcc, mism);
    // negative control: L2-scope data on workgroups spread over the XCDs must show stale bytes (agent-scope barrier)
    run("L2-scope data, agent barrier, 8 XCDs (control)", (const void *)link_bench<1, 0, false>, plain, 32, 32, b, mem, data, xcc, mism);
    // one XCD: workgroups 0, 8, 16, ... of a 256-workgroup grid
    run("agent scope, one XCD", (const void *)link_bench<0, 0, true>, plain, 32, 256, b, mem, data, xcc, mism);
    run("L2-scope data, agent barrier, one XCD", (const void *)link_bench<1, 0, true>, plain, 32, 256, b, mem, data, xcc, mism);
    // (a barrier through the L2 alone -- counter and flag at workgroup scope -- is not measured: an sc0 load of the flag
    // hits in the polling CU's L1, and the compiler folds an idempotent workgroup-scope read-modify-write into such a
    // load; both variants ran into the spin bound.  The agent-scope barrier is what the rows above use.)
    return 0;
}
