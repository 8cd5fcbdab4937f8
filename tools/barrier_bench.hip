// Micro-benchmark (diagnostic, not part of the library): latency of a device-wide barrier between
// co-resident workgroups on gfx950, flat counter vs two-level counters.  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Bar {
    unsigned *sub;    // [16 * 16] sub-counters, one per 64-B line
    unsigned *top;    // top counter
    unsigned *flag;   // epoch flag
    unsigned *abort_; // set on timeout
};

template <bool FENCED>
__device__ __forceinline__ bool wait_flag(const Bar &b, unsigned epoch)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(b.flag, FENCED ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) {  // 20 ms
            __hip_atomic_store(b.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// variant 4's wait: the polls PIPELINED -- three loads of the flag in flight, one issued every ~100 ns, their values
// looked at in return order (a wave's loads return in issue order).  A lone poll samples the flag every latency +
// sleep; three in flight sample it three times as often, so the flip is seen about half a load latency sooner.
// One asm statement: the registers of loads still in flight belong to it until it has drained them.
__device__ __forceinline__ bool wait_flag_pipelined(const Bar &b, unsigned epoch)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        unsigned hit, a, bb, c, left = 64;  // at most 64 x 3 polls per statement, then the time limit is looked at
        asm volatile(
            "global_load_dword %[a], %[p], off sc1\n\t"
            "s_sleep 1\n\t"
            "global_load_dword %[b], %[p], off sc1\n\t"
            "s_sleep 1\n\t"
            "global_load_dword %[c], %[p], off sc1\n\t"
            "v_mov_b32 %[hit], 0\n"
            "1:\n\t"
            "s_waitcnt vmcnt(2)\n\t"
            "v_cmp_ge_u32 vcc, %[a], %[e]\n\t"
            "s_cbranch_vccnz 2f\n\t"
            "global_load_dword %[a], %[p], off sc1\n\t"
            "s_waitcnt vmcnt(2)\n\t"
            "v_cmp_ge_u32 vcc, %[b], %[e]\n\t"
            "s_cbranch_vccnz 2f\n\t"
            "global_load_dword %[b], %[p], off sc1\n\t"
            "s_waitcnt vmcnt(2)\n\t"
            "v_cmp_ge_u32 vcc, %[c], %[e]\n\t"
            "s_cbranch_vccnz 2f\n\t"
            "global_load_dword %[c], %[p], off sc1\n\t"
            "v_subrev_u32 %[n], 1, %[n]\n\t"
            "v_cmp_ne_u32 vcc, 0, %[n]\n\t"
            "s_cbranch_vccnz 1b\n\t"
            "s_branch 3f\n"
            "2:\n\t"
            "v_mov_b32 %[hit], 1\n"
            "3:\n\t"
            "s_waitcnt vmcnt(0)"
            : [a] "=&v"(a), [b] "=&v"(bb), [c] "=&v"(c), [hit] "=&v"(hit), [n] "+v"(left)
            : [p] "v"(b.flag), [e] "v"(epoch)
            : "vcc", "memory");
        if (hit) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) {  // 20 ms
            __hip_atomic_store(b.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}

// variant 3: 16 counters on 16 lines, fire-and-forget add, every workgroup polls all of them with one
// 16-lane load and sums (no atomic whose return value anybody waits for)
__device__ __forceinline__ bool poll_barrier(const Bar &b, unsigned G, unsigned epoch)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0)
            (void)__hip_atomic_fetch_add(b.sub + (blockIdx.x % 16) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            unsigned v = threadIdx.x < 16 ? __hip_atomic_load(b.sub + threadIdx.x * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
            v = __shfl(v, 0);
            if (v >= epoch * G) break;
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) {
                if (threadIdx.x == 0) __hip_atomic_store(b.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

template <int LEVELS, bool FENCED>
__device__ __forceinline__ bool grid_barrier(const Bar &b, unsigned G, unsigned epoch)
{
    if (!FENCED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt only: sc1 stores are out
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (LEVELS == 1) {
            const unsigned prev = __hip_atomic_fetch_add(b.top, 1u, FENCED ? __ATOMIC_ACQ_REL : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == epoch * G - 1) __hip_atomic_store(b.flag, epoch, FENCED ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ok = wait_flag<FENCED>(b, epoch);
        } else {
            const unsigned s = blockIdx.x % 16;
            const unsigned members = G / 16 + (s < G % 16 ? 1 : 0);
            const unsigned prev = __hip_atomic_fetch_add(b.sub + s * 16, 1u, FENCED ? __ATOMIC_ACQ_REL : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool last = false;
            if (prev == epoch * members - 1) {
                const unsigned p2 = __hip_atomic_fetch_add(b.top, 1u, FENCED ? __ATOMIC_ACQ_REL : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = p2 == epoch * 16 - 1;
            }
            if (last) __hip_atomic_store(b.flag, epoch, FENCED ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ok = LEVELS == 4 ? wait_flag_pipelined(b, epoch) : wait_flag<FENCED>(b, epoch);
        }
    }
    __syncthreads();
    return ok;
}

template <int LEVELS, bool FENCED>
__global__ void bench(Bar b, unsigned iters, unsigned *sink, unsigned char *scatter, unsigned n_scatter)
{
    unsigned acc = 0;
    for (unsigned e = 1; e <= iters; e++) {
        // a little cross-workgroup traffic so the fences have something to do
        if (scatter) {
            const unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + e * 40503u;
            if (FENCED) scatter[i % n_scatter] = (unsigned char)e;
            else __hip_atomic_store(scatter + i % n_scatter, (unsigned char)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (LEVELS == 3 ? !poll_barrier(b, gridDim.x, e) : !grid_barrier<LEVELS, FENCED>(b, gridDim.x, e)) break;
        if (scatter) {
            const unsigned j = (threadIdx.x * 7919u + blockIdx.x + e) % n_scatter;
            acc += FENCED ? scatter[j] : __hip_atomic_load(scatter + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main()
{
    Bar b;
    unsigned *mem;
    hipMalloc(&mem, 4096 * 4);
    b.sub = mem; b.top = mem + 512; b.flag = mem + 768; b.abort_ = mem + 1024;
    unsigned *sink; hipMalloc(&sink, 64);
    unsigned char *scatter; const unsigned n_scatter = 4u << 20; hipMalloc(&scatter, n_scatter);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned iters = 2000;
    for (int fenced = 0; fenced < 1; fenced++)
    for (int with_traffic = 0; with_traffic < 2; with_traffic++)
    for (int levels = 1; levels <= 4; levels++)
    for (unsigned threads : {256u, 1024u})
    for (unsigned G : {256u, 512u, 1024u}) {
        if ((size_t)G * threads > 256u * 2048u) continue;  // must be co-resident
        hipMemset(mem, 0, 4096 * 4);
        void *args[] = {&b, (void *)&iters, &sink, with_traffic ? (void *)&scatter : (void *)nullptr, (void *)&n_scatter};
        unsigned char *null_scatter = nullptr;
        if (!with_traffic) args[3] = &null_scatter;
        hipEventRecord(e0);
        hipError_t rc = hipLaunchCooperativeKernel(fenced ? (levels == 1 ? (const void *)bench<1, true> : levels == 2 ? (const void *)bench<2, true> : (const void *)bench<3, true>)
                                                          : (levels == 1 ? (const void *)bench<1, false> : levels == 2 ? (const void *)bench<2, false> : levels == 3 ? (const void *)bench<3, false> : (const void *)bench<4, false>),
                                                   dim3(G), dim3(threads), args, 0, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned ab = 0; hipMemcpy(&ab, b.abort_, 4, hipMemcpyDeviceToHost);
        printf("fenced %d traffic %d levels %d threads %4u G %4u: rc %d abort %u  %.3f us per barrier\n", fenced, with_traffic, levels,
               threads, G, (int)rc, ab, ms * 1000.0f / iters);
        fflush(stdout);
        if (ab) return 1;
    }
    return 0;
}
