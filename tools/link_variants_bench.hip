// Micro-benchmark (diagnostic, not part of the library): one link of the keyframe chain with the chain kernels' own
// traffic shape, in the forms VERDICT r02 asked to be measured against each other.
//   link = barrier opens | every workgroup loads the mask state of its 16-pixel pieces of keyframe e (1200 pixels) |
//          marks runs of 16 pixels in the masks of the four keyframes after it (about a quarter of its pieces) |
//          waits for the acknowledgement | arrives at the device-wide barrier
//   MASK 0: one byte per pixel, sc1 byte stores / sc1 byte loads             (what the chain kernels did in round 2)
//   MASK 1: one bit per pixel, marks combined per 32-bit word over rows of 16 lanes (DPP segmented OR), one
//           agent-scope atomicOr per word without return value, sc1 word loads
//   MASK 2: the same with returning atomics (the wave waits for the old values: the OR has executed for certain)
//   BAR 0:  two-level arrival counters (16 + 1), the last adder raises 8 flag copies, one lane polls  (round 2)
//   BAR 1:  no atomics: every workgroup stores its epoch into its word of 8 replicas of a 256-word array, one wave
//           polls its replica (1 KiB) until every word has the epoch
//   BAR 2:  16 first-level counters, their last adders store into 8 replicas of a 16-word line, 16 lanes poll one line
// Every mask value a workgroup loads is checked against the closed form of the marks (a stale byte or word counts as a
// mismatch); every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr unsigned kNpx = 640 * 480, kPieces = kNpx / 16, kG = 256, kRounds = (kPieces + kG - 1) / kG;  // 75 pieces per workgroup
constexpr unsigned kLater = 4;
constexpr unsigned kImgWords = (kNpx + 31) / 32;

struct Bar {
    unsigned *sub;     // [16] counters, 64 B apart
    unsigned *top;
    unsigned *flag;    // [8] copies, 4 KiB apart
    unsigned *all;     // [8][256] words (BAR 1), replicas 4 KiB apart; [8][16] for BAR 2 in the same space
    unsigned *abort_;
};

__host__ __device__ inline unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// does piece `pc` of keyframe e survive (all 16 pixels accepted)?
__host__ __device__ inline bool piece_marks(unsigned e, unsigned pc) { return hash32(e * 19349663u + pc * 83492791u) % 100u < 24u; }
__host__ __device__ inline unsigned shift_of(unsigned e, unsigned k) { return k * (37u + e % 5u) + 3u * (e % 7u); }
// closed form: is pixel p of keyframe e marked by one of the four keyframes before it?
__host__ __device__ inline bool expect_marked(unsigned e, unsigned p)
{
    for (unsigned k = 1; k <= kLater; k++) {
        if (e < k + 1) continue;  // keyframes are 1-based
        const unsigned src = (p + kNpx - shift_of(e - k, k) % kNpx) % kNpx;
        if (piece_marks(e - k, src / 16u)) return true;
    }
    return false;
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

__device__ __forceinline__ bool bounded_wait_fail(unsigned long long t0, const Bar &b)
{
    if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) {  // 20 ms
        __hip_atomic_store(b.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return __hip_atomic_load(b.abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// the whole workgroup calls it behind its own store waits; returns false on timeout
template <int BAR>
__device__ __forceinline__ bool grid_barrier(const Bar &b, unsigned e, unsigned *s_ok)
{
    const unsigned G = gridDim.x, wg = blockIdx.x;
    __syncthreads();
    if (BAR == 0) {
        if (threadIdx.x == 0) {
            unsigned ok = 1;
            const unsigned s = wg % 16u, members = G / 16u + (s < G % 16u ? 1u : 0u);
            const unsigned prev = __hip_atomic_fetch_add(b.sub + s * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == e * members - 1u) {
                const unsigned p2 = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (p2 == e * 16u - 1u)
                    for (int x = 0; x < 8; x++) __hip_atomic_store(b.flag + x * 1024, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            while (__hip_atomic_load(b.flag + (wg % 8u) * 1024, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 31u) == 0u && bounded_wait_fail(t0, b)) { ok = 0; break; }
            }
            *s_ok = ok;
        }
    } else if (BAR == 1) {
        if (threadIdx.x < 64) {
            const unsigned lane = threadIdx.x;
            if (lane < 8) __hip_atomic_store(b.all + lane * 1024 + wg, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned *rep = b.all + (wg % 8u) * 1024;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned ok = 1, spins = 0;
            for (;;) {
                bool all = true;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned i = lane + 64u * q;
                    const unsigned v = i < G ? __hip_atomic_load(rep + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : e;
                    all = all && v >= e;
                }
                if (__ballot(all) == ~0ull) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 31u) == 0u && bounded_wait_fail(t0, b)) { ok = 0; break; }
            }
            if (lane == 0) *s_ok = ok;
        }
    } else {
        if (threadIdx.x < 64) {
            const unsigned lane = threadIdx.x;
            const unsigned s = wg % 16u, members = G / 16u + (s < G % 16u ? 1u : 0u);
            unsigned last = 0;
            if (lane == 0) {
                const unsigned prev = __hip_atomic_fetch_add(b.sub + s * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = prev == e * members - 1u ? 1u : 0u;
            }
            last = __builtin_amdgcn_readfirstlane(last);
            if (last && lane < 8) __hip_atomic_store(b.all + lane * 1024 + s, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned *rep = b.all + (wg % 8u) * 1024;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned ok = 1, spins = 0;
            for (;;) {
                const unsigned v = lane < 16 ? __hip_atomic_load(rep + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : e;
                if (__ballot(v >= e) == ~0ull) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 31u) == 0u && bounded_wait_fail(t0, b)) { ok = 0; break; }
            }
            if (lane == 0) *s_ok = ok;
        }
    }
    __syncthreads();
    return *s_ok != 0u;
}

// masks: MASK 0: [iters + 8][kNpx] bytes; MASK 1/2: [iters + 8][kImgWords] words
template <int MASK, int BAR>
__global__ __launch_bounds__(1024) void link_kernel(Bar b, unsigned iters, void *masks, unsigned do_marks, unsigned do_loads,
                                                    unsigned long long *mismatch, unsigned long long *n_atomics)
{
    __shared__ unsigned s_ok;
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6, wg = blockIdx.x, G = gridDim.x;
    unsigned bad = 0, atomics = 0;
    unsigned char *mb = static_cast<unsigned char *>(masks);
    unsigned *mw = static_cast<unsigned *>(masks);
    for (unsigned e = 1; e <= iters; e++) {
        if (e > 1 && !grid_barrier<BAR>(b, e - 1, &s_ok)) break;
        // ---- step A: the mask state of my pixels of keyframe e (two segments per wave, like fuse_chain_kernel)
        unsigned pix[2], mv[2];
        bool in[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const unsigned pr = ((unsigned)u * 16u + w) * 4u + lane / 16u;   // piece round of this lane
            const unsigned piece = G * pr + (wg + 97u * pr) % G;
            in[u] = pr < kRounds && piece < kPieces;
            pix[u] = in[u] ? piece * 16u + lane % 16u : 0u;
            mv[u] = 0;
            if (do_loads) {
                if (MASK == 0)
                    mv[u] = __hip_atomic_load(mb + (size_t)e * kNpx + pix[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    mv[u] = (__hip_atomic_load(mw + (size_t)e * kImgWords + (pix[u] >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (pix[u] & 31u)) & 1u;
            }
        }
        unsigned dep = 0;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (do_loads && in[u]) bad += (mv[u] != 0u) != expect_marked(e, pix[u]);
            unsigned v = mv[u];
            asm volatile("" : "+v"(v));  // the marks depend on the loaded value, as the accept decision does
            dep |= v & 0x100u;           // (always 0)
        }
        // ---- step C: marks in the four keyframes after this one
        if (do_marks) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const bool on = in[u] && piece_marks(e, pix[u] / 16u);
#pragma unroll
                for (unsigned k = 1; k <= kLater; k++) {
                    const unsigned hp = (pix[u] + shift_of(e, k)) % kNpx + dep;
                    if (MASK == 0) {
                        if (on) __hip_atomic_store(mb + (size_t)(e + k) * kNpx + hp, (unsigned char)255, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        // marks of one 32-bit word combined over the row of 16 lanes, the run's last lane issues the OR
                        const unsigned wid = on ? (hp >> 5) + 1u : 0u;
                        unsigned bits = on ? 1u << (hp & 31u) : 0u;
                        { const unsigned pw = dpp<0x111>(wid), pb = dpp<0x111>(bits); bits |= pw == wid ? pb : 0u; }
                        { const unsigned pw = dpp<0x112>(wid), pb = dpp<0x112>(bits); bits |= pw == wid ? pb : 0u; }
                        { const unsigned pw = dpp<0x114>(wid), pb = dpp<0x114>(bits); bits |= pw == wid ? pb : 0u; }
                        { const unsigned pw = dpp<0x118>(wid), pb = dpp<0x118>(bits); bits |= pw == wid ? pb : 0u; }
                        const unsigned nw = dpp<0x101>(wid);  // row_shl:1: the next lane's word (0 past the row's end)
                        if (on && nw != wid) {
                            unsigned *p = mw + (size_t)(e + k) * kImgWords + (wid - 1u);
                            if (MASK == 1) {
                                (void)__hip_atomic_fetch_or(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
                                const unsigned old = __hip_atomic_fetch_or(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                asm volatile("" ::"v"(old));
                            }
                            atomics++;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o); atomics += __shfl_xor(atomics, o); }
    if (lane == 0 && bad) atomicAdd(mismatch, (unsigned long long)bad);
    if (lane == 0 && atomics) atomicAdd(n_atomics, (unsigned long long)atomics);
}

template <int MASK, int BAR>
static int run(const char *name, Bar b, unsigned *barmem, size_t barbytes, void *masks, size_t maskbytes, unsigned iters,
               unsigned do_marks, unsigned do_loads, unsigned long long *d_cnt)
{
    hipMemset(barmem, 0, barbytes);
    hipMemset(masks, 0, maskbytes);
    hipMemset(d_cnt, 0, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long *mm = d_cnt, *na = d_cnt + 1;
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL((link_kernel<MASK, BAR>), dim3(kG), dim3(1024), 0, nullptr, b, iters, masks, do_marks, do_loads, mm, na);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned ab = 0;
    unsigned long long h[2] = {0, 0};
    hipMemcpy(&ab, b.abort_, 4, hipMemcpyDeviceToHost);
    hipMemcpy(h, d_cnt, 16, hipMemcpyDeviceToHost);
    printf("%-44s marks %u loads %u: abort %u mismatches %8llu  atomics/keyframe %7.0f  %.3f us per link\n", name, do_marks, do_loads,
           ab, h[0], (double)h[1] / iters, ms * 1000.0f / iters);
    fflush(stdout);
    return ab ? 1 : 0;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const unsigned iters = 1000;
    unsigned *barmem;
    const size_t barbytes = 64 * 1024 * 4;
    hipMalloc(&barmem, barbytes);
    Bar b;
    b.sub = barmem;
    b.top = barmem + 512;
    b.abort_ = barmem + 768;
    b.flag = barmem + 1024;            // 8 copies, 1024 words apart
    b.all = barmem + 16 * 1024;        // 8 replicas, 1024 words apart
    void *masks;
    const size_t maskbytes = (size_t)(iters + 8) * kNpx;
    hipMalloc(&masks, maskbytes);
    unsigned long long *d_cnt;
    hipMalloc(&d_cnt, 16);
    int rc = 0;
    for (int rep = 0; rep < 2 && !rc; rep++) {
        rc |= run<0, 0>("bytes, two-level counters + flags (r02)", b, barmem, barbytes, masks, maskbytes, iters, 1, 1, d_cnt);
        rc |= run<0, 1>("bytes, all-to-all flag words", b, barmem, barbytes, masks, maskbytes, iters, 1, 1, d_cnt);
        rc |= run<0, 2>("bytes, counters + flag line", b, barmem, barbytes, masks, maskbytes, iters, 1, 1, d_cnt);
        rc |= run<1, 0>("bits (atomicOr, no return), two-level", b, barmem, barbytes, masks, maskbytes, iters, 1, 1, d_cnt);
        rc |= run<1, 1>("bits (atomicOr, no return), all-to-all", b, barmem, barbytes, masks, maskbytes, iters, 1, 1, d_cnt);
        rc |= run<1, 2>("bits (atomicOr, no return), flag line", b, barmem, barbytes, masks, maskbytes, iters, 1, 1, d_cnt);
        rc |= run<2, 0>("bits (atomicOr, returning), two-level", b, barmem, barbytes, masks, maskbytes, iters, 1, 1, d_cnt);
        rc |= run<2, 1>("bits (atomicOr, returning), all-to-all", b, barmem, barbytes, masks, maskbytes, iters, 1, 1, d_cnt);
    }
    // where the link's time goes: without the marks, without the loads, the barrier alone
    for (int dm = 0; dm < 2 && !rc; dm++)
        for (int dl = 0; dl < 2 && !rc; dl++) {
            if (dm && dl) continue;
            rc |= run<0, 0>("bytes, two-level counters + flags (r02)", b, barmem, barbytes, masks, maskbytes, iters, dm, dl, d_cnt);
            rc |= run<0, 1>("bytes, all-to-all flag words", b, barmem, barbytes, masks, maskbytes, iters, dm, dl, d_cnt);
            rc |= run<1, 0>("bits (atomicOr, no return), two-level", b, barmem, barbytes, masks, maskbytes, iters, dm, dl, d_cnt);
            rc |= run<1, 1>("bits (atomicOr, no return), all-to-all", b, barmem, barbytes, masks, maskbytes, iters, dm, dl, d_cnt);
        }
    return rc;
}
