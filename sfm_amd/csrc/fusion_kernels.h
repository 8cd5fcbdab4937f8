// fusion_kernels.h -- hand-written gfx950 kernels of the dense-fusion path.
//
// Compiled with -ffp-contract=off: every double operation below is a single IEEE-754
// round-to-nearest op in the order the reference evaluates it (reference builds with
// clang -O3, no -march => SSE2, no FMA; reconstruction/CMakeLists.txt:3-5).  The Eigen
// evaluation orders that matter are spelled out next to each function.
//
// Execution model: one workgroup = ONE wave64.  A wave streams 256-pixel tiles of the
// current keyframe (coalesced ushort4 depth + uchar4 mask loads), ballot/popcount-compacts
// the live pixels (mask==0 && depth>0) into a wave-private LDS queue, and runs the
// expensive per-pixel work only on dense batches of 64 queued pixels, so every lane of a
// batch is busy.  Keyframe/neighbour intrinsics and poses are wave-uniform: they are read
// through the scalar cache into SGPRs (the CDNA form of "stage the pose once per tile"),
// which keeps ~70 VGPRs per neighbour free compared with an LDS/VGPR copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfm {

constexpr int kWave = 64;
constexpr int kPxPerLane = 4;
constexpr int kTile = kWave * kPxPerLane;  // 256 pixels per wave-tile
constexpr int kMaxK = 32;                  // SFM_MAX_NEIGHBORS
constexpr uint64_t kEmpty = ~0ull;
constexpr uint32_t kNoHit = 0xFFFFFFFFu;

// error bits in Counters::err
constexpr uint32_t kErrTableFull = 1u;
constexpr uint32_t kErrKeyRange = 2u;
constexpr uint32_t kErrListFull = 4u;

struct FrameDev {
    double K[4];      // fx fy cx cy
    double Twk[12];   // rows 0..2 of camera->world, row-major [r*4+c]
    double Tkw[12];   // rows 0..2 of world->camera
    const uint16_t *depth;  // W*H, packed rows
    const uint8_t *bgr;     // W*H*3
    uint8_t *mask;          // W*H visibility mask (dense_reconstruction.hpp:35-39)
    int32_t W, H;
    uint32_t nbr_begin, nbr_count;  // filtered neighbours in nbr_idx[]
    uint32_t order, pad;
};

struct DevParams {
    double base_line, T_cov, voxel_size;
    double tau_dist2;  // smallest s with sqrt(s) >= T_dist: "norm < T_dist"  <=>  s < tau_dist2
    double S0, S1, S2, depth_div, close_frac;
};

struct Slot {
    unsigned long long key;  // packed 3x21-bit voxel key, kEmpty when free
    unsigned long long tag;  // min over inserted (order<<32 | pixel); kEmpty when free
};

struct Counters {
    unsigned long long list_count;
    unsigned long long out_count;
    unsigned long long table_used;
    unsigned long long n_gather;
    unsigned long long n_hit;
    unsigned int err;
    unsigned int max_probe;
};

struct FuseArgs {
    Slot *table;
    uint32_t *mark;          // per slot: 1 + last frame order that touched it
    uint64_t table_mask;     // capacity-1
    // accepted-point list (SoA)
    unsigned long long *list_key;
    unsigned long long *list_tag;
    double *list_xyz;        // 3 per point
    uint32_t *list_bgr;      // b | g<<8 | r<<16
    uint64_t list_cap;
    Counters *ctr;
    unsigned long long *frame_before;
    unsigned long long *frame_after;
    DevParams P;
    uint32_t f;
};

// ---- exact arithmetic helpers ---------------------------------------------------------

// trace(J*S*J^T), dense_reconstruction.cpp:292-313.  Eigen: M=J*S temp, P=M*J^T, lazy
// coefficient a0*b0+(a1*b1+a2*b2) where all but one/two terms are exact zeros, then
// trace = P00 + (P11 + P22)  (redux_novec_unroller<0,3>).
__device__ __forceinline__ double cov_trace(const DevParams &P, int u, int v, double d, double f)
{
    const double disp = (f * P.base_line) / (d / P.depth_div);
    const double dd = disp * disp;
    const double J00 = P.base_line / disp;
    const double J02 = ((double)(-u) * P.base_line) / dd;
    const double J12 = ((double)(-v) * P.base_line) / dd;
    const double J22 = ((-f) * P.base_line) / dd;
    const double P00 = (J00 * P.S0) * J00 + (J02 * P.S2) * J02;
    const double P11 = (J00 * P.S1) * J00 + (J12 * P.S2) * J12;
    const double P22 = (J22 * P.S2) * J22;
    return P00 + (P11 + P22);
}

// row r of T*[p;1]: Homogeneous product = leftCols<3>()*p (inner index ascending) + col(3).
__device__ __forceinline__ double row_apply(const double *T, int r, double x, double y, double z)
{
    double acc = T[r * 4 + 0] * x;
    acc = acc + T[r * 4 + 1] * y;
    acc = acc + T[r * 4 + 2] * z;
    return acc + T[r * 4 + 3];
}

// static_cast<size_t>(coord / voxel_size) as x86-64 evaluates it, returned
// as a signed 64-bit index; INT64_MIN stands for the 0x8000000000000000 "indefinite" key.
__device__ __forceinline__ long long voxel_index(double coord, double vs)
{
    const double q = coord / vs;
    if (q != q) return INT64_MIN;
    const double t = trunc(q);
    if (t >= -9223372036854775808.0 && t < 9223372036854775808.0) return (long long)t;
    if (t >= 9223372036854775808.0 && t < 18446744073709551616.0)
        return (long long)(unsigned long long)t;
    return INT64_MIN;
}

// 21-bit two's-complement code per axis; -2^20 is reserved for INT64_MIN.
__device__ __forceinline__ bool axis_code(long long k, unsigned long long &code)
{
    if (k == INT64_MIN) { code = 0x100000ull; return true; }
    if (k < -1048575ll || k > 1048575ll) return false;
    code = (unsigned long long)k & 0x1FFFFFull;
    return true;
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << lane_id()) - 1ull;
}

// Claims/finds the slot of `key` and lowers its tag to `tag` (first-wins == min tag,
// dense_reconstruction.cpp:283-287 composed per frame and globally).  Returns the slot or
// ~0 on a full table.  Stale plain reads are safe: keys never change once set, tags only
// decrease.
__device__ __forceinline__ uint64_t table_insert(const FuseArgs &a, unsigned long long key,
                                                 unsigned long long tag, unsigned &probes)
{
    uint64_t h = mix64(key) & a.table_mask;
    probes = 0;
    for (;;) {
        unsigned long long cur = a.table[h].key;
        if (cur == kEmpty) {
            cur = atomicCAS(&a.table[h].key, kEmpty, key);
            if (cur == kEmpty) {
                atomicAdd(&a.ctr->table_used, 1ull);
                cur = key;
            }
        }
        if (cur == key) {
            if (a.table[h].tag > tag) atomicMin(&a.table[h].tag, tag);
            return h;
        }
        h = (h + 1) & a.table_mask;
        if (++probes > a.table_mask) return ~0ull;
    }
}

// ---- the per-keyframe kernel ------------------------------------------------------------

// process_pixel + geometric_check (dense_reconstruction.cpp:126-234) for one queued pixel
// per lane.  `act` = this lane holds a real pixel.
__device__ __forceinline__ void process_batch(const FrameDev *__restrict__ frames,
                                              const uint32_t *__restrict__ nbr_idx,
                                              const FuseArgs &a, const FrameDev &F, uint32_t pix,
                                              bool act, uint32_t *s_hits,
                                              unsigned &c_gather, unsigned &c_hit,
                                              unsigned &c_acc, unsigned &c_after)
{
    const DevParams &P = a.P;
    const unsigned lane = lane_id();
    const int W = F.W;
    const int u = (int)(pix % (uint32_t)W);
    const int v = (int)(pix / (uint32_t)W);
    bool ok = act;
    double pwx = 0, pwy = 0, pwz = 0;
    if (ok) {
        const double d = (double)F.depth[pix];
        const double tr = cov_trace(P, u, v, d, F.K[0]);             // :136
        if (tr > P.T_cov) ok = false;                                // :138
        const double x = ((double)u - F.K[2]) / F.K[0];              // pinhole.cpp:12
        const double y = ((double)v - F.K[3]) / F.K[1];
        const double pcx = x * d, pcy = y * d, pcz = d;              // :15-19
        pwx = row_apply(F.Twk, 0, pcx, pcy, pcz);                    // :144
        pwy = row_apply(F.Twk, 1, pcx, pcy, pcz);
        pwz = row_apply(F.Twk, 2, pcx, pcy, pcz);
    }
    const uint32_t nK = F.nbr_count;
    if (nK == 0) ok = false;                                         // :180
    double wx = 0, wy = 0, wz = 0, total = 0;
    unsigned n_obs = 0, n_close = 0;
    for (uint32_t k = 0; k < nK; k++) {                              // :187
        const uint32_t nb = nbr_idx[F.nbr_begin + k];
        const FrameDev &L = frames[nb];
        uint32_t hit = kNoHit;
        if (ok) {
            // KeyFrame::point_in_frame, keyframe.cpp:275-294 / pinhole.cpp:32-45
            const double cx_ = row_apply(L.Tkw, 0, pwx, pwy, pwz);
            const double cy_ = row_apply(L.Tkw, 1, pwx, pwy, pwz);
            const double cz_ = row_apply(L.Tkw, 2, pwx, pwy, pwz);
            const double px = (cx_ / cz_) * L.K[0] + L.K[2];
            const double py = (cy_ / cz_) * L.K[1] + L.K[3];
            // NaN projections are "not visible" (UB in the reference), hence the >= 0 form
            const bool vis = (px >= 0) && (px < (double)L.W) && (py >= 0) && (py < (double)L.H) &&
                             !(cz_ < 0);
            if (vis) {
                const int ix = (int)floor(px);
                const int iy = (int)floor(py);
                const uint32_t npix = (uint32_t)iy * (uint32_t)L.W + (uint32_t)ix;
                const double d = (double)L.depth[npix];              // :193 (no d>0 test)
                c_gather++;
                const double x = ((double)ix - L.K[2]) / L.K[0];
                const double y = ((double)iy - L.K[3]) / L.K[1];
                const double qcx = x * d, qcy = y * d, qcz = d;
                const double qx = row_apply(L.Twk, 0, qcx, qcy, qcz);
                const double qy = row_apply(L.Twk, 1, qcx, qcy, qcz);
                const double qz = row_apply(L.Twk, 2, qcx, qcy, qcz);
                const double w = cov_trace(P, ix, iy, d, L.K[0]);    // :202
                if (w < P.T_cov) {                                   // :204
                    n_obs++;
                    const double dx = qx - pwx, dy = qy - pwy, dz = qz - pwz;
                    const double s = (dx * dx + dy * dy) + dz * dz;  // norm(): (e0+e1)+e2
                    if (s < P.tau_dist2) n_close++;                  // :224
                    const double inv = 1.0 / w;                      // :161-162
                    wx = wx + qx * inv;
                    wy = wy + qy * inv;
                    wz = wz + qz * inv;
                    total = total + inv;
                    hit = npix;
                }
            }
        }
        s_hits[k * kWave + lane] = hit;
    }
    if (n_obs < 1) ok = false;                                       // :218
    if ((double)n_close < (double)n_obs * P.close_frac) ok = false;  // :229

    unsigned long long key = 0, tag = 0;
    double fx = 0, fy = 0, fz = 0;
    uint32_t col = 0;
    if (ok) {
        for (uint32_t k = 0; k < nK; k++) {                          // :152-154
            const uint32_t hit = s_hits[k * kWave + lane];
            if (hit != kNoHit) {
                c_hit++;
                const uint32_t nb = nbr_idx[F.nbr_begin + k];
                // masks of already-processed keyframes are never read again
                if (nb > a.f) frames[nb].mask[hit] = 255;
            }
        }
        fx = wx / total; fy = wy / total; fz = wz / total;           // :164
        const uint8_t *c = F.bgr + 3ull * pix;                       // :170, features.cpp:76-80
        col = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
        tag = ((unsigned long long)F.order << 32) | pix;
        unsigned long long c0, c1, c2;
        const int fits = (int)axis_code(voxel_index(fx, P.voxel_size), c0) &
                         (int)axis_code(voxel_index(fy, P.voxel_size), c1) &
                         (int)axis_code(voxel_index(fz, P.voxel_size), c2);
        if (!fits) {
            atomicOr(&a.ctr->err, kErrKeyRange);
            ok = false;
        } else {
            key = (c0 << 42) | (c1 << 21) | c2;
        }
        c_acc++;
    }
    // wave-aggregated append to the accepted-point list
    const unsigned long long m = __ballot(ok);
    if (m != 0) {
        unsigned long long base = 0;
        const unsigned cnt = (unsigned)__popcll(m);
        if (lane == (unsigned)__ffsll((long long)m) - 1u) base = atomicAdd(&a.ctr->list_count, (unsigned long long)cnt);
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (ok) {
            const unsigned long long idx = base + (unsigned)__popcll(m & lanemask_lt());
#ifndef SFM_ABLATE_LIST
            if (idx < a.list_cap) {
                a.list_key[idx] = key;
                a.list_tag[idx] = tag;
                a.list_xyz[3 * idx + 0] = fx;
                a.list_xyz[3 * idx + 1] = fy;
                a.list_xyz[3 * idx + 2] = fz;
                a.list_bgr[idx] = col;
            } else {
                atomicOr(&a.ctr->err, kErrListFull);
            }
#endif
#ifndef SFM_ABLATE_TABLE
            unsigned probes;
            const uint64_t h = table_insert(a, key, tag, probes);
            if (h == ~0ull) {
                atomicOr(&a.ctr->err, kErrTableFull);
            } else {
                if (probes > a.ctr->max_probe) atomicMax(&a.ctr->max_probe, probes);
                // per-frame voxel filter count (:119-120): first touch of this slot by frame f
                if (a.mark[h] < a.f + 1u) {
                    if (atomicMax(&a.mark[h], a.f + 1u) < a.f + 1u) c_after++;
                }
            }
#endif
        }
    }
}

__device__ __forceinline__ unsigned wave_sum(unsigned v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One launch per keyframe, in processing order (the visibility masks make keyframe k
// depend on every earlier covisible keyframe).  frames/nbr_idx are never
// written by any kernel: __restrict__ lets the compiler keep them on the scalar path.
__global__ __launch_bounds__(64) void fuse_frame_kernel(const FrameDev *__restrict__ frames,
                                                        const uint32_t *__restrict__ nbr_idx,
                                                        FuseArgs a)
{
    __shared__ uint32_t s_queue[kWave + kTile];
    __shared__ uint32_t s_hits[kMaxK * kWave];
    const unsigned lane = lane_id();
    const FrameDev &F = frames[a.f];
    const uint32_t npx = (uint32_t)F.W * (uint32_t)F.H;
    const uint32_t ntiles = (npx + kTile - 1) / kTile;
    unsigned c_gather = 0, c_hit = 0, c_acc = 0, c_after = 0;
    uint32_t qn = 0;  // wave-uniform queue fill
    for (uint32_t tile = blockIdx.x;; tile += gridDim.x) {
        const bool more = tile < ntiles;
        if (more) {
            const uint32_t i0 = tile * kTile + lane * kPxPerLane;
            uint16_t d[kPxPerLane];
            uint8_t mk[kPxPerLane];
            if (i0 + kPxPerLane <= npx) {
                const ushort4 dv = *reinterpret_cast<const ushort4 *>(F.depth + i0);
                const uchar4 mv = *reinterpret_cast<const uchar4 *>(F.mask + i0);
                d[0] = dv.x; d[1] = dv.y; d[2] = dv.z; d[3] = dv.w;
                mk[0] = mv.x; mk[1] = mv.y; mk[2] = mv.z; mk[3] = mv.w;
            } else {
#pragma unroll
                for (int j = 0; j < kPxPerLane; j++) {
                    const bool in = i0 + j < npx;
                    d[j] = in ? F.depth[i0 + j] : (uint16_t)0;
                    mk[j] = in ? F.mask[i0 + j] : (uint8_t)255;
                }
            }
            // ballot / popcount compaction of live pixels (:100-101)
            uint32_t off = qn;
#pragma unroll
            for (int j = 0; j < kPxPerLane; j++) {
                const bool live = (mk[j] == 0) && (d[j] > 0);
                const unsigned long long m = __ballot(live);
                if (live) s_queue[off + (unsigned)__popcll(m & lanemask_lt())] = i0 + j;
                off += (unsigned)__popcll(m);
            }
            qn = off;
        }
        __syncthreads();
        // dense batches of 64; the ragged remainder only once, after the last tile
        while (qn >= (uint32_t)kWave || (!more && qn > 0)) {
            const uint32_t take = qn < (uint32_t)kWave ? qn : (uint32_t)kWave;
            qn -= take;
            const bool act = lane < take;
            const uint32_t pix = act ? s_queue[qn + lane] : 0u;
            process_batch(frames, nbr_idx, a, F, pix, act, s_hits, c_gather, c_hit, c_acc, c_after);
        }
        __syncthreads();
        if (!more) break;
    }
    c_gather = wave_sum(c_gather);
    c_hit = wave_sum(c_hit);
    c_acc = wave_sum(c_acc);
    c_after = wave_sum(c_after);
    if (lane == 0) {
        if (c_gather) atomicAdd(&a.ctr->n_gather, (unsigned long long)c_gather);
        if (c_hit) atomicAdd(&a.ctr->n_hit, (unsigned long long)c_hit);
        if (c_acc) atomicAdd(&a.frame_before[a.f], (unsigned long long)c_acc);
        if (c_after) atomicAdd(&a.frame_after[a.f], (unsigned long long)c_after);
    }
}

// ---- finalize: gather the winners ---------------------------------------------------------

struct FinalizeArgs {
    const Slot *table;
    uint64_t table_mask;
    const unsigned long long *list_key;
    const unsigned long long *list_tag;
    const double *list_xyz;
    const uint32_t *list_bgr;
    uint64_t n_list;
    Counters *ctr;
    // output (SoA, device)
    double *out_xyz;
    uint32_t *out_bgr;
    unsigned long long *out_key;  // packed
    unsigned long long *out_tag;
};

// A list entry is an output point iff its tag is the minimum recorded for its voxel.
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a)
{
    const unsigned lane = lane_id();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (a.n_list + 63) / 64 * 64;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        bool win = false;
        unsigned long long key = 0, tag = 0;
        if (i < a.n_list) {
            key = a.list_key[i];
            tag = a.list_tag[i];
            uint64_t h = mix64(key) & a.table_mask;
            for (;;) {
                const unsigned long long cur = a.table[h].key;
                if (cur == key) { win = a.table[h].tag == tag; break; }
                if (cur == kEmpty) break;  // cannot happen: every listed key was inserted
                h = (h + 1) & a.table_mask;
            }
        }
        const unsigned long long m = __ballot(win);
        if (m != 0) {
            unsigned long long base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if ((int)lane == leader) base = atomicAdd(&a.ctr->out_count, (unsigned long long)__popcll(m));
            base = __shfl(base, leader);
            if (win) {
                const unsigned long long o = base + (unsigned)__popcll(m & lanemask_lt());
                a.out_xyz[3 * o + 0] = a.list_xyz[3 * i + 0];
                a.out_xyz[3 * o + 1] = a.list_xyz[3 * i + 1];
                a.out_xyz[3 * o + 2] = a.list_xyz[3 * i + 2];
                a.out_bgr[o] = a.list_bgr[i];
                a.out_key[o] = key;
                a.out_tag[o] = tag;
            }
        }
    }
}

}  // namespace sfm
