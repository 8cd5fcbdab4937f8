// fusion_kernels.h -- hand-written gfx950 kernels of the dense-fusion path.
//
// Compiled with -ffp-contract=off: every double operation below is a single IEEE-754
// round-to-nearest op in the order the reference evaluates it (reference builds with
// clang -O3, no -march => SSE2, no FMA; reconstruction/CMakeLists.txt:3-5).  The Eigen
// evaluation orders that matter are spelled out next to each function.
//
// Kernels:
//   fuse_chain_kernel   (default) one persistent launch per group of <= 32 keyframes, one 1024-thread
//                       workgroup per CU, a device-wide barrier between keyframes (the visibility
//                       masks chain the keyframes); fuse_chain_kernel_any is the same
//                       body without the XCD-aware pixel mapping, for grids other than 256.
//   scan_frame_kernel, observe_frame_kernel
//                       the fallback: two launches per keyframe, in order; see "the per-keyframe
//                       kernels" below.
//   table_group_kernel  first-wins voxel filter for a group of <= 32 keyframes: open-addressing
//                       hash in HBM, atomicCAS key claim + atomicMin order tag.  Runs on a
//                       second stream, beside the keyframe chain (<= 32 VGPRs: see build.sh).
//   finalize_kernel     gathers a group's winners (and their colours) into the output cloud.
//   partition_*, merge_*  multi-GPU exchange by voxel-key hash and the owners' reduce.
//   iota_kernel, cloud_gather_kernel  the cloud in key order for sfm_fusion_read_points.
#pragma once
#ifndef SFM_OBSERVE_WAVES_PER_EU
#define SFM_OBSERVE_WAVES_PER_EU 6
#endif
#if SFM_OBSERVE_WAVES_PER_EU > 0
#define SFM_OBSERVE_OCC __attribute__((amdgpu_waves_per_eu(SFM_OBSERVE_WAVES_PER_EU, 8)))
#else
#define SFM_OBSERVE_OCC
#endif
#ifndef SFM_FIN_TOP
#define SFM_FIN_TOP 0  // fuse_chain_kernel: 1 = chunks are fused by the last waves of the workgroup instead of the first (measured: +1.4 % on C2)
#endif
#ifndef SFM_TABLE_MARK_CAS
// table_group_kernel's per-keyframe mark: 1 = plain read + compare-and-swap loop (round 2), 0 = mark read with key and tag,
// idempotent atomicMin reset + atomicAnd, no CAS, no retry.  Both are bit-exact on the whole GPU suite;
// measured on C2 (profiles/r03_ab_table_mark.txt) the CAS-free kernel is the FASTER kernel and the SLOWER run (7.78 against
// 7.62 ms at 16 workgroups per shard, 7.74 at its best grid of 12): what it saves is a round trip per entry, what slows
// the chain beside it is its traffic per microsecond, which goes up.
#define SFM_TABLE_MARK_CAS 1
#endif
// table_group_kernel: an entry that repeats its left neighbour's voxel (same keyframe, larger tag) skips the table
// diagnostic (wrong cloud): accept_chain_kernel reads one stored hit per pixel instead of four -- what the bytes cost
#ifndef SFM_DIAG_ACC_ONEHIT
#define SFM_DIAG_ACC_ONEHIT 0
#endif
// (Measured and removed, profiles/r03_ab_accept_chain.txt exp23: the accepted pixels' hit words packed through LDS so that
// the marks leave as full 64-lane instructions instead of sixteen quarter-full ones per wave -- bit-exact, no gain on C3,
// slower on C2: what the marks' acknowledgement waits for is partial-line writes at the memory side, not instructions.)
// accept_chain_kernel: 1 = only the marks into the next keyframe are waited for at the barrier, the marks into later
// keyframes leave behind the arrival (they have one more barrier to go, and the next keyframe's counted wait covers
// them).  Bit-exact; measured no gain on C3 (18.4 ms either way) and slower on C2 (2.94 against 2.75 ms),
// profiles/r03_ab_accept_chain.txt exp24: a quarter of the marks is acknowledged no sooner than all of them, and the
// other three quarters then stand in front of the next keyframe's mask loads.  Off.
#ifndef SFM_ACC_DEFER_FAR_MARKS
#define SFM_ACC_DEFER_FAR_MARKS 0
#endif
// (Measured and removed, profiles/r03_ab_accept_chain.txt exp25: every mark has the scalar load of its mask pointer and a
// wait for it in front of it -- the stores are `asm volatile` with a memory clobber, behind which the compiler reads
// everything again.  With the four pointers read once per keyframe, ahead of the barrier, nothing changes: C3 18.1-18.2
// against 18.1-18.3 ms, C2 2.79 against 2.74.  The scalar cache serves them; the wait is the marks' round trip.)
#ifndef SFM_TABLE_DEDUP
#define SFM_TABLE_DEDUP 1
#endif
#ifndef SFM_DIAG_ACC_NOREC
#define SFM_DIAG_ACC_NOREC 0    // diagnostic only (wrong cloud): accept_chain_kernel writes tags without reading or writing the voxel keys
#endif
#ifndef SFM_PREP_FLIP
#define SFM_PREP_FLIP 0
#endif
#ifndef SFM_FIN_ROT
#define SFM_FIN_ROT 0        // fuse_chain_kernel: chunk j of a round is fused by wave (j + SFM_FIN_ROT) mod 16 (4, 8: measured slower)
#endif
#ifndef SFM_WAIT_THREAD
// Who does what in a fuse_chain_kernel workgroup (profiles/r03_ab_wave_roles.txt, C2): the waves that fuse (0 .. chunks-1)
// should neither wait for the arrival atomic's return nor poll.  Arrival from wave 8, polling from wave 15: 7.46 against
// 7.60 ms with both on thread 0 (arrival on waves 5 / 15: 7.63 / 7.55; polling on waves 7 / 8: 7.46 / 7.59).
#define SFM_WAIT_THREAD 960  // the thread that polls the barrier flag
#endif
#ifndef SFM_ARRIVE_THREAD
#define SFM_ARRIVE_THREAD 512  // the thread that makes the barrier arrival (and waits for its atomic's return)
#endif
#ifndef SFM_STAMPS
#define SFM_STAMPS 0  // diagnostic build only (tools/chain_stamps.py): adds fuse_chain_kernel_stamps, thread-0 phase timestamps of one launch
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfm {

constexpr int kWave = 64;
constexpr int kPxPerLane = 4;
constexpr int kTile = kWave * kPxPerLane;  // 256 pixels per workgroup tile
constexpr int kMaxK = 32;                  // SFM_MAX_NEIGHBORS: filtered neighbours the fused kernels hold; more take the two-phase chain
constexpr int kMaxKTwoPhase = 4095;        // ... whose per-pixel counters are 12 bits wide
constexpr int kMaxWaves = 8;               // waves per workgroup of fuse_frame_kernel
constexpr int kListShards = 16;            // append cursors (one 64-B line each)
constexpr int kStatShards = 64;
constexpr int kGroup = 32;                 // keyframes per table_group_kernel launch
constexpr uint64_t kEmpty = ~0ull;
constexpr uint32_t kNoHit = 0xFFFFFFFFu;

// error bits
constexpr uint32_t kErrTableFull = 1u;
constexpr uint32_t kErrKeyRange = 2u;
constexpr uint32_t kErrListFull = 4u;
constexpr uint32_t kErrBarrier = 8u;  // fuse_chain_kernel: a device-wide barrier timed out
constexpr uint32_t kErrOutFull = 16u;  // finalize_kernel: more winners than output slots
constexpr uint32_t kErrDivRange = 32u; // fuse_chain_kernel: a division operand outside div_pair's range
constexpr uint32_t kErrDivRangeA = 64u; // host only: the same from verdict_kernel (phase A of the two-phase chain)

struct FrameDev {
    double K[4];      // fx fy cx cy
    double Twk[12];   // rows 0..2 of camera->world, row-major [r*4+c]
    double Tkw[12];   // rows 0..2 of world->camera
    const uint16_t *depth;  // W*H, packed rows
    const uint8_t *bgr;     // W*H*3
    uint8_t *mask;          // W*H visibility mask (dense_reconstruction.hpp:35-39)
    // exact lookup tables shared by keyframes with the same intrinsics (build_*_table below)
    const double *xtab;     // [W]  (u - cx) / fx                        pinhole.cpp:12
    const double *ytab;     // [H]  (v - cy) / fy                        pinhole.cpp:13
    const double *ctab;     // [65536][4] depth-only terms of trace(J S J^T), see cov_trace_tab
    int32_t W, H;
    uint32_t nbr_begin, nbr_count;  // filtered neighbours in nbr_idx[]
    uint32_t order, pad;
};

// Per axis, the voxel indices a run's key packing holds: [lo, hi] plus x86's "indefinite" index.  Also what a
// range-scan run reports: the smallest and largest index of any fused point (kStatShards copies, combined on the host).
struct KeyRangeDev {
    long long lo[3], hi[3];
};

struct DevParams {
    double base_line, T_cov, voxel_size;
    double tau_dist2;  // smallest s with sqrt(s) >= T_dist: "norm < T_dist"  <=>  s < tau_dist2
    double S0, S1, S2, depth_div, close_frac;
    // The voxel key (three size_t, :270-274) packed EXACTLY into one 64-bit word.  Per axis a code of key_bits[a] bits:
    //   index in [max(lo,0), hi] -> index - max(lo,0);  the indefinite index -> n_nonneg;  index in [lo, -1] -> n_nonneg
    //   + 1 + (index - lo)   -- ascending codes = ascending unsigned size_t, so that sorting packed keys sorts the
    // reference's tuples.  Default: lo = -(2^20-1), hi = 2^20-1, 21 bits, i.e. plain two's complement with 0x100000 for
    // the indefinite index (key_wide == 0: 32-bit conversions only).  A run whose indices do not fit is followed by a
    // range-scan run (key_scan) and a run with the ranges it found; exact while the widths sum to <= 63 bits.
    KeyRangeDev key;
    uint32_t key_bits[3];
    uint32_t key_wide;       // not the default packing: indices go through 64-bit conversions
    uint32_t key_scan;       // range-scan run: record every fused point's indices in key_scan_out, emit no records
    KeyRangeDev *key_scan_out;  // [kStatShards]
};

struct Slot {  // 32 bytes: everything an insert touches sits in one 64-byte sector
    unsigned long long key;   // packed 3x21-bit voxel key, kEmpty when free
    unsigned long long tag;   // min over inserted (order<<32 | pixel); kEmpty when free
    unsigned long long mark;  // (group+1) << 32 | bitmask of the group's keyframes that touched the voxel
    unsigned long long pad;
};

struct StatShard {  // one 64-byte line per shard
    unsigned long long n_gather;
    unsigned long long n_hit;
    unsigned int err;
    unsigned int pad[11];
};

struct Counters {
    unsigned long long out_count;
    unsigned long long table_used;
    unsigned long long cand_count;  // entries of the winner-candidate index list
    unsigned int err;
    unsigned int max_probe;
};

// Candidate-record list: kListShards independent append regions.  Shard s owns entries
// [s*shard_cap, (s+1)*shard_cap).  Every live pixel reserves one entry; rejected pixels
// leave tag == kEmpty.
struct ListDev {
    unsigned long long *key;
    unsigned long long *tag;
    double *xyz;             // 3 per entry
    uint64_t shard_cap;
    unsigned long long *cursor;     // [kListShards * 8]  (stride 8 => one line per cursor)
    // [n_groups][kListShards * 8]: copy of cursor[] taken on the keyframe stream after the last
    // keyframe of each group (a 1-KiB device-to-device copy), i.e. the groups' list boundaries
    unsigned long long *group_end;
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

// n0/d and n1/d, bit for bit what two IEEE divisions give.  Shared modes: the reciprocal and its
// refinement are computed once.  It is the compiler's own f64 division sequence (v_div_scale, v_rcp, two Newton
// steps, quotient, residual, v_div_fmas, v_div_fixup) for operands that v_div_scale leaves alone: |d|
// and every |n| either zero, infinite, NaN (v_div_fixup decides those from the operands, whatever the
// arithmetic produced) or within [2^-400, 2^400].  A finite non-zero operand outside that range sets
// `bad` in the checked mode (tests/hip/div_probe.hip); the kernels use the unchecked mode and the host
// (chain_div_ok) sends inputs that could leave the range to the kernel pair, which divides plainly.  A pair costs 17 instructions instead of 22 and one v_rcp_f64 (16 cycles) instead of two.
// true: x is finite, not zero, and outside [2^-400, 2^400] (from the exponent field alone)
__device__ __forceinline__ bool div_operand_bad(double x)
{
    const unsigned e = ((unsigned)__double2hiint(x) >> 20) & 0x7FFu;
    if (e == 0u) return x != 0.0;                 // denormal
    return e != 0x7FFu && (e - 623u) > 800u;     // 623 = 1023 - 400
}

enum DivMode { kDivPlain, kDivShared, kDivSharedChecked };
// -DSFM_DIV_CHECKED=1 (build.sh: libsfm_fusion_checked.so, loaded by tests/test_gpu_div_checked.py only): the kernels'
// shared divisions check their operands and raise kErrDivRange, which the host counts as a chain fallback -- the
// test that the host-side bound (chain_div_ok) never lets an operand outside the range through
#ifndef SFM_DIV_CHECKED
#define SFM_DIV_CHECKED 0
#endif
constexpr DivMode kDivSharedMode = SFM_DIV_CHECKED ? kDivSharedChecked : kDivShared;
// the part of a shared division that depends on the denominator alone: v_rcp_f64 and two Newton steps
__device__ __forceinline__ double div_recip(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
// ... and one quotient from it: n/d for operands inside the range (r = div_recip(d))
__device__ __forceinline__ double div_with(double n, double d, double r)
{
    const double t = n * r;
    return __builtin_amdgcn_div_fixup(__builtin_fma(__builtin_fma(-d, t, n), r, t), d, n);
}

template <DivMode MODE>
__device__ __forceinline__ void div_pair(double n0, double n1, double d, double &q0, double &q1, bool &bad)
{
    if (MODE == kDivPlain) {
        q0 = n0 / d;
        q1 = n1 / d;
        return;
    }
    // kDivShared: the caller's operands were validated elsewhere (host, sfm_fusion_run)
    if (MODE == kDivSharedChecked) bad = bad || div_operand_bad(d) || div_operand_bad(n0) || div_operand_bad(n1);
    const double r = div_recip(d);
    q0 = div_with(n0, d, r);
    q1 = div_with(n1, d, r);
}

// n0/d, n1/d, n2/d, bit for bit three IEEE divisions, for ANY operands: the wave looks at its operands and takes the
// shared-reciprocal form only when every active lane's are inside its range, else it divides plainly.  For the two
// triples of the fusion step -- the weighted sums over their total (:164) and the fused coordinates over the voxel
// size (:270-274) -- whose operands (sums that may cancel) have no bound that the host could check ahead.
// SHARED = false: three plain divisions.  Measured on C2 (tools/ab_bench.sh, profiles/r03_ab_fusion_step.txt): in
// fuse_chain_kernel the shared form is 3.3 % SLOWER (the fusion step is a dependent chain on a few waves, not an issue
// bound: the operand checks lengthen it), so the chain keeps plain divisions there.
#ifndef SFM_DIV_TRIPLE_CHAIN
#define SFM_DIV_TRIPLE_CHAIN 0
#endif
#ifndef SFM_DIV_TRIPLE_VERDICT
#define SFM_DIV_TRIPLE_VERDICT 1
#endif
template <bool SHARED>
__device__ __forceinline__ void div_triple(double n0, double n1, double n2, double d, bool active, double &q0, double &q1,
                                           double &q2)
{
    const bool bad = SHARED && active && (div_operand_bad(d) || div_operand_bad(n0) || div_operand_bad(n1) || div_operand_bad(n2));
    if (!SHARED || __ballot(bad) != 0ull) {  // wave-uniform
        q0 = n0 / d;
        q1 = n1 / d;
        q2 = n2 / d;
        return;
    }
    const double r = div_recip(d);
    q0 = div_with(n0, d, r);
    q1 = div_with(n1, d, r);
    q2 = div_with(n2, d, r);
}

// Same value as cov_trace(), bit for bit, with the four divisions that depend on the depth
// alone taken from the per-focal-length table e = ctab + 4*depth:
//   e[0] = (J00*S0)*J00, e[1] = (J00*S1)*J00, e[2] = disparity^2, e[3] = (J22*S2)*J22.
template <bool SHARED>
__device__ __forceinline__ double cov_trace_vals(const DevParams &P, int u, int v, double2 e01, double2 e23, bool &bad)
{
    const double dd = e23.x;
    double J02, J12;
    // shared: sfm_fusion_run has checked the table's range of disparity^2 and of the numerators (chain_div_ok)
    div_pair<SHARED ? kDivSharedMode : kDivPlain>((double)(-u) * P.base_line, (double)(-v) * P.base_line, dd, J02, J12, bad);
    const double P00 = e01.x + (J02 * P.S2) * J02;
    const double P11 = e01.y + (J12 * P.S2) * J12;
    return P00 + (P11 + e23.y);
}

__device__ __forceinline__ double cov_trace_tab(const DevParams &P, int u, int v, const double *e)
{
    bool bad = false;
    return cov_trace_vals<false>(P, u, v, *reinterpret_cast<const double2 *>(e), *reinterpret_cast<const double2 *>(e + 2), bad);
}

// One entry per raw depth value for focal length f; the operations and their order are those
// of cov_trace().
__global__ __launch_bounds__(256) void build_cov_table(double *tab, double f, DevParams P)
{
    const unsigned d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= 65536u) return;
    const double disp = (f * P.base_line) / ((double)d / P.depth_div);
    const double dd = disp * disp;
    const double J00 = P.base_line / disp;
    const double J22 = ((-f) * P.base_line) / dd;
    tab[4 * d + 0] = (J00 * P.S0) * J00;
    tab[4 * d + 1] = (J00 * P.S1) * J00;
    tab[4 * d + 2] = dd;
    tab[4 * d + 3] = (J22 * P.S2) * J22;
}

// tab[i] = (i - c) / f   (pinhole.cpp:12-13)
__global__ __launch_bounds__(256) void build_axis_table(double *tab, int n, double c, double f)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[i] = ((double)i - c) / f;
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

// Default packing: 21-bit two's-complement code of static_cast<size_t>(coord / vs); 0x100000 is reserved for the
// indefinite index (NaN, >= 2^64, <= -2^63 -- voxel_index's INT64_MIN cases).  32-bit conversions only.
// Returns false when the index does not fit.
__device__ __forceinline__ bool axis_code_of(double coord, double vs, unsigned long long &code)
{
    const double q = coord / vs;
    double t = trunc(q);
    // [2^63, 2^64) converts to the size_t bits of the negative index t - 2^64 (exact subtraction)
    if (t >= 9223372036854775808.0 && t < 18446744073709551616.0) t = t - 18446744073709551616.0;
    if (t >= -1048575.0 && t <= 1048575.0) {
        code = (unsigned long long)(unsigned int)(int)t & 0x1FFFFFull;
        return true;
    }
    code = 0x100000ull;
    return (q != q) || t >= 18446744073709551616.0 || t <= -9223372036854775808.0;
}

// static_cast<size_t>(q), q = coord / voxel_size, as x86-64 evaluates it, returned
// as a signed 64-bit index; INT64_MIN stands for the 0x8000000000000000 "indefinite" key.
__device__ __forceinline__ long long voxel_index_q(double q)
{
    if (q != q) return INT64_MIN;
    const double t = trunc(q);
    if (t >= -9223372036854775808.0 && t < 9223372036854775808.0) return (long long)t;
    if (t >= 9223372036854775808.0 && t < 18446744073709551616.0)
        return (long long)(unsigned long long)t;
    return INT64_MIN;
}

// Default packing: 21-bit two's-complement code of static_cast<size_t>(q); 0x100000 is reserved for the
// indefinite index (NaN, >= 2^64, <= -2^63 -- voxel_index_q's INT64_MIN cases).  32-bit conversions only.
// Returns false when the index does not fit.
__device__ __forceinline__ bool axis_code_of_q(double q, unsigned long long &code)
{
    double t = trunc(q);
    // [2^63, 2^64) converts to the size_t bits of the negative index t - 2^64 (exact subtraction)
    if (t >= 9223372036854775808.0 && t < 18446744073709551616.0) t = t - 18446744073709551616.0;
    if (t >= -1048575.0 && t <= 1048575.0) {
        code = (unsigned long long)(unsigned int)(int)t & 0x1FFFFFull;
        return true;
    }
    code = 0x100000ull;
    return (q != q) || t >= 18446744073709551616.0 || t <= -9223372036854775808.0;
}

// General packing (DevParams::key): code of index k (INT64_MIN = indefinite) within [lo, hi]
__host__ __device__ inline bool axis_code_range(long long k, long long lo, long long hi, unsigned long long &code)
{
    const long long nn0 = lo > 0 ? lo : 0;                                   // first non-negative index held
    const unsigned long long n_nonneg = hi >= 0 ? (unsigned long long)(hi - nn0) + 1ull : 0ull;
    if (k == INT64_MIN) { code = n_nonneg; return true; }
    if (k < lo || k > hi) { code = n_nonneg; return false; }
    code = k >= 0 ? (unsigned long long)(k - nn0) : n_nonneg + 1ull + (unsigned long long)(k - lo);
    return true;
}

// ... and back to the reference's size_t key component
__host__ __device__ inline unsigned long long axis_decode_range(unsigned long long code, long long lo, long long hi)
{
    const long long nn0 = lo > 0 ? lo : 0;
    const unsigned long long n_nonneg = hi >= 0 ? (unsigned long long)(hi - nn0) + 1ull : 0ull;
    if (code < n_nonneg) return (unsigned long long)(nn0 + (long long)code);
    if (code == n_nonneg) return 0x8000000000000000ull;
    return (unsigned long long)(lo + (long long)(code - n_nonneg - 1ull));
}

// codes an axis [lo, hi] needs: its indices and the indefinite one
__host__ __device__ inline unsigned long long axis_code_count(long long lo, long long hi)
{
    return (unsigned long long)(hi - lo) + 2ull;  // (lo <= hi; callers keep hi - lo < 2^53)
}

// The packed key of a fused point from its three quotients coord / voxel_size (:270-274).  false: an index does not
// fit the run's packing (the caller raises kErrKeyRange), or this is a range-scan run (every point is "unfit" and
// leaves its indices in key_scan_out).
__device__ __forceinline__ bool pack_voxel_key_q(const DevParams &P, double qx, double qy, double qz, unsigned long long &key)
{
    unsigned long long c0, c1, c2;
    if (!P.key_wide && !P.key_scan) {  // wave-uniform
        const int fits = (int)axis_code_of_q(qx, c0) & (int)axis_code_of_q(qy, c1) & (int)axis_code_of_q(qz, c2);
        key = (c0 << 42) | (c1 << 21) | c2;
        return fits != 0;
    }
    const long long i0 = voxel_index_q(qx), i1 = voxel_index_q(qy), i2 = voxel_index_q(qz);
    if (P.key_scan) {
        KeyRangeDev *r = P.key_scan_out + (blockIdx.x + threadIdx.x / 64u) % kStatShards;
        const long long v[3] = {i0, i1, i2};
        for (int a = 0; a < 3; a++) {
            if (v[a] == INT64_MIN) continue;
            // (the plain reads may be stale: then an atomic too many is issued, never one too few)
            if (v[a] < r->lo[a]) __hip_atomic_fetch_min(&r->lo[a], v[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v[a] > r->hi[a]) __hip_atomic_fetch_max(&r->hi[a], v[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        key = 0;
        return false;
    }
    const int fits = (int)axis_code_range(i0, P.key.lo[0], P.key.hi[0], c0) & (int)axis_code_range(i1, P.key.lo[1], P.key.hi[1], c1) &
                     (int)axis_code_range(i2, P.key.lo[2], P.key.hi[2], c2);
    key = (c0 << (P.key_bits[1] + P.key_bits[2])) | (c1 << P.key_bits[2]) | c2;
    return fits != 0;
}

// ... from the coordinates, with plain divisions (round 2's form, kept verbatim: fuse_chain_kernel's code generation
// is sensitive to it)
__device__ __forceinline__ bool pack_voxel_key(const DevParams &P, double fx, double fy, double fz, unsigned long long &key)
{
    unsigned long long c0, c1, c2;
    if (!P.key_wide && !P.key_scan) {  // wave-uniform
        const int fits = (int)axis_code_of(fx, P.voxel_size, c0) & (int)axis_code_of(fy, P.voxel_size, c1) &
                         (int)axis_code_of(fz, P.voxel_size, c2);
        key = (c0 << 42) | (c1 << 21) | c2;
        return fits != 0;
    }
    const long long i0 = voxel_index(fx, P.voxel_size), i1 = voxel_index(fy, P.voxel_size), i2 = voxel_index(fz, P.voxel_size);
    if (P.key_scan) {
        KeyRangeDev *r = P.key_scan_out + (blockIdx.x + threadIdx.x / 64u) % kStatShards;
        const long long v[3] = {i0, i1, i2};
        for (int a = 0; a < 3; a++) {
            if (v[a] == INT64_MIN) continue;
            // (the plain reads may be stale: then an atomic too many is issued, never one too few)
            if (v[a] < r->lo[a]) __hip_atomic_fetch_min(&r->lo[a], v[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v[a] > r->hi[a]) __hip_atomic_fetch_max(&r->hi[a], v[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        key = 0;
        return false;
    }
    const int fits = (int)axis_code_range(i0, P.key.lo[0], P.key.hi[0], c0) & (int)axis_code_range(i1, P.key.lo[1], P.key.hi[1], c1) &
                     (int)axis_code_range(i2, P.key.lo[2], P.key.hi[2], c2);
    key = (c0 << (P.key_bits[1] + P.key_bits[2])) | (c1 << P.key_bits[2]) | c2;
    return fits != 0;
}


// Every NaN the reference can produce on this path is x86's "real indefinite" (sign bit set, quiet,
// zero payload: SSE2 returns it for 0*inf, inf-inf, inf/inf, and propagates it unchanged); gfx950
// generates the positive quiet NaN.  The fused coordinates leave the kernels with the reference's bits.
__device__ __forceinline__ double x86_nan(double v)
{
    return v != v ? __longlong_as_double((long long)0xFFF8000000000000ull) : v;
}

// The tail of process_pixel for one accepted pixel in the chain kernels: the fused point (:164) and its packed voxel
// key (:270-274), six divisions as two shared-reciprocal triples.  `ok` = the lane holds an accepted pixel.
template <bool SHARED>
__device__ __forceinline__ bool fuse_point_and_key(const DevParams &P, double wx, double wy, double wz, double total, bool ok,
                                                   double &fx, double &fy, double &fz, unsigned long long &key)
{
    div_triple<SHARED>(wx, wy, wz, total, ok, fx, fy, fz);
    fx = x86_nan(fx); fy = x86_nan(fy); fz = x86_nan(fz);
    double qx, qy, qz;
    div_triple<SHARED>(fx, fy, fz, P.voxel_size, ok, qx, qy, qz);
    if (!ok) {  // (a range-scan run records the indices of every lane that packs)
        key = 0;
        return false;
    }
    return pack_voxel_key_q(P, qx, qy, qz, key);
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Loads through pointers that were themselves loaded from memory (FrameDev fields) are "flat" to the
// compiler; flat loads also count on lgkmcnt and so tie every scalar-load wait to the slowest
// gather in flight.  All image/table memory is global: say so.
typedef double dbl2_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ T ldg(const T *p)
{
    return *(const __attribute__((address_space(1))) T *)p;
}
__device__ __forceinline__ double2 ldg2(const double *p)
{
    const dbl2_t v = *(const __attribute__((address_space(1))) dbl2_t *)p;
    return make_double2(v.x, v.y);
}
// 4 bytes at any byte address in one load (HSA devices access global memory unaligned; the colour
// buffers end with 8 bytes of slack)
typedef uint32_t u32_unaligned_t __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ldg_u32_unaligned(const uint8_t *p)
{
    return *(const __attribute__((address_space(1))) u32_unaligned_t *)p;
}
// visibility-mask accesses of fuse_chain_kernel: agent-scope relaxed atomics (write-through store,
// cache-bypassing load), see the kernel's header comment
__device__ __forceinline__ uint32_t mask_load_coherent(const uint8_t *p)
{
    return __hip_atomic_load((const __attribute__((address_space(1))) uint8_t *)p, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mask_store_coherent(uint8_t *p)
{
    __hip_atomic_store((__attribute__((address_space(1))) uint8_t *)p, (uint8_t)255, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}


__device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << lane_id()) - 1ull;
}

// number of set bits of m below this lane (v_mbcnt: no lane mask kept in registers)
__device__ __forceinline__ unsigned lane_rank(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__device__ __forceinline__ unsigned wave_sum(unsigned v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- the per-keyframe kernels ------------------------------------------------------------
//
// Two launches per keyframe, strictly in keyframe order (the visibility masks chain the
// keyframes):
//
//   scan_frame_kernel     one wave per 256-pixel tile: coalesced ushort4/uchar4 loads,
//                         ballot/popcount compaction of live pixels (:100-101), covariance gate
//                         (:136-140) and back-projection (:143-144) on dense batches, survivors
//                         (pixel, world point) appended to the keyframe's queue in HBM.
//   observe_frame_kernel  one workgroup per 64 queued survivors -- every chunk is full, every
//                         workgroup does the same amount of work whatever the live-pixel
//                         distribution over the image is.  Wave w re-observes the chunk in
//                         neighbours w, w+NW, ...; the waves meet in LDS; every wave scatters the
//                         masks of its own neighbours; one wave forms the fused point and the record.

// Survivor queue of one keyframe: kQueueShards independent append regions (one cursor each on
// its own 64-B line), refilled by every scan and drained by the observe kernel that follows.
constexpr int kQueueShards = 16;

struct QueueDev {
    uint32_t *pix;        // [kQueueShards][shard_cap]
    double *pw;           // [3][kQueueShards * shard_cap]  world point, SoA
    uint32_t shard_cap;   // entries per shard (>= 256 * ceil(tiles / kQueueShards))
    unsigned int *count;  // this keyframe's cursors: [kQueueShards * 16] (stride 16 u32)
};

struct ScanArgs {
    FrameDev F;
    QueueDev q;
    DevParams P;
};

__global__ __launch_bounds__(kWave) void scan_frame_kernel(ScanArgs a)
{
    __shared__ uint32_t s_pix[kTile];
    __shared__ uint16_t s_d[kTile];
    __shared__ uint32_t s_spix[kTile];
    __shared__ double s_pw[3 * kTile];
    const unsigned lane = lane_id();
    const FrameDev &F = a.F;
    const DevParams &P = a.P;
    const uint32_t npx = (uint32_t)F.W * (uint32_t)F.H;
    // -- live pixels of the tile (:100-101); a keyframe without neighbours yields nothing (:180)
    uint32_t qn = 0;
    {
        const uint32_t i0 = blockIdx.x * kTile + lane * kPxPerLane;
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
#pragma unroll
        for (int j = 0; j < kPxPerLane; j++) {
            const bool live = (mk[j] == 0) && (d[j] > 0) && (F.nbr_count > 0);
            const unsigned long long m = __ballot(live);
            if (live) {
                const unsigned pos = qn + (unsigned)__popcll(m & lanemask_lt());
                s_pix[pos] = i0 + j;
                s_d[pos] = d[j];
            }
            qn += (unsigned)__popcll(m);
        }
    }
    if (qn == 0) return;
    __syncthreads();  // one-wave workgroup: orders the LDS writes above before the reads below
    // -- covariance gate + back-projection on dense batches of 64
    uint32_t ns = 0;
    for (uint32_t c0 = 0; c0 < qn; c0 += kWave) {
        const bool act = c0 + lane < qn;
        const uint32_t pix = act ? s_pix[c0 + lane] : 0u;
        const uint32_t draw = act ? (uint32_t)s_d[c0 + lane] : 1u;
        const double dsrc = (double)draw;
        const int u = (int)(pix % (uint32_t)F.W);
        const int v = (int)(pix / (uint32_t)F.W);
        const double tr = cov_trace_tab(P, u, v, F.ctab + 4 * (size_t)draw);  // :136
        const bool pass = act && !(tr > P.T_cov);                             // :138
        const unsigned long long m = __ballot(pass);
        if (pass) {
            const double xn = F.xtab[u];                                      // pinhole.cpp:12
            const double yn = F.ytab[v];
            const double pcx = xn * dsrc, pcy = yn * dsrc, pcz = dsrc;        // :15-19
            const unsigned pos = ns + (unsigned)__popcll(m & lanemask_lt());
            s_spix[pos] = pix;
            s_pw[pos] = row_apply(F.Twk, 0, pcx, pcy, pcz);                   // :144
            s_pw[kTile + pos] = row_apply(F.Twk, 1, pcx, pcy, pcz);
            s_pw[2 * kTile + pos] = row_apply(F.Twk, 2, pcx, pcy, pcz);
        }
        ns += (unsigned)__popcll(m);
    }
    if (ns == 0) return;
    __syncthreads();
    // -- one reservation per tile, then a coalesced copy into the keyframe's queue
    const uint32_t shard = blockIdx.x % kQueueShards;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&a.q.count[shard * 16], ns);
    base = __shfl(base, 0);
    const size_t plane = (size_t)kQueueShards * a.q.shard_cap;
    for (uint32_t i = lane; i < ns; i += kWave) {
        const size_t o = (size_t)shard * a.q.shard_cap + base + i;
        a.q.pix[o] = s_spix[i];
        a.q.pw[o] = s_pw[i];
        a.q.pw[plane + o] = s_pw[kTile + i];
        a.q.pw[2 * plane + o] = s_pw[2 * kTile + i];
    }
}

// LDS per workgroup of observe_frame_kernel (dynamic, K = filtered neighbours):
//   s_obs  [K][4][64] double   per neighbour, per lane: q*(1/w) (x,y,z) and 1/w
//   s_hit  [K][64]    u32      re-observed pixel in the neighbour
//   s_kept [K], s_close[K] u64 ballots: observation kept (:204) / within T_dist (:224)
__host__ __device__ inline size_t fuse_lds_bytes(uint32_t K)
{
    return (size_t)K * (4 * 64 * 8 + 64 * 4 + 16) + 16;
}

struct FuseArgs {
    FrameDev F;                // the keyframe being fused (by value: arrives with the kernargs)
    QueueDev q;
    ListDev list;
    StatShard *stat;           // [kStatShards]
    DevParams P;
};

// One re-observation, split at its two memory round trips so that observe_frame_kernel can keep
// two neighbours in flight per wave.
struct ObsProj {
    uint32_t npix;  // floored projection as a pixel index (0 when not visible)
    int ix, iy;
    bool vis;
};

// KeyFrame::point_in_frame, keyframe.cpp:275-294 / pinhole.cpp:32-45
template <bool SHARED>
__device__ __forceinline__ ObsProj obs_project(const FrameDev &L, double pwx, double pwy, double pwz, bool on, bool &bad)
{
    const double cx_ = row_apply(L.Tkw, 0, pwx, pwy, pwz);
    const double cy_ = row_apply(L.Tkw, 1, pwx, pwy, pwz);
    const double cz_ = row_apply(L.Tkw, 2, pwx, pwy, pwz);
    double xz, yz;
    // shared: sfm_fusion_run has bounded the poses and intrinsics so that cx_, cy_, cz_ are zero or within
    // div_pair's range (chain_div_ok); a check here costs more than the shared reciprocal saves
    div_pair<SHARED ? kDivSharedMode : kDivPlain>(cx_, cy_, cz_, xz, yz, bad);
    const double px = xz * L.K[0] + L.K[2];
    const double py = yz * L.K[1] + L.K[3];
    ObsProj r;
    // NaN projections are "not visible" (UB in the reference): hence the >= 0 form
    r.vis = on && (px >= 0) && (px < (double)L.W) && (py >= 0) && (py < (double)L.H) && !(cz_ < 0);
    r.ix = r.vis ? (int)floor(px) : 0;
    r.iy = r.vis ? (int)floor(py) : 0;
    r.npix = (uint32_t)r.iy * (uint32_t)L.W + (uint32_t)r.ix;
    return r;
}

// dense_reconstruction.cpp:193-216 after the gathers: back-projection of the neighbour's depth,
// its covariance trace (bit-equal to cov_trace_tab), the kept (:204) / close (:224) tests, and the
// weighted observation (:161-162): q*(1/w) and 1/w (valid where kept).
struct ObsVal {
    double wx, wy, wz, inv;
    bool kept, close;
};

template <bool SHARED>
__device__ __forceinline__ ObsVal obs_values(const FrameDev &L, const DevParams &P, const ObsProj &p, uint32_t draw,
                                             double x, double y, double2 e01, double2 e23, double pwx, double pwy,
                                             double pwz, bool &bad)
{
    const double d = (double)draw;
    const double qcx = x * d, qcy = y * d, qcz = d;
    const double qx = row_apply(L.Twk, 0, qcx, qcy, qcz);
    const double qy = row_apply(L.Twk, 1, qcx, qcy, qcz);
    const double qz = row_apply(L.Twk, 2, qcx, qcy, qcz);
    const double wgt = cov_trace_vals<SHARED>(P, p.ix, p.iy, e01, e23, bad);  // :202
    ObsVal r;
    r.kept = p.vis && (wgt < P.T_cov);                                   // :204
    const double dx = qx - pwx, dy = qy - pwy, dz = qz - pwz;
    const double s = (dx * dx + dy * dy) + dz * dz;                      // norm(): (e0+e1)+e2
    r.close = r.kept && (s < P.tau_dist2);                               // :224
    r.inv = 1.0 / wgt;                                                   // :161-162
    r.wx = qx * r.inv;
    r.wy = qy * r.inv;
    r.wz = qz * r.inv;
    return r;
}

// ... left in LDS for the fusing wave (the chain kernels)
template <bool SHARED>
__device__ __forceinline__ void obs_finish(const FrameDev &L, const DevParams &P, const ObsProj &p, uint32_t draw,
                                           double x, double y, double2 e01, double2 e23, double pwx, double pwy,
                                           double pwz, double *o, uint32_t *hit, bool &kept, bool &close, bool &bad)
{
    const ObsVal r = obs_values<SHARED>(L, P, p, draw, x, y, e01, e23, pwx, pwy, pwz, bad);
    kept = r.kept;
    close = r.close;
    if (kept) {
        o[0] = r.wx;
        o[64] = r.wy;
        o[128] = r.wz;
        o[192] = r.inv;
        *hit = p.npix;
    }
}

// geometric_check + the tail of process_pixel (dense_reconstruction.cpp:146-175, :177-234).
// nbp = this keyframe's filtered neighbours, one FrameDev copy each, in neighbour order
// (never written by any kernel: __restrict__ keeps the reads on the scalar path).
__global__ __launch_bounds__(kMaxWaves * kWave) SFM_OBSERVE_OCC void observe_frame_kernel(
    const FrameDev *__restrict__ nbp, FuseArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const unsigned lane = lane_id();
    // wave index as a scalar: everything indexed by it stays wave-uniform (SGPRs, s_load)
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned NW = blockDim.x >> 6;
    const FrameDev &F = a.F;
    const DevParams &P = a.P;
    const uint32_t K = F.nbr_count;
    double *s_obs = reinterpret_cast<double *>(smem);
    uint32_t *s_hit = reinterpret_cast<uint32_t *>(s_obs + (size_t)K * 256);
    unsigned long long *s_kept = reinterpret_cast<unsigned long long *>(s_hit + (size_t)K * 64);
    unsigned long long *s_close = s_kept + K;

    // Chunks of 64 survivors.  Chunk c is the (c / 16)-th chunk of queue shard c % 16: where its
    // data lives does not depend on the shard counts, so the loads go out together with the
    // count reads instead of one memory round trip behind them.
    const size_t plane = (size_t)kQueueShards * a.q.shard_cap;
    unsigned c_gather = 0, c_hit = 0;
    uint32_t max_cnt = 0;
#pragma unroll
    for (int s = 0; s < kQueueShards; s++) max_cnt = max(max_cnt, a.q.count[s * 16]);
    const uint32_t total_chunks = ((max_cnt + kWave - 1) / kWave) * kQueueShards;
    for (uint32_t c = blockIdx.x; c < total_chunks; c += gridDim.x) {
        const uint32_t shard = c % kQueueShards, j = c / kQueueShards;
        const uint32_t cnt_s = a.q.count[shard * 16];
        const uint32_t qo = min(j * kWave + lane, a.q.shard_cap - 1);  // in bounds whatever cnt_s is
        const size_t qi = (size_t)shard * a.q.shard_cap + qo;
        const double pwx = a.q.pw[qi];
        const double pwy = a.q.pw[plane + qi];
        const double pwz = a.q.pw[2 * plane + qi];
        const uint32_t pix_q = a.q.pix[qi];
        if (j * kWave >= cnt_s) continue;  // workgroup-uniform: no chunk here (shards differ in length)
        const uint32_t n_in = min(cnt_s - j * kWave, (uint32_t)kWave);
        const bool act = lane < n_in;
        const unsigned p3 = c % NW;  // the wave that finishes this chunk (spreads over the SIMDs)
        uint32_t pix = 0;
        unsigned long long base_r = 0;
        if (w == p3) {
            // list space for the chunk: issued now, consumed after the neighbour work
            if (lane == 0) base_r = atomicAdd(&a.list.cursor[(c % kListShards) * 8], (unsigned long long)n_in);
            if (act) pix = pix_q;
        }
        // wave w re-observes the chunk in neighbours w, w+NW, ...  (:187-216), two at a time:
        // both projections, then both depth gathers, then both table reads are in flight
        // together (branch-free; a lane without a visible projection reads element 0)
        for (uint32_t k0 = w; k0 < K; k0 += 2 * NW) {
            const uint32_t k1 = k0 + NW;
            const bool has1 = k1 < K;  // wave-uniform
            const FrameDev &L0 = nbp[k0];
            const FrameDev &L1 = nbp[has1 ? k1 : k0];
            bool nobad = false;  // plain divisions on this path
            ObsProj p0 = obs_project<false>(L0, pwx, pwy, pwz, act, nobad);
            ObsProj p1 = obs_project<false>(L1, pwx, pwy, pwz, act && has1, nobad);
            const uint32_t draw0 = ldg(L0.depth + p0.npix);              // :193 (no d>0 test)
            const uint32_t draw1 = ldg(L1.depth + p1.npix);
            const double x0 = ldg(L0.xtab + p0.ix), y0 = ldg(L0.ytab + p0.iy);
            const double x1 = ldg(L1.xtab + p1.ix), y1 = ldg(L1.ytab + p1.iy);
            const double *e0 = L0.ctab + 4 * (size_t)draw0;
            const double *e1 = L1.ctab + 4 * (size_t)draw1;
            const double2 e0a = ldg2(e0), e0b = ldg2(e0 + 2);
            const double2 e1a = ldg2(e1), e1b = ldg2(e1 + 2);
            bool kept0, close0, kept1, close1;
            obs_finish<false>(L0, P, p0, draw0, x0, y0, e0a, e0b, pwx, pwy, pwz, s_obs + (size_t)k0 * 256 + lane,
                       s_hit + k0 * 64 + lane, kept0, close0, nobad);
            obs_finish<false>(L1, P, p1, draw1, x1, y1, e1a, e1b, pwx, pwy, pwz,
                       s_obs + (size_t)(has1 ? k1 : k0) * 256 + lane, s_hit + (has1 ? k1 : k0) * 64 + lane,
                       kept1, close1, nobad);
            const unsigned long long mk0 = __ballot(kept0), mc0 = __ballot(close0);
            const unsigned long long mk1 = __ballot(kept1), mc1 = __ballot(close1);
            c_gather += (unsigned)__popcll(__ballot(p0.vis)) + (unsigned)__popcll(__ballot(p1.vis));
            if (lane == 0) {
                s_kept[k0] = mk0;
                s_close[k0] = mc0;
                if (has1) {
                    s_kept[k1] = mk1;
                    s_close[k1] = mc1;
                }
            }
        }
        __syncthreads();
        // every wave: the decision (:218, :229) from the ballots ...
        unsigned n_obs = 0, n_close = 0;
        for (uint32_t k = 0; k < K; k++) {
            n_obs += (unsigned)((s_kept[k] >> lane) & 1ull);
            n_close += (unsigned)((s_close[k] >> lane) & 1ull);
        }
        bool ok = act && n_obs >= 1;                                         // :218
        if ((double)n_close < (double)n_obs * P.close_frac) ok = false;      // :229
        // ... and the mask scatter of its own neighbours (:152-154); masks of already-processed
        // keyframes are never read again
        for (uint32_t k = w; k < K; k += NW) {
            const bool hitk = ok && ((s_kept[k] >> lane) & 1ull);
            c_hit += (unsigned)__popcll(__ballot(hitk));
            if (hitk && nbp[k].order > F.order)
                *(__attribute__((address_space(1))) uint8_t *)(nbp[k].mask + s_hit[k * 64 + lane]) = 255;
        }
        // one wave: weighted fusion in neighbour order (:157-165), voxel key, record
        if (w == p3) {
            // the sums run in neighbour order; the LDS reads do not depend on them and are issued
            // kFuseBatch neighbours at a time (a slot without a kept observation holds stale bits that
            // the select below never lets into the sums)
            double wx = 0, wy = 0, wz = 0, total = 0;
            constexpr int kFuseBatch = 2;
            for (uint32_t kb = 0; kb < K; kb += kFuseBatch) {
                double ox[kFuseBatch], oy[kFuseBatch], oz[kFuseBatch], ow[kFuseBatch];
                bool on[kFuseBatch];
#pragma unroll
                for (int i = 0; i < kFuseBatch; i++) {
                    const uint32_t k = min(kb + i, K - 1);
                    const double *o = s_obs + (size_t)k * 256 + lane;
                    ox[i] = o[0]; oy[i] = o[64]; oz[i] = o[128]; ow[i] = o[192];
                    on[i] = (kb + i < K) && ((s_kept[k] >> lane) & 1ull);
                }
#pragma unroll
                for (int i = 0; i < kFuseBatch; i++) {
                    wx = on[i] ? wx + ox[i] : wx;
                    wy = on[i] ? wy + oy[i] : wy;
                    wz = on[i] ? wz + oz[i] : wz;
                    total = on[i] ? total + ow[i] : total;
                }
            }
            unsigned long long key = 0, tag = kEmpty;
            double fx = 0, fy = 0, fz = 0;
            if (ok) {
                fx = x86_nan(wx / total); fy = x86_nan(wy / total); fz = x86_nan(wz / total);           // :164
                if (pack_voxel_key(P, fx, fy, fz, key)) {
                    tag = ((unsigned long long)F.order << 32) | pix;
                } else {
                    atomicOr(&a.stat[blockIdx.x % kStatShards].err, kErrKeyRange);
                }
            }
            base_r = __shfl(base_r, 0);
            if (act) {
                const unsigned long long rel = base_r + lane;
                if (rel < a.list.shard_cap) {
                    const unsigned long long idx = (unsigned long long)(c % kListShards) * a.list.shard_cap + rel;
                    a.list.tag[idx] = tag;
                    if (tag != kEmpty) {
                        a.list.key[idx] = key;
                        a.list.xyz[3 * idx + 0] = fx;
                        a.list.xyz[3 * idx + 1] = fy;
                        a.list.xyz[3 * idx + 2] = fz;
                    }
                } else {
                    atomicOr(&a.stat[blockIdx.x % kStatShards].err, kErrListFull);
                }
            }
        }
        if (c + gridDim.x < total_chunks) __syncthreads();  // LDS is reused by the next chunk
    }
    if (lane == 0 && (c_gather | c_hit)) {
        StatShard *st = &a.stat[(blockIdx.x * NW + w) % kStatShards];
        if (c_gather) atomicAdd(&st->n_gather, (unsigned long long)c_gather);
        if (c_hit) atomicAdd(&st->n_hit, (unsigned long long)c_hit);
    }
}

// ---- the keyframe chain as one persistent launch ---------------------------------------------
//
// fuse_chain_kernel runs a whole group of keyframes in ONE launch: one 16-wave workgroup per CU,
// all co-resident (cooperative launch), a device-wide barrier between keyframes instead of a
// kernel boundary.  Per keyframe and workgroup:
//   A  the workgroup's share of the image (64-pixel segments b, b+G, b+2G, ...: spread over the
//      whole image so every workgroup sees the average live fraction): live test (:100-101),
//      covariance gate (:136-140), back-projection (:143-144); survivors compacted into LDS;
//   B  (chunk of 64 survivors, neighbour) items dealt round-robin to the 16 waves (:187-216);
//   C  one wave per chunk: decision (:218-231), mask scatter (:152-154), fusion (:157-165),
//      voxel key, record.
// The only data that crosses workgroups inside a launch are the visibility masks.  They are
// written and read with agent-scope relaxed atomics (write-through stores, cache-bypassing loads on
// gfx950) and ordered by "wait until my stores are acknowledged, then arrive at the barrier":
// no L2 write-back/invalidate is needed, which is what makes the barrier (about 2 us for 256
// workgroups, tools/barrier_bench.hip) cheaper than a kernel boundary.  Every spin is bounded:
// a workgroup that waits longer than the launch's timeout raises the abort flags and all leave.
//
// Why relaxed atomics suffice here (no release/acquire pair).  In the AMDGPU memory model for gfx942/gfx950
// (LLVM AMDGPUUsage, "Memory Model gfx942") an agent-scope release is `buffer_wbl2 sc1; s_waitcnt vmcnt(0)`
// and an agent-scope acquire is `s_waitcnt vmcnt(0); buffer_inv sc1`: the write-back and the invalidate exist
// for bytes that PLAIN stores left dirty in an XCD's L2 and that PLAIN loads may find stale in a CU's L1.
// Agent-scope atomic stores and loads are `global_store/load ... sc1`: the store is written through past the
// XCD's L2 and its vmcnt slot is released when the write is acknowledged; the load bypasses L1 and the
// non-coherent L2 lines.  Every access to a mask byte inside this kernel is such an atomic, so what is left of
// the release is the wait, and nothing is left of the acquire.  That is MI355X_MICROARCH.md's valid form
// "sc1 loads in place of the acquire" (section "Workgroup dispatch, XCD placement & inter-workgroup
// visibility"), whose conditions hold: (1) every load of a mask byte in this kernel is mask_load_coherent,
// (2) every store is mask_store_coherent, (3) every wave waits `s_waitcnt vmcnt(0)` (inline asm: the compiler
// cannot drop it) after its stores and the workgroup's one arriving lane adds behind a workgroup barrier,
// (4) the signal is an agent-scope atomic add by one lane per workgroup, the last adder raises sc1 flags, the
// consumer polls one flag with sc1 loads and its other waves load behind a workgroup barrier.  The guide's
// measured rows cover 2- to 16-byte accesses; the 1-byte accesses used here are covered by this repository's own
// bit-exact runs (tests/test_gpu_fullsize.py: 153.6 M pixels, every mask byte decides a point's membership).

constexpr int kChainWaves = 16;
constexpr int kChainThreads = kChainWaves * kWave;
#ifndef SFM_FUSE_RESERVED
#define SFM_FUSE_RESERVED 0  // fuse_chain_kernel: the first N waves own no pixels of step A (they finish the chunks); 0: every wave does
#endif
constexpr int kChainPrepWaves = kChainWaves - SFM_FUSE_RESERVED;  // waves that own pixels of step A
constexpr int kChainSegs = 2;                                    // segments per wave and pass
constexpr int kChainSurv = 1024;                                 // survivors per batch (LDS); a pass with more runs in batches
constexpr int kChainMaxItems = 64;                               // (chunk, neighbour) slots in LDS
constexpr int kBarSub = 16;                                      // first-level barrier counters
constexpr unsigned long long kChainTimeoutTicks = 5000000ull;    // 50 ms of the 100-MHz clock
// barrier words, one per 64-B line: sub[kBarSub], top, (a flag slot no longer used), abort, done; then the flags
constexpr int kBarSubSpace = 16;  // first-level counters one 64-B line apart (4 KiB apart measured 1 % slower)
constexpr int kBarTop = kBarSub * kBarSubSpace, kBarFlag = kBarTop + 16, kBarAbort = kBarFlag + 16;
constexpr int kBarDone = kBarAbort + 16;  // workgroups that have left the kernel (the last one snapshots the list cursors)
// The release flag that everybody polls exists kBarFlags times, 4 KiB apart and away from the counters:
// workgroup b polls copy b mod kBarFlags, so that 256 pollers do not queue on the line (and memory
// channel) that the arrivals' atomics need.  8 copies: -1.3 % per C2 run; 2, 4, 16, 32 copies measured no better.
constexpr int kBarFlags = 8;
constexpr int kBarFlagX = (kBarAbort + 32 + 1023) / 1024 * 1024, kBarFlagSpace = 1024;
// one region per launch (keyframe group), all zeroed by one memset per run: no fill between the launches
constexpr int kBarStride = kBarFlagX + kBarFlags * kBarFlagSpace;
constexpr size_t kBarBytes = (size_t)kBarStride * sizeof(unsigned int);

struct ChainArgs {
    uint32_t f0, f1;           // keyframes [f0, f1) of this launch
    uint32_t item_cap;         // (chunk, neighbour) observation slots that fit in LDS
    uint32_t group;            // index of this launch's keyframe group (list.group_end row it writes)
    ListDev list;
    StatShard *stat;
    unsigned int *bar;
    DevParams P;
    unsigned int *abort_all;            // ctx-wide: set by the first barrier timeout of a run, later launches leave at once
    unsigned long long timeout_ticks;   // barrier spin limit in 100-MHz ticks
    unsigned long long *stamps;  // diagnostic instantiation only: [f1 - f0][gridDim.x][16] 100-MHz timestamps of thread 0
};

// Phase timestamps of thread 0 (tools/chain_stamps.py); compiled into the diagnostic instantiation only
// (fuse_chain_kernel_stamps, -DSFM_STAMPS=1 builds): the production kernels carry none of it.
constexpr int kStampSlots = 48;  // per keyframe and workgroup: 16 of thread 0, 16 + 16 per wave
template <bool STAMPS>
__device__ __forceinline__ void chain_stamp(const ChainArgs &a, uint32_t f, int i, unsigned long long v = ~0ull)
{
    if (STAMPS) {
        if (a.stamps && threadIdx.x == 0)
            a.stamps[((size_t)(f - a.f0) * gridDim.x + blockIdx.x) * kStampSlots + i] = v == ~0ull ? __builtin_amdgcn_s_memrealtime() : v;
    }
}
// ... and one per wave (slots 16 + w and 32 + w): when wave w finished a keyframe's work / left step B
template <bool STAMPS>
__device__ __forceinline__ void chain_wave_stamp(const ChainArgs &a, uint32_t f, int base, unsigned w)
{
    if (STAMPS) {
        if (a.stamps && lane_id() == 0)
            a.stamps[((size_t)(f - a.f0) * gridDim.x + blockIdx.x) * kStampSlots + base + w] = __builtin_amdgcn_s_memrealtime();
    }
}
#define SFM_CSTAMP(i) chain_stamp<STAMPS>(a, f, (i))

// Host: may fuse_chain_kernel's unchecked shared-reciprocal divisions be used for this keyframe?
// (1) cov_trace_vals: operands (-u*b), (-v*b) for 1 <= u < W, 1 <= v < H (zero passes) and disparity^2
//     of every non-zero depth, which is monotonic in the depth.
// (2) obs_project, with this keyframe as source or as neighbour: cx_, cy_, cz_ are sums of products of
//     pose entries, (u - cx)/fx, (v - cy)/fy and a depth in [1, 65535].  A floating-point sum that is
//     not zero is at least an ulp of its smallest term, so with every non-zero pose entry in
//     [2^-100, 2^100], fx, fy in [2^-40, 2^40] and |cx|, |cy| zero or in [2^-40, 2^40] the camera
//     coordinates are zero or within [2^-398, 2^284]: inside div_pair's range.
// Anything else (denormal-scale poses, astronomically scaled intrinsics) runs on the kernel pair.
inline bool chain_div_ok(const DevParams &P, const double K[4], int W, int H, const double *Twk, const double *Tkw)
{
    auto mag = [](double x) { return x < 0 ? -x : x; };
    auto in_range = [&](double x, double lo, double hi) { return mag(x) >= lo && mag(x) <= hi; };  // false for NaN
    auto zero_or = [&](double x, double lo, double hi) { return x == 0.0 || in_range(x, lo, hi); };
    const double f = K[0];
    const double d1 = (f * P.base_line) / (1.0 / P.depth_div), d2 = (f * P.base_line) / (65535.0 / P.depth_div);
    bool ok = in_range(d1 * d1, 0x1p-390, 0x1p390) && in_range(d2 * d2, 0x1p-390, 0x1p390) &&
              in_range(P.base_line, 0x1p-390, 0x1p390) &&
              in_range(P.base_line * (double)(W > H ? (W > 1 ? W : 1) : (H > 1 ? H : 1)), 0x1p-390, 0x1p390);
    ok = ok && in_range(K[0], 0x1p-40, 0x1p40) && in_range(K[1], 0x1p-40, 0x1p40) &&
         zero_or(K[2], 0x1p-40, 0x1p40) && zero_or(K[3], 0x1p-40, 0x1p40) && W < (1 << 24) && H < (1 << 24);
    for (int i = 0; ok && i < 12; i++)  // rows 0..2 of both matrices (row-major device copies)
        ok = zero_or(Twk[i], 0x1p-100, 0x1p100) && zero_or(Tkw[i], 0x1p-100, 0x1p100);
    return ok;
}

__host__ __device__ inline size_t chain_lds_bytes(uint32_t item_cap)
{
    return (size_t)kChainSurv * (3 * 8 + 4) + (size_t)item_cap * (4 * 64 * 8 + 64 * 4 + 16) + kMaxK * 8 +
           2 * kChainWaves * kWave * 4 + 64;
}

__device__ __forceinline__ unsigned bar_load(const unsigned int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread 0 of every workgroup, once per keyframe; the caller has waited for its own stores
__device__ __forceinline__ void chain_arrive(unsigned int *bar, unsigned G, unsigned epoch)
{
    const unsigned s = blockIdx.x % kBarSub;
    const unsigned members = G / kBarSub + (s < G % kBarSub ? 1u : 0u);
    const unsigned prev = __hip_atomic_fetch_add(bar + s * kBarSubSpace, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == epoch * members - 1u) {
        const unsigned subs = G < (unsigned)kBarSub ? G : (unsigned)kBarSub;
        const unsigned p2 = __hip_atomic_fetch_add(bar + kBarTop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p2 == epoch * subs - 1u) {
#pragma unroll
            for (int x = 0; x < kBarFlags; x++)
                __hip_atomic_store(bar + kBarFlagX + x * kBarFlagSpace, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// chain_arrive in two halves, for a caller with other work to issue under the first atomic's round trip
__device__ __forceinline__ unsigned chain_arrive_issue(unsigned int *bar)
{
    return __hip_atomic_fetch_add(bar + (blockIdx.x % kBarSub) * kBarSubSpace, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_arrive_finish(unsigned int *bar, unsigned G, unsigned epoch, unsigned prev)
{
    const unsigned s = blockIdx.x % kBarSub;
    const unsigned members = G / kBarSub + (s < G % kBarSub ? 1u : 0u);
    if (prev == epoch * members - 1u) {
        const unsigned subs = G < (unsigned)kBarSub ? G : (unsigned)kBarSub;
        const unsigned p2 = __hip_atomic_fetch_add(bar + kBarTop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p2 == epoch * subs - 1u) {
#pragma unroll
            for (int x = 0; x < kBarFlags; x++)
                __hip_atomic_store(bar + kBarFlagX + x * kBarFlagSpace, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ unsigned chain_progress(const unsigned int *bar, unsigned G)
{
    (void)G;
    return bar_load(bar + kBarFlagX + (blockIdx.x % (unsigned)kBarFlags) * kBarFlagSpace);
}

__device__ __forceinline__ bool chain_wait(unsigned int *bar, unsigned epoch, unsigned G, unsigned long long timeout_ticks,
                                           unsigned int *abort_all)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (chain_progress(bar, G) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 31u) == 0u || timeout_ticks < 64ull) {
            if (bar_load(bar + kBarAbort) != 0u) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                __hip_atomic_store(bar + kBarAbort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the launches still queued behind this one read this word on entry and leave at once
                __hip_atomic_store(abort_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

// The part of step A that does not depend on the visibility masks, for the workgroup's pixels of
// one pass: validity (:100-101 without the mask), covariance gate (:136-140), back-projection
// (:143-144).  It runs for keyframe f+1 while the workgroup waits at the barrier after keyframe f.
// Pixels are dealt in pieces of 16 (chain_piece), so that a run of live pixels is shared by several
// workgroups and every workgroup samples the whole image.
constexpr int kPiece = 16;
struct ChainPre {
    uint32_t pixel[kChainSegs];
    bool cand[kChainSegs];                 // in the image, depth > 0, gate passed
    double pwx[kChainSegs], pwy[kChainSegs], pwz[kChainSegs];
};

// Piece `pr` of workgroup b (a bijection between the workgroups and the pieces [G*pr, G*(pr+1))).
// On the full chip (G = 256 = 8 XCDs x 32 CUs; workgroup b runs on XCD b mod 8) the pieces are dealt
// XCD-aware: a round is four groups of eight 128-pixel superpieces, XCD x takes one superpiece of
// each group (rotated by three per group so that no XCD keeps an image column), and the XCD's 32
// workgroups share its 32 pieces (rotated per round).  Every 128-byte line of the keyframe's depth,
// mask and colour, and the lines its pixels project to in the neighbours' depth images, are then
// fetched into ONE XCD's L2 instead of (nearly) all eight: FETCH_SIZE of a C2 launch fell from 197 to
// 87 MiB.  The run time did not follow (8.12 -> 8.19 ms; 64-pixel superpieces 8.58, 256-pixel 8.26):
// the gathers' latency is the same from the local L2 and from the Infinity Cache.
#ifndef SFM_XCD_SP
#define SFM_XCD_SP 8
#endif
template <bool XCD8>
__device__ __forceinline__ uint32_t chain_piece(uint32_t pr, unsigned b, unsigned G)
{
    if (XCD8) {  // the host launches this instantiation with G == 256 only
        constexpr uint32_t kSp = SFM_XCD_SP;            // pieces per superpiece
        constexpr uint32_t kSpShift = kSp == 16 ? 4 : kSp == 8 ? 3 : 2;
        const uint32_t x = b & 7u, jj = ((b >> 3) + 13u * pr) & 31u;
        const uint32_t gl = jj >> kSpShift, q = jj & (kSp - 1u);
        const uint32_t t = (x - 3u * ((32u / kSp) * pr + gl)) & 7u;
        return 256u * pr + gl * (8u * kSp) + t * kSp + q;
    }
    return G * pr + (b + 97u * pr) % G;
}

__device__ __forceinline__ uint32_t chain_slots(const FrameDev &F, unsigned G)
{
    if (F.nbr_count == 0) return 0u;  // a keyframe without neighbours yields nothing (:180)
    const uint32_t npx = (uint32_t)F.W * (uint32_t)F.H;
    const uint32_t npieces = (npx + kPiece - 1) / kPiece;
    const uint32_t piece_rounds = (npieces + G - 1) / G;
    return (piece_rounds + (kWave / kPiece) - 1) / (kWave / kPiece);  // 64-lane slots per workgroup
}

template <bool XCD8>
__device__ __forceinline__ void chain_prepare(const FrameDev &F, const DevParams &P, uint32_t i0, unsigned w,
                                              unsigned lane, unsigned b, unsigned G, ChainPre &r, unsigned &err)
{
    bool bad = false;
    const uint32_t npx = (uint32_t)F.W * (uint32_t)F.H;
    const uint32_t slots = chain_slots(F, G);
    // slots == 0 (no pixels or no neighbours): every lane reads element 0, which the host keeps
    // readable even for an empty image (build_layout)
    uint32_t dv[kChainSegs];
#pragma unroll
    for (int u = 0; u < kChainSegs; u++) {
        // (SFM_PREP_FLIP: odd segments are dealt from the last wave down, so that a pass's last, partly filled segment
        // row lands on the waves that do not fuse)
#if SFM_FUSE_RESERVED
        // waves 0 .. SFM_FUSE_RESERVED-1 own no pixels (they fuse); a slot beyond `slots` marks "none"
        const uint32_t i = w < (unsigned)SFM_FUSE_RESERVED ? 0xFFFFFF00u : i0 + (uint32_t)u * kChainPrepWaves + (w - (unsigned)SFM_FUSE_RESERVED);
#else
        const uint32_t i = i0 + (uint32_t)u * kChainWaves + ((SFM_PREP_FLIP && (u & 1)) ? (uint32_t)kChainWaves - 1u - w : w);
#endif
        const uint32_t pr = i * (kWave / kPiece) + lane / kPiece;
        const uint32_t piece = chain_piece<XCD8>(pr, b, G);
        r.pixel[u] = piece * kPiece + lane % kPiece;
        r.cand[u] = (i < slots) && (r.pixel[u] < npx);
        dv[u] = ldg(F.depth + (r.cand[u] ? r.pixel[u] : 0u));
    }
    double2 e01[kChainSegs], e23[kChainSegs];
    double xn[kChainSegs], yn[kChainSegs];
    int uu[kChainSegs], vv[kChainSegs];
#pragma unroll
    for (int u = 0; u < kChainSegs; u++) {
        r.cand[u] = r.cand[u] && (dv[u] > 0u);                                // :101
        const double *e = F.ctab + 4 * (size_t)(r.cand[u] ? dv[u] : 1u);
        e01[u] = ldg2(e);
        e23[u] = ldg2(e + 2);
        uu[u] = r.cand[u] ? (int)(r.pixel[u] % (uint32_t)F.W) : 0;
        vv[u] = r.cand[u] ? (int)(r.pixel[u] / (uint32_t)F.W) : 0;
        xn[u] = ldg(F.xtab + uu[u]);                                          // pinhole.cpp:12
        yn[u] = ldg(F.ytab + vv[u]);
    }
#pragma unroll
    for (int u = 0; u < kChainSegs; u++) {
        bool bd = false;
        const double tr = cov_trace_vals<true>(P, uu[u], vv[u], e01[u], e23[u], bd);  // :136
        bad = bad || (bd && r.cand[u]);  // lanes without a pixel compute on table entry 1
        r.cand[u] = r.cand[u] && !(tr > P.T_cov);                             // :138
        const double dsrc = (double)dv[u];
        const double pcx = xn[u] * dsrc, pcy = yn[u] * dsrc, pcz = dsrc;      // pinhole.cpp:15-19
        r.pwx[u] = row_apply(F.Twk, 0, pcx, pcy, pcz);                        // :144
        r.pwy[u] = row_apply(F.Twk, 1, pcx, pcy, pcz);
        r.pwz[u] = row_apply(F.Twk, 2, pcx, pcy, pcz);
    }
    if (bad) err |= kErrDivRange;
}

// Step B of fuse_chain_kernel: the (chunk, neighbour) items of one round.  Item (c, k) uses LDS slot
// c*K + k (c counted within the round).  A wave runs its items as a three-stage software pipeline,
// so that arithmetic of one item covers the two dependent gathers (depth, then covariance table)
// of the next ones:   S1 projection, depth/x/y gathers   S2 table gather   S3 the rest (:193-216).
// The pipeline is fully unrolled (a wave has at most kChainMaxItems / kChainWaves items) -- no loop
// back-edge at which the compiler would have to drain the loads in flight -- and the vector loads of
// S1/S2 are issued whether or not the stage holds an item (element 0 then), so that their number per
// iteration is fixed and every wait can name exactly the load it needs.
// UNI (K divides the 16 waves): wave w keeps ONE neighbour, k = w mod K, for all its items and takes
// the chunks w/K, w/K + 16/K, ...: the neighbour's intrinsics and poses are loaded into SGPRs once
// per round instead of once per item and stage.  Otherwise items w, w+16, ... in slot order.
template <bool UNI>
__device__ __forceinline__ void chain_items(const FrameDev *__restrict__ nbp, const DevParams &P, uint32_t K, uint32_t rc,
                                            uint32_t c0, uint32_t n, unsigned w, unsigned lane, const double *s_pw,
                                            double *s_obs, uint32_t *s_hit, unsigned long long *s_kept,
                                            unsigned int *cnt_now, unsigned &c_gather, unsigned &err)
{
    constexpr uint32_t kMaxIt = kChainMaxItems / kChainWaves;
    bool bad = false;
    const uint32_t items = rc * K;
    const uint32_t g = UNI ? (uint32_t)kChainWaves / K : 0u;
    const uint32_t k_w = UNI ? w % K : 0u, c_w = UNI ? w / K : 0u;
    const uint32_t nit = UNI ? (c_w < rc ? (rc - c_w + g - 1) / g : 0u)
                             : (items > w ? (items - w + kChainWaves - 1) / kChainWaves : 0u);
    ObsProj p1, p2;
    p1.npix = p2.npix = 0; p1.ix = p1.iy = p2.ix = p2.iy = 0; p1.vis = p2.vis = false;
    uint32_t c1 = 0, c2 = 0, k1 = 0, k2 = 0, draw1 = 0, draw2 = 0;  // c: chunk within the round
    bool v1 = false, v2 = false;  // wave-uniform: stage holds an item
    double x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    double2 e2a = make_double2(0, 0), e2b = make_double2(0, 0);
#pragma unroll
    for (uint32_t t = 0; t < kMaxIt + 2; t++) {  // t is a compile-time constant in every copy
        // S2 of item t-1
        double2 e1a = make_double2(0, 0), e1b = make_double2(0, 0);
        if (t >= 1 && t <= kMaxIt) {
            const double *e = nbp[k1].ctab + 4 * (size_t)draw1;
            e1a = ldg2(e);
            e1b = ldg2(e + 2);
        }
        // S1 of item t
        const bool v0 = t < kMaxIt && t < nit;
        uint32_t cc0 = c1, k0 = k1;
        if (v0) {
            if (UNI) {
                cc0 = c_w + t * g;
                k0 = k_w;
            } else {
                const uint32_t it = w + t * kChainWaves;
                cc0 = it / K;
                k0 = it - cc0 * K;
            }
        }
        const FrameDev &L = nbp[k0];
        ObsProj p0;
        p0.npix = 0; p0.ix = p0.iy = 0; p0.vis = false;
        if (v0) {
            const uint32_t si = min((c0 + cc0) * kWave + lane, (uint32_t)kChainSurv - 1);
            bool bd = false;
            p0 = obs_project<true>(L, s_pw[si], s_pw[kChainSurv + si], s_pw[2 * kChainSurv + si], (c0 + cc0) * kWave + lane < n, bd);
            bad = bad || (bd && (c0 + cc0) * kWave + lane < n);
        }
        uint32_t draw0 = 0;
        double x0 = 0, y0 = 0;
        if (t < kMaxIt) {
            draw0 = ldg(L.depth + p0.npix);  // :193 (no d>0 test)
            x0 = ldg(L.xtab + p0.ix);
            y0 = ldg(L.ytab + p0.iy);
        }
        // S3 of item t-2
        if (t >= 2 && v2) {
            const uint32_t it = c2 * K + k2;
            const uint32_t si = min((c0 + c2) * kWave + lane, (uint32_t)kChainSurv - 1);
            bool kept, close;
            bool bd = false;
            obs_finish<true>(nbp[k2], P, p2, draw2, x2, y2, e2a, e2b, s_pw[si], s_pw[kChainSurv + si], s_pw[2 * kChainSurv + si],
                       s_obs + (size_t)it * 256 + lane, s_hit + it * 64 + lane, kept, close, bd);
            bad = bad || (bd && p2.vis);  // an invisible lane's gathers read element 0
            const unsigned long long mk = __ballot(kept);
            c_gather += (unsigned)__popcll(__ballot(p2.vis));
            if (kept) atomicAdd(&cnt_now[c2 * kWave + lane], close ? 0x10001u : 1u);
            if (lane == 0) s_kept[it] = mk;
        }
        p2 = p1; c2 = c1; k2 = k1; v2 = v1; draw2 = draw1; x2 = x1; y2 = y1; e2a = e1a; e2b = e1b;
        p1 = p0; c1 = cc0; k1 = k0; v1 = v0; draw1 = draw0; x1 = x0; y1 = y0;
    }
    if (bad) err |= kErrDivRange;
}

// frames = every keyframe in processing order, nbrpack = FrameDev copies of the filtered neighbours
// (frames[i].nbr_begin ...); neither is written by any kernel.
// Keep this kernel at or below 120 VGPRs (tools/isa_stats.py; __graft_entry__.build() checks it): with 4
// waves per SIMD that leaves 32 of the 512 registers per lane for a wave of the table kernel of the
// previous group.  At 121+ the allocation is 128, the CU is full, the chain workgroups cannot be placed
// before that kernel has drained, and every launch loses its ~200 us.
template <bool XCD8, bool STAMPS>
__device__ __forceinline__ void fuse_chain_body(const FrameDev *__restrict__ frames,
                                                const FrameDev *__restrict__ nbrpack, const ChainArgs &a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const unsigned lane = lane_id();
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned b = blockIdx.x, G = gridDim.x;
    const DevParams &P = a.P;
    double *s_pw = reinterpret_cast<double *>(smem);                         // [3][kChainSurv]
    double *s_obs = s_pw + 3 * kChainSurv;                                   // [item_cap][4][64]
    uint32_t *s_spix = reinterpret_cast<uint32_t *>(s_obs + (size_t)a.item_cap * 256);  // [kChainSurv]
    uint32_t *s_hit = s_spix + kChainSurv;                                   // [item_cap][64]
    unsigned long long *s_kept = reinterpret_cast<unsigned long long *>(s_hit + (size_t)a.item_cap * 64);
    unsigned long long *s_close = s_kept + a.item_cap;
    uint8_t **s_nbmask = reinterpret_cast<uint8_t **>(s_close + a.item_cap);  // [kMaxK] masks of later neighbours
    // per chunk and lane: kept observations | close observations << 16, summed by LDS atomics while
    // the items run; two buffers, a round clears the one the next round uses
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_nbmask + kMaxK);   // [2][kChainWaves][64]
    unsigned int *s_misc = s_cnt + 2 * kChainWaves * kWave;                     // [0] survivors, [1] go
    unsigned c_gather = 0, c_hit = 0, err = 0;
    // an earlier launch of this run timed out at a barrier (the run is discarded and redone on the kernel pair): every
    // workgroup of every later launch sees the word set -- it was written before this launch began -- and leaves
    // (it still takes part in the list-boundary snapshot below, so that the group's table kernel walks an empty slice)
    const bool aborted = bar_load(a.abort_all) != 0u;
    if (aborted) err |= kErrBarrier;
    for (unsigned i = threadIdx.x; i < 2u * kChainWaves * kWave; i += kChainThreads) s_cnt[i] = 0;
    unsigned round_no = 0;

    ChainPre pre;
    unsigned pf_v = 0;
    chain_prepare<XCD8>(frames[a.f0], P, 0, w, lane, b, G, pre, err);
    for (uint32_t f = a.f0; f < a.f1 && !aborted; f++) {
        SFM_CSTAMP(0);
        chain_wave_stamp<STAMPS>(a, f, 16, w);  // this wave is ready for keyframe f (its fusion and preparation are done)
        if (f > a.f0) {
            // every workgroup arrived after keyframe f-1 (below); its mask stores are out
            if (threadIdx.x == SFM_WAIT_THREAD) s_misc[1] = chain_wait(a.bar, f - a.f0, G, a.timeout_ticks, a.abort_all) ? 1u : 0u;
            __syncthreads();
            if (s_misc[1] == 0u) {  // workgroup-uniform
                err |= kErrBarrier;
                break;
            }
        }
        SFM_CSTAMP(1);
        asm volatile("" ::"v"(pf_v));  // the prefetch issued before the barrier is only retired here
        bool arrived = false;  // workgroup-uniform: this keyframe's barrier arrival is done
        const FrameDev &F = frames[f];
        const uint32_t K = F.nbr_count;
        const FrameDev *nbp = nbrpack + F.nbr_begin;
        const uint32_t slots = chain_slots(F, G);
        // masks are only ever read of keyframes still to come: the scatter skips the others
        if (threadIdx.x < K)
            s_nbmask[threadIdx.x] = nbp[threadIdx.x].order > F.order ? nbp[threadIdx.x].mask : nullptr;
        for (uint32_t i0 = 0; i0 < slots; i0 += kChainPrepWaves * kChainSegs) {
            // A pass is tried whole (part 0); when more than kChainSurv of its pixels survive (only
            // the first keyframes of a sequence), it is redone one segment per wave at a time
            // (parts 1..kChainSegs), which always fits.
            for (uint32_t part = 0; part <= (uint32_t)kChainSegs; part++) {
            __syncthreads();
            if (threadIdx.x == 0) s_misc[0] = 0;
            __syncthreads();
            // ---- A: the mask-independent part was prepared ahead for the first pass
            if (i0 || part) chain_prepare<XCD8>(F, P, i0, w, lane, b, G, pre, err);
            uint32_t pos[kChainSegs], mv[kChainSegs];
#pragma unroll
            for (int u = 0; u < kChainSegs; u++) {
                // Unconditional: every lane with a pixel in the image asks for its own byte, candidate or not (a branch
                // around the load costs 1.7 % of a C2 run; one default address for all idle lanes would make a single
                // mask byte the target of every workgroup, and these loads go past the L2 to the memory side).
                mv[u] = mask_load_coherent(F.mask + (pre.pixel[u] < (uint32_t)F.W * (uint32_t)F.H ? pre.pixel[u] : 0u));
            }
#pragma unroll
            for (int u = 0; u < kChainSegs; u++) {
                const bool pass = pre.cand[u] && (mv[u] == 0u) && (part == 0 || part == (uint32_t)u + 1);  // :100
                const unsigned long long m = __ballot(pass);
                unsigned base = 0;
                if (m != 0ull) {
                    if (lane == 0) base = atomicAdd(&s_misc[0], (unsigned)__popcll(m));
                    base = __builtin_amdgcn_readfirstlane(base);
                }
                pos[u] = pass ? base + (unsigned)__popcll(m & lanemask_lt()) : 0xFFFFFFFFu;
            }
            __syncthreads();
            SFM_CSTAMP(2);
            const uint32_t n = s_misc[0];
            chain_stamp<STAMPS>(a, f, 7, n);
            if (n > (uint32_t)kChainSurv) continue;  // part 0 only: redo in parts (workgroup-uniform)
            // list space for all survivors of this pass: one atomic per workgroup, issued now, its
            // result parked in LDS after step B and used at the end of step C -- nobody waits for it
            unsigned long long res_base = 0;
            if (threadIdx.x == 0 && n) res_base = atomicAdd(&a.list.cursor[(b % kListShards) * 8], (unsigned long long)n);
            {
#pragma unroll
                for (int u = 0; u < kChainSegs; u++) {
                    if (pos[u] != 0xFFFFFFFFu) {
                        s_spix[pos[u]] = pre.pixel[u];
                        s_pw[pos[u]] = pre.pwx[u];
                        s_pw[kChainSurv + pos[u]] = pre.pwy[u];
                        s_pw[2 * kChainSurv + pos[u]] = pre.pwz[u];
                    }
                }
                __syncthreads();
                const uint32_t chunks = (n + kWave - 1) / kWave;
                const uint32_t R = min(min((uint32_t)kChainWaves, a.item_cap / K), chunks);  // chunks per round (host: item_cap >= K)
                for (uint32_t c0 = 0; c0 < chunks; c0 += R) {
                    const uint32_t rc = min(R, chunks - c0);
                    const uint32_t items = rc * K;
                    // The round's chunks are finished (step C's fusion) by the LAST waves, chunk c0+j by wave 15-j: wave 0
                    // waits for the barrier arrival's atomic to return, and the first waves have the second segment of
                    // the next keyframe's preparation and (K dividing 16) one item more in step B -- the workgroup is
                    // as late for the next keyframe as its busiest wave
#if SFM_FIN_ROT
                    const unsigned wf = (w + (unsigned)kChainWaves - SFM_FIN_ROT) % (unsigned)kChainWaves;
#else
                    const unsigned wf = SFM_FIN_TOP ? (unsigned)kChainWaves - 1u - w : w;
#endif
                    const bool fin = wf < rc;
                    const uint32_t cf = c0 + wf;
                    const uint32_t n_in = fin ? min(n - cf * kWave, (uint32_t)kWave) : 0u;
                    const bool actf = lane < n_in;
                    unsigned int *cnt_now = s_cnt + (round_no & 1u) * (kChainWaves * kWave);
                    unsigned int *cnt_next = s_cnt + ((round_no + 1u) & 1u) * (kChainWaves * kWave);
                    round_no++;
                    // ---- B: the (chunk, neighbour) items of this round (:187-216)
                    if (kChainWaves % K == 0)
                        chain_items<true>(nbp, P, K, rc, c0, n, w, lane, s_pw, s_obs, s_hit, s_kept, cnt_now, c_gather, err);
                    else
                        chain_items<false>(nbp, P, K, rc, c0, n, w, lane, s_pw, s_obs, s_hit, s_kept, cnt_now, c_gather, err);
                    if (threadIdx.x == 0 && c0 == 0) *reinterpret_cast<unsigned long long *>(s_misc + 2) = res_base;
                    if (c0 == 0) chain_wave_stamp<STAMPS>(a, f, 32, w);  // this wave's items of the first round are done
                    __syncthreads();
                    SFM_CSTAMP(3);
                    // ---- C: every wave scatters the masks of its own items (:152-154) ...
                    for (uint32_t it = w; it < items; it += kChainWaves) {
                        const uint32_t c = it / K, k = it % K;
                        const unsigned pc = cnt_now[c * kWave + lane];
                        const unsigned long long mk = s_kept[it];
                        uint8_t *mp = s_nbmask[k];
                        const uint32_t hp = s_hit[it * 64 + lane];
                        const unsigned n_obs = pc & 0xFFFFu, n_close = pc >> 16;
                        bool okc = n_obs >= 1;                                               // :218
                        if ((double)n_close < (double)n_obs * P.close_frac) okc = false;     // :229
                        const bool hitk = okc && ((mk >> lane) & 1ull);
                        c_hit += (unsigned)__popcll(__ballot(hitk));
                        if (hitk && mp) mask_store_coherent(mp + hp);
                    }
                    cnt_next[w * kWave + lane] = 0;  // all 16 chunk rows of the next round's buffer
                    // The barrier only orders the mask stores.  After the last scatter of the keyframe the
                    // workgroup arrives at once, and forms its fused points while the others arrive:
                    // the fusion below is off the keyframe chain's critical path.
                    if (f + 1 < a.f1 && c0 + R >= chunks && i0 + kChainPrepWaves * kChainSegs >= slots &&
                        (part == 0 || part == (uint32_t)kChainSegs)) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // my mask stores are acknowledged (write-through)
                        __syncthreads();
                        if (threadIdx.x == SFM_ARRIVE_THREAD) chain_arrive(a.bar, G, f - a.f0 + 1);
                        arrived = true;
                        SFM_CSTAMP(6);
                    }
                    // ... and one wave per chunk forms the fused point and its record
                    if (fin) {
                        // (the colour, :170, is read by finalize_kernel for the voxels' winners only)
                        const uint32_t pixf = s_spix[min(cf * kWave + lane, (uint32_t)kChainSurv - 1)];
                        const uint32_t it0 = wf * K;
                        const unsigned pc = cnt_now[wf * kWave + lane];
                        const unsigned n_obs = pc & 0xFFFFu, n_close = pc >> 16;
                        bool ok = actf && n_obs >= 1;                                        // :218
                        if ((double)n_close < (double)n_obs * P.close_frac) ok = false;      // :229
                        SFM_CSTAMP(9);
                        // weighted fusion in neighbour order (:157-165); the LDS reads run ahead of the sums
                        double wx = 0, wy = 0, wz = 0, total = 0;
                        constexpr int kFuseBatch = 2;  // (4: +0.5 % on a C2 run)
                        for (uint32_t kb = 0; kb < K; kb += kFuseBatch) {
                            double ox[kFuseBatch], oy[kFuseBatch], oz[kFuseBatch], ow[kFuseBatch];
                            bool on[kFuseBatch];
#pragma unroll
                            for (int i = 0; i < kFuseBatch; i++) {
                                const uint32_t k = min(kb + i, K - 1);
                                const double *o = s_obs + (size_t)(it0 + k) * 256 + lane;
                                ox[i] = o[0]; oy[i] = o[64]; oz[i] = o[128]; ow[i] = o[192];
                                on[i] = (kb + i < K) && ((s_kept[it0 + k] >> lane) & 1ull);
                            }
#pragma unroll
                            for (int i = 0; i < kFuseBatch; i++) {
                                wx = on[i] ? wx + ox[i] : wx;
                                wy = on[i] ? wy + oy[i] : wy;
                                wz = on[i] ? wz + oz[i] : wz;
                                total = on[i] ? total + ow[i] : total;
                            }
                        }
                        SFM_CSTAMP(10);
                        unsigned long long key = 0, tag = kEmpty;
                        double fx = 0, fy = 0, fz = 0;
#if SFM_DIV_TRIPLE_CHAIN
                        if (__ballot(ok) != 0ull) {  // wave-uniform: the shared-reciprocal triples look at the whole wave's operands
                            const bool fits = fuse_point_and_key<true>(P, wx, wy, wz, total, ok, fx, fy, fz, key);  // :164, :270-274
                            if (ok) {
                                if (fits) tag = ((unsigned long long)F.order << 32) | pixf;
                                else err |= kErrKeyRange;
                            }
                        }
#else
                        if (ok) {
                            fx = x86_nan(wx / total); fy = x86_nan(wy / total); fz = x86_nan(wz / total);               // :164
                            if (pack_voxel_key(P, fx, fy, fz, key)) {
                                tag = ((unsigned long long)F.order << 32) | pixf;
                            } else {
                                err |= kErrKeyRange;
                            }
                        }
#endif
                        SFM_CSTAMP(11);
                        if (actf) {
                            const unsigned long long rel = *reinterpret_cast<unsigned long long *>(s_misc + 2) + cf * kWave + lane;
                            if (rel < a.list.shard_cap) {
                                const unsigned long long idx = (unsigned long long)(b % kListShards) * a.list.shard_cap + rel;
                                a.list.tag[idx] = tag;
                                if (tag != kEmpty) {
                                    a.list.key[idx] = key;
                                    a.list.xyz[3 * idx + 0] = fx;
                                    a.list.xyz[3 * idx + 1] = fy;
                                    a.list.xyz[3 * idx + 2] = fz;
                                }
                            } else {
                                err |= kErrListFull;
                            }
                        }
                        SFM_CSTAMP(12);
                    }
                    // LDS slots are reused by the next round / part / pass.  After the keyframe's last
                    // round nothing writes LDS before the workgroup-wide sync behind the barrier wait:
                    // the waves that have no chunk to finish go straight on to the next keyframe's
                    // preparation while the others fuse.
                    if (!arrived) __syncthreads();
                    SFM_CSTAMP(4);
                }
            }
            if (part == 0) break;  // the whole pass fitted
            }
        }
        SFM_CSTAMP(5);
        if (f + 1 < a.f1) {
            if (!arrived) {  // a keyframe share without survivors: nothing was scattered
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == SFM_ARRIVE_THREAD) chain_arrive(a.bar, G, f - a.f0 + 1);
                SFM_CSTAMP(6);
            }
            // while the others arrive: pull the next keyframe's neighbour descriptions into this
            // XCD's L2 (they are read through the scalar cache right after the barrier) ...
            {
                const FrameDev &Fn = frames[f + 1];
                const char *pf = reinterpret_cast<const char *>(nbrpack + Fn.nbr_begin);
                if (threadIdx.x * 64u < Fn.nbr_count * (uint32_t)sizeof(FrameDev))
                    pf_v = ldg(reinterpret_cast<const unsigned *>(pf + threadIdx.x * 64u));
            }
            // ... and do its mask-independent work
            chain_prepare<XCD8>(frames[f + 1], P, 0, w, lane, b, G, pre, err);
        }
    }
    // The group's list boundaries for its table kernel: every workgroup's reservations have returned
    // (thread 0 used their values), so the workgroup that leaves last sees the final cursors.  (A
    // 1-KiB copy behind every launch cost a dispatch gap of ~10 us on the chain's stream.)
    if (threadIdx.x == 0) {
        const unsigned left = __hip_atomic_fetch_add(a.bar + kBarDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == G - 1u) {
            unsigned long long *ge = a.list.group_end + (size_t)a.group * kListShards * 8;
            for (int sh = 0; sh < kListShards; sh++)
                ge[sh * 8] = __hip_atomic_load(a.list.cursor + sh * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned any_err = (unsigned)__builtin_amdgcn_readfirstlane((int)(__ballot(err != 0) != 0ull));
    if (any_err) {
        for (int o = 32; o > 0; o >>= 1) err |= __shfl_xor(err, o);
    }
    if (lane == 0 && (c_gather | c_hit | err)) {
        StatShard *st = &a.stat[(b * kChainWaves + w) % kStatShards];
        if (c_gather) atomicAdd(&st->n_gather, (unsigned long long)c_gather);
        if (c_hit) atomicAdd(&st->n_hit, (unsigned long long)c_hit);
        if (err) atomicOr(&st->err, err);
    }
}

// The two instantiations: fuse_chain_kernel for the full chip (256 workgroups, XCD-aware pieces),
// fuse_chain_kernel_any for any other grid.
__global__ __launch_bounds__(kChainThreads) void fuse_chain_kernel(const FrameDev *__restrict__ frames,
                                                                   const FrameDev *__restrict__ nbrpack, ChainArgs a)
{
    fuse_chain_body<true, false>(frames, nbrpack, a);
}

__global__ __launch_bounds__(kChainThreads) void fuse_chain_kernel_any(const FrameDev *__restrict__ frames,
                                                                       const FrameDev *__restrict__ nbrpack, ChainArgs a)
{
    fuse_chain_body<false, false>(frames, nbrpack, a);
}

#if SFM_STAMPS
__global__ __launch_bounds__(kChainThreads) void fuse_chain_kernel_stamps(const FrameDev *__restrict__ frames,
                                                                          const FrameDev *__restrict__ nbrpack, ChainArgs a)
{
    fuse_chain_body<true, true>(frames, nbrpack, a);
}
#endif

// ---- the keyframe chain in two phases (one sequence over several GPUs) -------------------------------------
//
// Everything process_pixel computes except its first test, `mask == 0` (:100), is independent of the visibility
// masks: the covariance gate, the re-observations in the neighbours, the geometric verdict (:218-231), the pixels
// it would mark (:152-154) and the fused point (:157-165).  So the work splits exactly:
//   verdict_kernel        (phase A) for EVERY gate-passing pixel of a keyframe, whatever its mask: verdict, hit
//                         pixels in the later neighbours, fused point and key.  No ordering between keyframes:
//                         a throughput kernel that a rank runs on its block while the ranks before it still chain.
//   accept_chain_kernel   (phase B) the chain proper, keyframe after keyframe behind the same device-wide barrier
//                         as fuse_chain_kernel: read mask and verdict, accept, scatter the stored hits, emit the
//                         record.  A few memory round trips per keyframe, no arithmetic.
// Phase A does the re-observations of every pixel, about four times what fuse_chain_kernel does for the pixels the
// masks leave: on one GPU the fused kernel is faster.  With N GPUs phase A divides by N and phase B does not.

struct VerdictRec {  // 32 bytes per pixel, valid where the verdict is positive
    unsigned long long key;  // packed voxel key of the fused point
    double x, y, z;          // the fused point (:164), x86 NaN bits included
};

constexpr uint32_t kVOk = 0x80000000u;        // geometric_check passed: the pixel yields a point if its mask is clear
constexpr uint32_t kVKeyRange = 0x40000000u;  // ... but its voxel index does not fit the packed key
constexpr uint32_t kVCand = 0x20000000u;      // depth > 0 and the covariance gate passed: the pixel reaches geometric_check
// info = flags | kept observations << 12 | visible projections  (12 bits each: any number of neighbours up to 4095;
// this path has no per-keyframe neighbour limit in LDS or registers -- a pixel's neighbours are a loop)

struct VerdictDev {  // one per keyframe of the active range
    uint32_t *info;        // [W*H]
    VerdictRec *rec;       // [W*H]
    uint32_t *hits;        // [n_later][W*H] pixel marked in the s-th LATER neighbour, kNoHit where nothing is kept
    uint32_t n_later;      // later neighbours of the keyframe ...
    uint32_t later_begin;  // ... whose masks are later_masks[later_begin ...] (a kernel argument), in neighbour order
};

template <bool SHARED>
__global__ __launch_bounds__(256) void verdict_kernel(const FrameDev *__restrict__ frames,
                                                      const FrameDev *__restrict__ nbrpack,
                                                      const VerdictDev *__restrict__ verdicts, uint32_t f0, uint32_t v0,
                                                      DevParams P, StatShard *stat)
{
    const uint32_t f = f0 + blockIdx.y;
    const FrameDev &F = frames[f];
    const VerdictDev &V = verdicts[v0 + blockIdx.y];
    const uint32_t npx = (uint32_t)F.W * (uint32_t)F.H;
    const uint32_t pix = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x * 256u >= npx) return;  // the grid is sized for the launch's largest keyframe
    const bool in = pix < npx;
    const uint32_t K = F.nbr_count;
    const FrameDev *nbp = nbrpack + F.nbr_begin;
    bool bad = false;
    // validity (:101), covariance gate (:136-140), back-projection (:143-144): chain_prepare's arithmetic
    const uint32_t dv = ldg(F.depth + (in ? pix : 0u));
    bool cand = in && dv > 0u && K > 0u;
    const int u = cand ? (int)(pix % (uint32_t)F.W) : 0, v = cand ? (int)(pix / (uint32_t)F.W) : 0;
    const double *e = F.ctab + 4 * (size_t)(cand ? dv : 1u);
    const double2 e01 = ldg2(e), e23 = ldg2(e + 2);
    const double xn = ldg(F.xtab + u), yn = ldg(F.ytab + v);
    {
        bool bd = false;
        const double tr = cov_trace_vals<SHARED>(P, u, v, e01, e23, bd);
        bad = bad || (bd && cand);
        cand = cand && !(tr > P.T_cov);
    }
    const double dsrc = (double)dv;
    const double pcx = xn * dsrc, pcy = yn * dsrc, pcz = dsrc;
    const double pwx = row_apply(F.Twk, 0, pcx, pcy, pcz);
    const double pwy = row_apply(F.Twk, 1, pcx, pcy, pcz);
    const double pwz = row_apply(F.Twk, 2, pcx, pcy, pcz);
    // geometric_check (:187-216) over the neighbours in order; the sums of :157-165 run along
    double wx = 0, wy = 0, wz = 0, total = 0;
    unsigned n_obs = 0, n_close = 0, n_vis = 0, slot = 0;
    if (__ballot(cand) != 0ull) {
        for (uint32_t k = 0; k < K; k++) {
            const FrameDev &L = nbp[k];
            bool bd = false;
            const ObsProj p = obs_project<SHARED>(L, pwx, pwy, pwz, cand, bd);
            const uint32_t draw = ldg(L.depth + p.npix);                 // :193 (no d>0 test)
            const double x = ldg(L.xtab + p.ix), y = ldg(L.ytab + p.iy);
            const double *en = L.ctab + 4 * (size_t)draw;
            const double2 n01 = ldg2(en), n23 = ldg2(en + 2);
            const ObsVal o = obs_values<SHARED>(L, P, p, draw, x, y, n01, n23, pwx, pwy, pwz, bd);
            bad = bad || (bd && p.vis);
            n_vis += p.vis ? 1u : 0u;
            n_obs += o.kept ? 1u : 0u;
            n_close += o.close ? 1u : 0u;
            wx = o.kept ? wx + o.wx : wx;
            wy = o.kept ? wy + o.wy : wy;
            wz = o.kept ? wz + o.wz : wz;
            total = o.kept ? total + o.inv : total;
            if (L.order > F.order) {  // wave-uniform: masks of earlier keyframes are never read again
                if (cand) V.hits[(size_t)slot * npx + pix] = o.kept ? p.npix : kNoHit;  // read for accepted pixels only
                slot++;
            }
        }
    }
    bool ok = cand && n_obs >= 1;                                        // :218
    if ((double)n_close < (double)n_obs * P.close_frac) ok = false;      // :229
    uint32_t info = cand ? (kVCand | (n_obs << 12) | n_vis) : 0u;
    if (__ballot(ok) != 0ull) {  // wave-uniform
        double fx, fy, fz;
        VerdictRec r;
        const bool fits = fuse_point_and_key<SHARED && SFM_DIV_TRIPLE_VERDICT != 0>(P, wx, wy, wz, total, ok, fx, fy, fz, r.key);  // :164, :270-274
        if (ok) {
            info |= fits ? kVOk : (kVOk | kVKeyRange);
            r.x = fx; r.y = fy; r.z = fz;
            V.rec[pix] = r;
        }
    }
    if (in) V.info[pix] = info;
    if (__ballot(bad) != 0ull && lane_id() == 0) atomicOr(&stat[blockIdx.x % kStatShards].err, kErrDivRange);
}

// accept_chain_kernel: keyframes [f0, f1) of one launch, one 1024-thread workgroup per CU, the barrier, abort and
// list-boundary machinery of fuse_chain_kernel.  A keyframe is dealt in runs of 64 pixels (one wave, coalesced):
// run t goes to workgroup t mod G, so that every workgroup samples the whole image -- accepted pixels cluster where
// the camera sees something new, and the marks of a 1024-pixel tile full of them (4096 one-byte write-through
// stores from one CU) made the keyframe wait for that one workgroup.
//
// Roles (round 3).  Wave 0 is the workgroup's CONTROL wave and touches no pixel: it polls the barrier flag, makes the
// barrier arrival, scans the pass's per-wave counts and reserves the list space.  Waves 1..15 are DATA waves, kAccIter
// runs each per pass.  A wave's vector-memory operations complete in issue order, so whatever a wave has in flight
// stands in front of its next poll or its next acknowledgement wait: in round 2 thread 0's polls queued behind its own
// wave's twenty prefetch loads and record stores, and every wave's wait for the marks' acknowledgement also waited for
// the cold record loads issued in front of it.  Now the control wave's queue holds its polls and two atomics only, and a
// data wave has nothing but its marks (and loads that returned long ago) in flight when it waits; the records of the
// accepted pixels are read and written behind the arrival.  tools/link_variants_bench.hip measures the same link
// with no records at all at 4.1 us per keyframe (C2 shape); round 2's kernel took 9.0.
constexpr int kAccIter = 4;
constexpr int kAccDataWaves = kChainWaves - 1;

struct AcceptArgs {
    ChainArgs c;                 // f0, f1, group, list, stat, bar, P, abort_all, timeout_ticks (item_cap, stamps unused)
    uint32_t v_first, pad;       // the kernel's `verdicts` argument holds keyframe f at [f - v_first]
};

// frames and verdicts are never written by any kernel while this one runs: as __restrict__ const kernel arguments
// their fields arrive through the scalar cache into SGPRs (passed inside the argument struct, the verdict pointers
// were re-read by vector loads with a full vmcnt(0) wait in front of every prefetch load: 6 us per keyframe)
template <bool STAMPS>
__device__ __forceinline__ void accept_chain_body(const FrameDev *__restrict__ frames,
                                                  const VerdictDev *__restrict__ verdicts,
                                                  uint8_t *const *__restrict__ later_masks, const AcceptArgs &aa)
{
    static_assert(kAccIter * kAccDataWaves <= kWave, "one wave scans the pass's per-wave counts");
    __shared__ unsigned int s_wave[kWave], s_off[kWave];
    __shared__ unsigned int s_go;
    __shared__ unsigned long long s_base;
    const ChainArgs &a = aa.c;
    const unsigned lane = lane_id();
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned b = blockIdx.x, G = gridDim.x;
    unsigned c_gather = 0, c_hit = 0, err = 0;
    const bool aborted = bar_load(a.abort_all) != 0u;
    if (aborted) err |= kErrBarrier;
    if (threadIdx.x < kWave) s_wave[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_go = 1u;
    __syncthreads();
    // The two roles run the same sequence of workgroup barriers -- one per keyframe (B1: the device-wide barrier has
    // opened), two per pass (B2: every data wave's marks are acknowledged; B3: the pass's list space is known) -- each in
    // its own code path (the branch is wave-uniform).
    if (w == 0u) {
        // ---- control wave
        for (uint32_t f = a.f0; f < a.f1 && !aborted; f++) {
            const FrameDev &F = frames[f];
            const uint32_t npx = (uint32_t)F.W * (uint32_t)F.H;
            SFM_CSTAMP(0);
            const uint32_t runs = F.nbr_count ? (npx + kWave - 1) / kWave : 0u;
            const uint32_t mine = (runs + G * kAccDataWaves - 1) / (G * kAccDataWaves);
            // every workgroup arrived after keyframe f-1 (below); its marks are out
            if (f > a.f0 && lane == 0) s_go = chain_wait(a.bar, f - a.f0, G, a.timeout_ticks, a.abort_all) ? 1u : 0u;
            __syncthreads();  // B1
            if (s_go == 0u) {
                err |= kErrBarrier;
                break;
            }
            SFM_CSTAMP(1);
            for (uint32_t t0 = 0; t0 < mine || t0 == 0; t0 += kAccIter) {
                __syncthreads();  // B2
                SFM_CSTAMP(2);
                // after the keyframe's last pass the workgroup arrives at once; under the arrival's round trip ...
                const bool arrive = t0 + kAccIter >= mine && f + 1 < a.f1 && lane == 0;
                unsigned prev = 0;
                if (arrive) prev = chain_arrive_issue(a.bar);
                // ... list space for the pass: a scan of the per-wave counts, one atomic per workgroup
                const unsigned cnt = s_wave[lane];
                unsigned inc = cnt;
                for (int o = 1; o < kWave; o <<= 1) {
                    const unsigned t = __shfl_up(inc, o);
                    if ((int)lane >= o) inc += t;
                }
                s_off[lane] = inc - cnt;
                unsigned long long res = 0ull;
                if (lane == kWave - 1 && inc) res = atomicAdd(&a.list.cursor[(b % kListShards) * 8], (unsigned long long)inc);
                if (arrive) chain_arrive_finish(a.bar, G, f - a.f0 + 1, prev);
                SFM_CSTAMP(6);
                if (lane == kWave - 1) s_base = res;
                __syncthreads();  // B3
                SFM_CSTAMP(4);
                if (mine == 0) break;
            }
        }
    } else {
        // ---- data waves
        const unsigned dw = w - 1u;
        for (uint32_t f = a.f0; f < a.f1 && !aborted; f++) {
            const FrameDev &F = frames[f];
            const VerdictDev &V = verdicts[f - aa.v_first];
            uint8_t *const *later_mask = later_masks + V.later_begin;
            const uint32_t npx = (uint32_t)F.W * (uint32_t)F.H;
            const uint32_t runs = F.nbr_count ? (npx + kWave - 1) / kWave : 0u;              // 64-pixel runs of the keyframe
            const uint32_t mine = (runs + G * kAccDataWaves - 1) / (G * kAccDataWaves);     // runs per data wave (same for every workgroup)
            // Neither the verdicts nor the pixels they would mark depend on the masks: the first pass's verdicts and its
            // stored hit pixels in the first four later neighbours are requested before the barrier opens (dense,
            // coalesced loads for every pixel of the tile, accepted or not: four times the bytes, none of the latency;
            // the loads are unconditional -- a lane without a pixel reads pixel 0, a missing neighbour slot the last
            // one -- and nothing looks at their values before the barrier: no wait is placed in front of it)
            uint32_t pixv[kAccIter], infov[kAccIter], h[kAccIter][4];
            bool inv[kAccIter];
            const uint32_t nl = V.n_later;
            const uint32_t ql = nl ? nl - 1u : 0u;
#pragma unroll
            for (int j = 0; j < kAccIter; j++) {
                pixv[j] = (((uint32_t)j * kAccDataWaves + dw) * G + b) * kWave + lane;
                inv[j] = (uint32_t)j < mine && pixv[j] < npx;
                const uint32_t pa = inv[j] ? pixv[j] : 0u;
                infov[j] = mine ? (uint32_t)ldg(V.info + pa) : 0u;  // (uniform conditions: an empty keyframe has nothing to read)
#pragma unroll
                for (int q = 0; q < 4; q++) h[j][q] = (mine && nl) ? ldg(V.hits + (size_t)(SFM_DIAG_ACC_ONEHIT ? 0u : min((uint32_t)q, ql)) * npx + pa) : kNoHit;
            }
            __syncthreads();  // B1
            if (s_go == 0u) {
                err |= kErrBarrier;
                break;
            }
            for (uint32_t t0 = 0; t0 < mine || t0 == 0; t0 += kAccIter) {
                if (t0) {
#pragma unroll
                    for (int j = 0; j < kAccIter; j++) {
                        pixv[j] = (((t0 + (uint32_t)j) * kAccDataWaves + dw) * G + b) * kWave + lane;
                        inv[j] = t0 + (uint32_t)j < mine && pixv[j] < npx;
                        const uint32_t pa = inv[j] ? pixv[j] : 0u;
                        infov[j] = (uint32_t)ldg(V.info + pa);
#pragma unroll
                        for (int q = 0; q < 4; q++) h[j][q] = nl ? ldg(V.hits + (size_t)(SFM_DIAG_ACC_ONEHIT ? 0u : min((uint32_t)q, ql)) * npx + pa) : kNoHit;
                    }
                }
                // mask == 0 (:100): every lane with a pixel asks for its own byte, whatever its verdict -- all the loads
                // of the pass are in flight together, one round trip (a load per branch was one round trip each)
                uint32_t mv[kAccIter];
#pragma unroll
                for (int j = 0; j < kAccIter; j++) mv[j] = mask_load_coherent(F.mask + (inv[j] ? pixv[j] : 0u));
#pragma unroll
                for (int j = 0; j < kAccIter; j++) infov[j] = inv[j] ? infov[j] : 0u;
                bool acc[kAccIter];
                unsigned pos[kAccIter];
#pragma unroll
                for (int j = 0; j < kAccIter; j++) {
                    const bool live = (infov[j] & kVCand) && mv[j] == 0u;
                    acc[j] = live && (infov[j] & kVOk);
                    c_gather += live ? (infov[j] & 0xFFFu) : 0u;
                    c_hit += acc[j] ? ((infov[j] >> 12) & 0xFFFu) : 0u;
                    if (acc[j] && (infov[j] & kVKeyRange)) err |= kErrKeyRange;
                    const unsigned long long m = __ballot(acc[j]);
                    pos[j] = lane_rank(m);
                    if (lane == 0) s_wave[j * kAccDataWaves + dw] = (unsigned)__popcll(m);
                }
                if (t0 == 0) chain_wave_stamp<STAMPS>(a, f, 16, w);  // diagnostic build: this wave's mask bytes are back
                // The marks (:152-154) of the accepted pixels, into the masks of keyframes still to come.  (Only the marks
                // into the NEXT keyframe have to be visible when the barrier opens: the later neighbours of keyframe f are
                // ascending, so slot q is keyframe f+1+q or later and has q more barriers to go.  SFM_ACC_DEFER_FAR_MARKS
                // sends slot 0 now and the others behind the arrival: measured, no gain, see the macro.)
#if SFM_ACC_DEFER_FAR_MARKS
                constexpr int kNearSlots = 1;
#else
                constexpr int kNearSlots = 4;
#endif
#pragma unroll
                for (int j = 0; j < kAccIter; j++)
#pragma unroll
                    for (int q = 0; q < kNearSlots; q++)
                        if (acc[j] && (uint32_t)q < nl && h[j][q] != kNoHit)
                            mask_store_coherent(later_mask[q] + h[j][q]);
#if !SFM_ACC_DEFER_FAR_MARKS
                // ... the others four at a time (unconditional loads: a lane without an accepted pixel reads element 0)
                for (uint32_t s0 = 4; s0 < nl; s0 += 4) {
                    uint32_t h2[kAccIter][4];
#pragma unroll
                    for (int j = 0; j < kAccIter; j++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            h2[j][q] = ldg(V.hits + (size_t)min(s0 + (uint32_t)q, nl - 1u) * npx + (acc[j] ? pixv[j] : 0u));
#pragma unroll
                    for (int j = 0; j < kAccIter; j++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (acc[j] && s0 + (uint32_t)q < nl && h2[j][q] != kNoHit)
                                mask_store_coherent(later_mask[s0 + q] + h2[j][q]);
                }
#endif
                // the accepted pixels' voxel keys (cold, scattered 8-byte reads of the verdict records; unconditional loads:
                // their number is fixed), requested now so that their round trip runs under the marks' acknowledgement and
                // the arrival ...  (the fused point itself stays where verdict_kernel left it: finalize_kernel reads it
                // through the tag, for the voxels' winners only -- one list entry is tag + key, 16 bytes instead of 40)
                unsigned long long rk[kAccIter];
#pragma unroll
                for (int j = 0; j < kAccIter; j++)
                    rk[j] = ldg(&V.rec[(acc[j] && !SFM_DIAG_ACC_NOREC) ? pixv[j] : 0u].key);
                // ... but not in front of it: a wave's vector-memory operations complete in issue order, and the wait
                // below leaves exactly the kAccIter loads just issued outstanding.  The barrier only orders the marks.
                // (Measured alternatives, profiles/r03_ab_accept_chain.txt: these loads behind the arrival 4.97 ms per C2
                // run, every pixel's record read ahead of the barrier 4.65 ms, against 4.14 ms with whole records here.)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                static_assert(kAccIter == 4, "the counted wait names kAccIter loads");
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                if (t0 == 0) chain_wave_stamp<STAMPS>(a, f, 32, w);  // ... its marks are acknowledged
                __syncthreads();  // B2
                __syncthreads();  // B3
                // the accepted pixels' list entries, behind the arrival
                const unsigned long long base = s_base;
#pragma unroll
                for (int j = 0; j < kAccIter; j++) {
                    if (acc[j]) {
                        const unsigned long long rel = base + s_off[j * kAccDataWaves + dw] + pos[j];
                        if (rel < a.list.shard_cap) {
                            const unsigned long long idx = (unsigned long long)(b % kListShards) * a.list.shard_cap + rel;
                            a.list.tag[idx] = (infov[j] & kVKeyRange) ? kEmpty : (((unsigned long long)F.order << 32) | pixv[j]);
                            if (!SFM_DIAG_ACC_NOREC) a.list.key[idx] = rk[j];
                        } else {
                            err |= kErrListFull;
                        }
                    }
                }
#if SFM_ACC_DEFER_FAR_MARKS
                // ... and the marks into the keyframes after the next one go out here, behind the arrival: they are older
                // than every memory operation this wave issues for keyframe f+1, so the counted wait of keyframe f+1 (all
                // but its kAccIter youngest loads) covers them -- they are acknowledged before the barrier in front of
                // keyframe f+2 opens, which is the first that may read them.  (The last keyframe of a launch: the kernel ends.)
#pragma unroll
                for (int j = 0; j < kAccIter; j++)
#pragma unroll
                    for (int q = kNearSlots; q < 4; q++)
                        if (acc[j] && (uint32_t)q < nl && h[j][q] != kNoHit)
                            mask_store_coherent(later_mask[q] + h[j][q]);
                for (uint32_t s0 = 4; s0 < nl; s0 += 4) {  // more than four later neighbours: four at a time
                    uint32_t h2[kAccIter][4];
#pragma unroll
                    for (int j = 0; j < kAccIter; j++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            h2[j][q] = ldg(V.hits + (size_t)min(s0 + (uint32_t)q, nl - 1u) * npx + (acc[j] ? pixv[j] : 0u));
#pragma unroll
                    for (int j = 0; j < kAccIter; j++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (acc[j] && s0 + (uint32_t)q < nl && h2[j][q] != kNoHit)
                                mask_store_coherent(later_mask[s0 + q] + h2[j][q]);
                }
#endif
                // (s_wave is rewritten by the next pass's data waves, s_off and s_base by its scan: both behind syncs
                // that every wave reaches only after it has read this pass's values)
                if (mine == 0) break;
            }
        }
    }
    // list boundaries of the group, as in fuse_chain_kernel (group == ~0: one launch per keyframe, no barrier words
    // of its own; the host snapshots the cursors behind the group's last launch).  The control wave has seen the value
    // of every reservation of this workgroup: they are in the cursors.
    __syncthreads();
    if (threadIdx.x == 0 && a.group != 0xFFFFFFFFu) {
        const unsigned left = __hip_atomic_fetch_add(a.bar + kBarDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == G - 1u) {
            unsigned long long *ge = a.list.group_end + (size_t)a.group * kListShards * 8;
            for (int sh = 0; sh < kListShards; sh++)
                ge[sh * 8] = __hip_atomic_load(a.list.cursor + sh * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    c_gather = wave_sum(c_gather);
    c_hit = wave_sum(c_hit);
    const unsigned any_err = (unsigned)__builtin_amdgcn_readfirstlane((int)(__ballot(err != 0) != 0ull));
    if (any_err) {
        for (int o = 32; o > 0; o >>= 1) err |= __shfl_xor(err, o);
    }
    if (lane == 0 && (c_gather | c_hit | err)) {
        StatShard *st = &a.stat[(b * kChainWaves + w) % kStatShards];
        if (c_gather) atomicAdd(&st->n_gather, (unsigned long long)c_gather);
        if (c_hit) atomicAdd(&st->n_hit, (unsigned long long)c_hit);
        if (err) atomicOr(&st->err, err);
    }
}

__global__ __launch_bounds__(kChainThreads) void accept_chain_kernel(const FrameDev *__restrict__ frames,
                                                                     const VerdictDev *__restrict__ verdicts,
                                                                     uint8_t *const *__restrict__ later_masks, AcceptArgs aa)
{
    accept_chain_body<false>(frames, verdicts, later_masks, aa);
}

#if SFM_STAMPS
__global__ __launch_bounds__(kChainThreads) void accept_chain_kernel_stamps(const FrameDev *__restrict__ frames,
                                                                            const VerdictDev *__restrict__ verdicts,
                                                                            uint8_t *const *__restrict__ later_masks,
                                                                            AcceptArgs aa)
{
    accept_chain_body<true>(frames, verdicts, later_masks, aa);
}
#endif

// ---- voxel table: first-wins filter for a group of keyframes ------------------------------

struct TableArgs {
    Slot *table;
    uint64_t table_mask;       // capacity-1
    ListDev list;
    Counters *ctr;
    unsigned long long *frame_before;  // [n_frames] "Points before voxel filter" (:118)
    unsigned long long *frame_after;   // [n_frames] "Points after voxel filter"  (:120)
    unsigned long long *cand;          // [list capacity] list indices that may still win
    uint32_t f0, f1;                   // keyframes [f0, f1), f1 - f0 <= kGroup (local indices)
    uint32_t group;
};

// For every accepted record of keyframes [f0,f1): claim/find the voxel's slot (atomicCAS on
// the key), lower its tag (atomicMin == first-wins per frame and globally,
// dense_reconstruction.cpp:283-287 applied at :119 and :57), and count the voxels each
// keyframe touches (= size of its per-frame filter output).  Stale plain reads are safe:
// keys never change once set, tags only decrease, mark bits only accumulate within a group.
// The kernel is latency-bound (a few resident waves per CU on purpose, it shares the CUs with the
// keyframe chain), so an entry costs as few dependent round trips as possible: tag and key together,
// the slot's key and tag in one read, the atomicMin and the mark's read in flight together, and
// candidate-list space reserved per wave in chunks of kCandChunk (one reservation per ~25 iterations,
// no workgroup-wide sync in the loop).  A chunk's unused tail is filled with kEmpty; finalize_kernel skips those.
constexpr unsigned kCandChunk = 128;
__global__ __launch_bounds__(256) void table_group_kernel(TableArgs a)
{
    __shared__ unsigned int s_before[kGroup], s_after[kGroup];
    __shared__ unsigned int s_claims, s_maxprobe, s_err;
    if (threadIdx.x < kGroup) { s_before[threadIdx.x] = 0; s_after[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { s_claims = 0; s_maxprobe = 0; s_err = 0; }
    __syncthreads();
    const uint32_t shard = blockIdx.y;
    const unsigned long long begin =
        a.group ? a.list.group_end[(size_t)(a.group - 1) * kListShards * 8 + shard * 8] : 0ull;
    unsigned long long end = a.list.group_end[(size_t)a.group * kListShards * 8 + shard * 8];
    if (end > a.list.shard_cap) end = a.list.shard_cap;
    const unsigned long long base = (unsigned long long)shard * a.list.shard_cap;
#if SFM_TABLE_MARK_CAS
    const unsigned long long want_hi = (unsigned long long)a.group + 1ull;
#else
    const uint32_t inv_hi = ~(a.group + 1u);
#endif
    const unsigned lane = lane_id();
    unsigned long long cbase = 0;  // wave-uniform: this wave's chunk of the candidate list
    unsigned cleft = 0;
    // positions as 32-bit offsets from the group's first entry of the shard (host: shard_cap < 2^32)
    const unsigned long long *tagp = a.list.tag + base + begin, *keyp = a.list.key + base + begin;
    const uint32_t n = end > begin ? (uint32_t)(end - begin) : 0u;
    for (uint32_t r0 = blockIdx.x * blockDim.x; r0 < n; r0 += gridDim.x * blockDim.x) {
        const uint32_t r = r0 + threadIdx.x;
        const uint32_t rc = r < n ? r : n - 1;  // both loads unconditional: one round trip
        const unsigned long long tag_ld = tagp[rc];
        const unsigned long long key = keyp[rc];   // unwritten (but readable) where the tag is empty
        const unsigned long long tag = r < n ? tag_ld : kEmpty;
        bool lower = false;
#if SFM_TABLE_DEDUP
        // Neighbouring list entries are neighbouring accepted pixels of one 64-pixel run, and a 1-cm voxel holds several
        // of them: an entry whose left neighbour in the wave is the same voxel of the same keyframe with a smaller tag
        // can neither win the voxel nor be the keyframe's first entry of it.  It counts as "before" and is done.
        const unsigned long long ptag = __shfl_up(tag, 1), pkey = __shfl_up(key, 1);
        const bool dup = lane != 0u && ptag < tag && pkey == key && (uint32_t)(ptag >> 32) == (uint32_t)(tag >> 32);
#else
        const bool dup = false;
#endif
        if (tag != kEmpty) atomicAdd(&s_before[(uint32_t)(tag >> 32) - a.f0], 1u);
        if (tag != kEmpty && !dup) {
            const uint32_t fr = (uint32_t)(tag >> 32) - a.f0;
            uint64_t h = mix64(key) & a.table_mask;
            unsigned probes = 0;
            bool found = false;
            unsigned long long seen = kEmpty;  // the slot's tag as last seen
#if !SFM_TABLE_MARK_CAS
            uint32_t mark_st = 1u;  // ... and of its (inverted) mark: 2 = the word is this group's, 1 = this keyframe's bit is still up
#endif
            for (;;) {
                const ulonglong2 s01 = *reinterpret_cast<const ulonglong2 *>(a.table + h);  // key and tag in one read
#if !SFM_TABLE_MARK_CAS
                const uint2 mk = *reinterpret_cast<const uint2 *>(&a.table[h].mark);       // ... and the mark with them
                mark_st = (mk.y == inv_hi ? 2u : 0u) | ((mk.x >> fr) & 1u);
#endif
                unsigned long long cur = s01.x;
                seen = s01.y;
                if (cur == kEmpty) {
                    cur = atomicCAS(&a.table[h].key, kEmpty, key);
                    if (cur == kEmpty) { atomicAdd(&s_claims, 1u); cur = key; }
                    seen = kEmpty;  // claimed this instant: the atomicMin below sees whatever the tag is by now
                }
                if (cur == key) { found = true; break; }
                h = (h + 1) & a.table_mask;
                if (++probes > 4096u || probes > a.table_mask) break;
            }
            if (!found) {
                s_err = kErrTableFull;
            } else {
                if (probes) atomicMax(&s_maxprobe, probes);
                // Only an entry that was its voxel's smallest tag at this moment can still win:
                // it goes on the candidate list that finalize_kernel walks.
                unsigned long long prev_tag = seen;
                if (seen > tag) prev_tag = atomicMin(&a.table[h].tag, tag);
                lower = prev_tag > tag;
#if SFM_TABLE_MARK_CAS
                unsigned long long old = a.table[h].mark;  // a fresh slot holds kEmpty: no group has that id
                const unsigned long long bit = 1ull << fr;
                for (;;) {
                    const bool same = (old >> 32) == want_hi;
                    if (same && (old & bit)) break;
                    const unsigned long long neu = same ? (old | bit) : ((want_hi << 32) | bit);
                    const unsigned long long prev = atomicCAS(&a.table[h].mark, old, neu);
                    if (prev == old) { atomicAdd(&s_after[fr], 1u); break; }
                    old = prev;
                }
#else
                // The keyframe's "after" count (:120): the first entry of (voxel, keyframe) counts.  The mark is kept
                // INVERTED, ~((group+1) << 32 | bits of the group's keyframes that touched the voxel): a fresh slot's
                // all-ones is "no group" and a later group's word is SMALLER, so an atomicMin with (this group, no bit
                // down) resets an older group's word and leaves this group's alone -- idempotent, nobody waits for it --
                // and an atomicAnd takes this keyframe's bit down; the value it returns says who was first.  No
                // compare-and-swap, and no atomic at all for an entry whose bit the slot's first read showed down
                // (a stale read only shows an older state: then an atomic too many is issued, never one too few).
                const uint32_t bit = 1u << fr;
                if (!(mark_st & 2u)) {
                    unsigned long long reset = ((unsigned long long)inv_hi << 32) | 0xFFFFFFFFull;
                    asm volatile("" : "+v"(reset));  // formed here: hoisted out of the loop it holds two registers (34 > 32, see build.sh)
                    (void)__hip_atomic_fetch_min(&a.table[h].mark, reset, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mark_st = 3u;
                }
                // (the reset and the AND of one lane are two accesses of one thread to one location: they execute in
                // program order -- coherence per location, which the compiler relies on for every store-then-load of a
                // thread as well)
                bool first = false;
                if (mark_st == 3u) first = ((uint32_t)atomicAnd(&a.table[h].mark, ~(unsigned long long)bit) & bit) != 0u;
                if (first) atomicAdd(&s_after[fr], 1u);
#endif
            }
        }
        // candidate append into the wave's chunk
        const unsigned long long m = __ballot(lower);
        if (m != 0) {
            const unsigned cnt = (unsigned)__popcll(m);
            if (cnt > cleft) {  // wave-uniform; cleft < cnt <= 64
                if (lane < cleft) a.cand[cbase + lane] = kEmpty;
                unsigned long long nb = 0;
                if (lane == 0) nb = atomicAdd(&a.ctr->cand_count, (unsigned long long)kCandChunk);
                cbase = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(nb >> 32)) << 32) |
                        (unsigned)__builtin_amdgcn_readfirstlane((int)nb);
                cleft = kCandChunk;
            }
            if (lower) a.cand[cbase + lane_rank(m)] = base + begin + r;
            cbase += cnt;
            cleft -= cnt;
        }
    }
    for (unsigned q = lane; q < cleft; q += 64) a.cand[cbase + q] = kEmpty;
    __syncthreads();
    if (threadIdx.x < kGroup && a.f0 + threadIdx.x < a.f1) {
        if (s_before[threadIdx.x]) atomicAdd(&a.frame_before[a.f0 + threadIdx.x], (unsigned long long)s_before[threadIdx.x]);
        if (s_after[threadIdx.x]) atomicAdd(&a.frame_after[a.f0 + threadIdx.x], (unsigned long long)s_after[threadIdx.x]);
    }
    if (threadIdx.x == 0) {
        if (s_claims) atomicAdd(&a.ctr->table_used, (unsigned long long)s_claims);
        if (s_maxprobe > a.ctr->max_probe) atomicMax(&a.ctr->max_probe, s_maxprobe);
        if (s_err) atomicOr(&a.ctr->err, s_err);
    }
}

// ---- finalize: gather the winners ---------------------------------------------------------

struct FinalizeArgs {
    const Slot *table;
    uint64_t table_mask;
    ListDev list;
    const unsigned long long *cand;
    // cand_end[g] = number of candidates after group g's table kernel (device copy of the counter,
    // taken in stream order): this launch walks [cand_end[group-1], cand_end[group])
    const unsigned long long *cand_end;
    const FrameDev *frames;  // every keyframe of the ctx (device copy): the winners' colours are read here
    // two-phase chain: accept_chain_kernel leaves tag and key in the list; a winner's fused point is read where
    // verdict_kernel left it (verdicts[keyframe - v_first].rec[pixel]).  nullptr: the point is in list.xyz.
    const struct VerdictDev *verdicts;
    uint32_t v_first;
    uint32_t group;
    uint64_t out_cap;  // output slots; the counter keeps counting beyond (the host grows and reruns)
    Counters *ctr;
    // output (SoA, device)
    double *out_xyz;
    uint32_t *out_bgr;
    unsigned long long *out_key;  // packed
    unsigned long long *out_tag;
};

// A candidate is an output point iff its tag is the minimum recorded for its voxel.  One launch per
// group of keyframes, right behind the group's table kernel: later groups hold later keyframes only,
// whose tags are larger, so a group's winners are final as soon as the group is in the table.
// Output space for a 256-thread block's winners: kFinItems candidates per thread (independent
// probes, but one after the other: more items only pay where the counter is the bottleneck), one global
// atomic per block and tile (one per wave serialises on the counter's address: 25 k atomics at
// ~90 per microsecond for a C2 cloud).  Every thread of the block must call it.
template <int kFinItems>
__device__ __forceinline__ void block_reserve(const bool (&win)[kFinItems], unsigned long long *counter,
                                              unsigned int *s_w, unsigned long long *s_base,
                                              unsigned long long (&off)[kFinItems])
{
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    unsigned run = 0, loc[kFinItems];
#pragma unroll
    for (int j = 0; j < kFinItems; j++) {
        const unsigned long long m = __ballot(win[j]);
        loc[j] = run + (unsigned)__popcll(m & lanemask_lt());
        run += (unsigned)__popcll(m);
    }
    if (lane == 0) s_w[wave] = run;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        *s_base = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long base = *s_base;
    for (unsigned w = 0; w < wave; w++) base += s_w[w];
#pragma unroll
    for (int j = 0; j < kFinItems; j++) off[j] = base + loc[j];
    __syncthreads();  // s_w and s_base are reused by the next tile
}

__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a)
{
    constexpr int kFinItems = 1;  // per group: a few thousand tiles, latency-bound probes
    __shared__ unsigned int s_w[4];
    __shared__ unsigned long long s_base;
    const unsigned long long begin = a.group ? a.cand_end[a.group - 1] : 0ull;
    const unsigned long long n = a.cand_end[a.group];
    const unsigned long long tile_sz = 256ull * kFinItems;
    const unsigned long long tiles = (n - begin + tile_sz - 1) / tile_sz;
    for (unsigned long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        bool win[kFinItems];
        unsigned long long key[kFinItems], tag[kFinItems], e[kFinItems], off[kFinItems];
#pragma unroll
        for (int j = 0; j < kFinItems; j++) {
            const unsigned long long i = begin + tile * tile_sz + (unsigned)j * 256u + threadIdx.x;
            win[j] = false;
            key[j] = tag[j] = e[j] = 0;
            if (i < n) e[j] = a.cand[i];
            if (i < n && e[j] != kEmpty) {  // kEmpty: the unused tail of a wave's chunk
                tag[j] = a.list.tag[e[j]];
                key[j] = a.list.key[e[j]];
                uint64_t h = mix64(key[j]) & a.table_mask;
                for (;;) {
                    const unsigned long long cur = a.table[h].key;
                    if (cur == key[j]) { win[j] = a.table[h].tag == tag[j]; break; }
                    if (cur == kEmpty) break;  // cannot happen: every candidate's key was inserted
                    h = (h + 1) & a.table_mask;
                }
            }
        }
        block_reserve(win, &a.ctr->out_count, s_w, &s_base, off);
#pragma unroll
        for (int j = 0; j < kFinItems; j++) {
            if (!win[j]) continue;
            const unsigned long long o = off[j];
            if (o >= a.out_cap) {
                atomicOr(&a.ctr->err, kErrOutFull);
            } else {
                if (a.verdicts) {  // wave-uniform
                    const VerdictRec *r = a.verdicts[(uint32_t)(tag[j] >> 32) - a.v_first].rec + (uint32_t)tag[j];
                    a.out_xyz[3 * o + 0] = r->x;
                    a.out_xyz[3 * o + 1] = r->y;
                    a.out_xyz[3 * o + 2] = r->z;
                } else {
                    a.out_xyz[3 * o + 0] = a.list.xyz[3 * e[j] + 0];
                    a.out_xyz[3 * o + 1] = a.list.xyz[3 * e[j] + 1];
                    a.out_xyz[3 * o + 2] = a.list.xyz[3 * e[j] + 2];
                }
                // colour of the source pixel (:170, features.cpp:76-80): the tag names keyframe and pixel
                a.out_bgr[o] = ldg_u32_unaligned(a.frames[tag[j] >> 32].bgr + 3ull * (uint32_t)tag[j]) & 0xFFFFFFu;
                a.out_key[o] = key[j];
                a.out_tag[o] = tag[j];
            }
        }
    }
}

// ---- multi-GPU: exchange by voxel-key hash partition, per-owner reduce --------------------
//
// After the local run every GPU holds the winners of its own keyframes.  Winners are bucketed
// by owner = part_of(key); the buckets travel by RCCL all-to-all (host side, sfm_amd/
// dist_fusion.py); each owner min-reduces what it receives in a fresh table and compacts.

struct Record {  // 48 bytes on the wire
    unsigned long long key;  // packed voxel key
    unsigned long long tag;  // (global keyframe order << 32) | pixel
    double x, y, z;
    uint32_t bgr, pad;
};

struct KeyPack {  // the packing as the read-out / exchange kernels need it (by value)
    long long lo[3], hi[3];
    uint32_t b[3];
};

__device__ __forceinline__ void unpack_key_dev(unsigned long long k, const KeyPack &kp, unsigned long long out[3])
{
    out[0] = axis_decode_range((k >> (kp.b[1] + kp.b[2])) & ((1ull << kp.b[0]) - 1ull), kp.lo[0], kp.hi[0]);
    out[1] = axis_decode_range((k >> kp.b[2]) & ((1ull << kp.b[1]) - 1ull), kp.lo[1], kp.hi[1]);
    out[2] = axis_decode_range(k & ((1ull << kp.b[2]) - 1ull), kp.lo[2], kp.hi[2]);
}

// The owner of a voxel: a hash of its three indices, NOT of the packed key -- ranks may pack with different widths
// until the exchange has agreed on common ones, the owner must not depend on that.
__device__ __forceinline__ uint32_t part_of(unsigned long long key, KeyPack kb, uint32_t n_parts)
{
    unsigned long long i[3];
    unpack_key_dev(key, kb, i);
    const unsigned long long h = mix64(i[0] ^ mix64(i[1] ^ mix64(i[2] ^ 0x9E3779B97F4A7C15ull)));
    return (uint32_t)((h >> 17) % n_parts);
}

// sfm_fusion_set_key_range on an existing cloud: the same keys in another packing (a key the new packing does not
// hold raises err; the host then leaves the cloud as it is)
__global__ __launch_bounds__(256) void rekey_kernel(const unsigned long long *out_key, unsigned long long *dst,
                                                    unsigned long long n, KeyPack from, KeyPack to, Counters *ctr)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long v[3], code[3];
        unpack_key_dev(out_key[i], from, v);
        bool ok = true;
        for (int a = 0; a < 3; a++)
            ok = axis_code_range(v[a] == 0x8000000000000000ull ? INT64_MIN : (long long)v[a], to.lo[a], to.hi[a], code[a]) && ok;
        if (!ok) atomicOr(&ctr->err, kErrKeyRange);
        dst[i] = (code[0] << (to.b[1] + to.b[2])) | (code[1] << to.b[2]) | code[2];
    }
}

// The smallest and largest voxel index per axis over a cloud's packed keys (the indefinite index aside): what a rank
// tells the others before they agree on a common packing.  out starts as an empty range (lo = max, hi = min).
__global__ __launch_bounds__(256) void key_range_kernel(const unsigned long long *out_key, unsigned long long n, KeyPack kp,
                                                        KeyRangeDev *out)
{
    long long lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long v[3];
        unpack_key_dev(out_key[i], kp, v);
        for (int a = 0; a < 3; a++) {
            if (v[a] == 0x8000000000000000ull) continue;
            const long long k = (long long)v[a];
            lo[a] = k < lo[a] ? k : lo[a];
            hi[a] = k > hi[a] ? k : hi[a];
        }
    }
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) {
            const long long l2 = __shfl_xor(lo[a], o), h2 = __shfl_xor(hi[a], o);
            lo[a] = l2 < lo[a] ? l2 : lo[a];
            hi[a] = h2 > hi[a] ? h2 : hi[a];
        }
        if (lane_id() == 0 && lo[a] <= hi[a]) {
            __hip_atomic_fetch_min(&out->lo[a], lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(&out->hi[a], hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// counts[p] += number of output points owned by part p (n_parts <= 1024)
__global__ __launch_bounds__(256) void partition_count_kernel(const unsigned long long *out_key,
                                                              unsigned long long n, uint32_t n_parts, KeyPack kb,
                                                              unsigned long long *counts)
{
    __shared__ unsigned int s_c[1024];
    for (uint32_t i = threadIdx.x; i < n_parts; i += blockDim.x) s_c[i] = 0;
    __syncthreads();
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x)
        atomicAdd(&s_c[part_of(out_key[i], kb, n_parts)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_parts; i += blockDim.x)
        if (s_c[i]) atomicAdd(&counts[i], (unsigned long long)s_c[i]);
}

// cursors[p] starts at the bucket's first record index.  Bucket space is reserved per tile of 256
// points and owner (LDS histogram, then one global atomic per tile and owner present): a global
// atomic per point on n_parts addresses would serialise the whole cloud on them.
__global__ __launch_bounds__(256) void partition_scatter_kernel(const unsigned long long *out_key,
                                                                const unsigned long long *out_tag,
                                                                const double *out_xyz, const uint32_t *out_bgr,
                                                                unsigned long long n, uint32_t n_parts, KeyPack kb,
                                                                const uint32_t *global_order,
                                                                unsigned long long *cursors, Record *dst)
{
    __shared__ unsigned int s_c[1024];
    __shared__ unsigned long long s_base[1024];
    const unsigned long long n_tiles = (n + blockDim.x - 1) / blockDim.x;
    for (unsigned long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (uint32_t p = threadIdx.x; p < n_parts; p += blockDim.x) s_c[p] = 0;
        __syncthreads();
        const unsigned long long i = tile * blockDim.x + threadIdx.x;
        const bool valid = i < n;
        unsigned long long key = 0;
        uint32_t part = 0, rank = 0;
        if (valid) {
            key = out_key[i];
            part = part_of(key, kb, n_parts);
            rank = atomicAdd(&s_c[part], 1u);
        }
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < n_parts; p += blockDim.x)
            if (s_c[p]) s_base[p] = atomicAdd(&cursors[p], (unsigned long long)s_c[p]);
        __syncthreads();
        if (valid) {
            Record r;
            r.key = key;
            const unsigned long long t = out_tag[i];  // local keyframe index -> global order
            r.tag = ((unsigned long long)global_order[t >> 32] << 32) | (t & 0xFFFFFFFFull);
            r.x = out_xyz[3 * i + 0];
            r.y = out_xyz[3 * i + 1];
            r.z = out_xyz[3 * i + 2];
            r.bgr = out_bgr[i];
            r.pad = 0;
            dst[s_base[part] + rank] = r;
        }
        __syncthreads();
    }
}

// owner side, pass 1: first-wins over everything received (min tag per key)
__global__ __launch_bounds__(256) void merge_insert_kernel(const Record *rec, unsigned long long n, Slot *table,
                                                           uint64_t table_mask, Counters *ctr)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long key = rec[i].key, tag = rec[i].tag;
        uint64_t h = mix64(key) & table_mask;
        unsigned probes = 0;
        for (;;) {
            unsigned long long cur = table[h].key;
            if (cur == kEmpty) {
                cur = atomicCAS(&table[h].key, kEmpty, key);
                if (cur == kEmpty) cur = key;
            }
            if (cur == key) {
                if (table[h].tag > tag) atomicMin(&table[h].tag, tag);
                break;
            }
            h = (h + 1) & table_mask;
            if (++probes > table_mask) { atomicOr(&ctr->err, kErrTableFull); break; }
        }
    }
}

// owner side, pass 2: a received record whose tag is its voxel's minimum is an output point
__global__ __launch_bounds__(256) void merge_finalize_kernel(const Record *rec, unsigned long long n,
                                                             const Slot *table, uint64_t table_mask, Counters *ctr,
                                                             double *out_xyz, uint32_t *out_bgr,
                                                             unsigned long long *out_key, unsigned long long *out_tag)
{
    constexpr int kFinItems = 4;  // the whole cloud at once: the counter would serialise 25 k wave atomics
    __shared__ unsigned int s_w[4];
    __shared__ unsigned long long s_base;
    const unsigned long long tile_sz = 256ull * kFinItems;
    const unsigned long long tiles = (n + tile_sz - 1) / tile_sz;
    for (unsigned long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        bool win[kFinItems];
        unsigned long long off[kFinItems];
#pragma unroll
        for (int j = 0; j < kFinItems; j++) {
            const unsigned long long i = tile * tile_sz + (unsigned)j * 256u + threadIdx.x;
            win[j] = false;
            if (i < n) {
                const unsigned long long key = rec[i].key, tag = rec[i].tag;
                uint64_t h = mix64(key) & table_mask;
                for (;;) {
                    const unsigned long long cur = table[h].key;
                    if (cur == key) { win[j] = table[h].tag == tag; break; }
                    if (cur == kEmpty) break;
                    h = (h + 1) & table_mask;
                }
            }
        }
        block_reserve(win, &ctr->out_count, s_w, &s_base, off);
#pragma unroll
        for (int j = 0; j < kFinItems; j++) {
            if (!win[j]) continue;
            const Record r = rec[tile * tile_sz + (unsigned)j * 256u + threadIdx.x];
            const unsigned long long o = off[j];
            out_xyz[3 * o + 0] = r.x;
            out_xyz[3 * o + 1] = r.y;
            out_xyz[3 * o + 2] = r.z;
            out_bgr[o] = r.bgr;
            out_key[o] = r.key;
            out_tag[o] = r.tag;
        }
    }
}

// ---- one sequence over several GPUs: the visibility masks an earlier block of keyframes left ----------------

// dst[i] |= src[i]: masks are 0 or 255, several earlier blocks may have marked the same keyframe
__global__ __launch_bounds__(256) void mask_or_kernel(uint8_t *dst, const uint8_t *src, unsigned long long n)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x)
        dst[i] |= src[i];
}

// ---- read-out: the cloud in voxel-key order, in the layout sfm_fusion_read_points hands out ----

__global__ __launch_bounds__(256) void iota_kernel(uint32_t *idx, unsigned long long n)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x)
        idx[i] = (uint32_t)i;
}

// perm: positions in ascending packed-key order, which is the lexicographic order of the three
// unsigned components (non-negative indices, then the indefinite value, then negative ones, per axis)
__global__ __launch_bounds__(256) void cloud_gather_kernel(const uint32_t *perm, unsigned long long n,
                                                           const unsigned long long *out_key,
                                                           const unsigned long long *out_tag, const double *out_xyz,
                                                           const uint32_t *out_bgr, const uint32_t *global_order,
                                                           uint32_t n_global_order, KeyPack kb, double *dst_xyz,
                                                           int32_t *dst_bgr, unsigned long long *dst_key,
                                                           unsigned long long *dst_tag)
{
    for (unsigned long long o = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; o < n;
         o += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t i = perm[o];
        dst_xyz[3 * o + 0] = out_xyz[3ull * i + 0];
        dst_xyz[3 * o + 1] = out_xyz[3ull * i + 1];
        dst_xyz[3 * o + 2] = out_xyz[3ull * i + 2];
        const uint32_t c = out_bgr[i];
        dst_bgr[3 * o + 0] = (int32_t)(c & 255);
        dst_bgr[3 * o + 1] = (int32_t)((c >> 8) & 255);
        dst_bgr[3 * o + 2] = (int32_t)((c >> 16) & 255);
        unsigned long long k3[3];
        unpack_key_dev(out_key[i], kb, k3);
        dst_key[3 * o + 0] = k3[0];
        dst_key[3 * o + 1] = k3[1];
        dst_key[3 * o + 2] = k3[2];
        unsigned long long t = out_tag[i];
        if ((t >> 32) < n_global_order) t = ((unsigned long long)global_order[t >> 32] << 32) | (t & 0xFFFFFFFFull);
        dst_tag[o] = t;
    }
}

}  // namespace sfm
