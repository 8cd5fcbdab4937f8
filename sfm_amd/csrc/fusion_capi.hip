// fusion_capi.hip -- host side of libsfm_fusion.so: the C ABI of include/sfm_fusion.h over
// the gfx950 kernels in fusion_kernels.h.  No CPU fallback exists: every entry point that
// computes needs a HIP device and fails with SFM_E_HIP otherwise.
//
// Stream plan of one run (DESIGN.md "Streams"):
//   stream A (caller's or private): the fills of masks and counters, then one fuse_chain_kernel per group of 32
//            keyframes (or scan_frame_kernel + observe_frame_kernel per keyframe), strictly in order -- the
//            visibility masks make this a chain -- and nothing else between the launches.
//   stream B (private): the table's fill, then table_group_kernel + finalize_kernel per group, each waiting on the
//            event that rides on the group's chain launch; they overlap the chain on stream A.
//   stream A then waits for B; one host synchronisation per run.
//   upload stream (private): host images of sfm_fusion_add_keyframe, through two pinned staging buffers.
#include "../../include/sfm_fusion.h"
#include "fusion_kernels.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

using namespace sfm;

namespace sfm {
// cloud_sort.hip (rocPRIM radix sort; tmp == nullptr: size query)
hipError_t sort_keys_with_index(void *tmp, size_t &tmp_bytes, const unsigned long long *keys_in,
                                unsigned long long *keys_out, const uint32_t *idx_in, uint32_t *idx_out, size_t n,
                                hipStream_t stream);
}  // namespace sfm

// Scratch of sfm_fusion_partition_counts_noalloc: the per-owner counters of partition_count_kernel, in the code
// object itself instead of a hipMalloc'ed buffer.  Every GPU the library runs on holds its own copy (8 KiB) at its own
// address until the process ends; host code never uses the symbol's name as an address, only the per-device pointer
// that hipGetSymbolAddress returned (sfm_fusion_create, with that device current).
constexpr uint32_t kScratchParts = 1024;
static __device__ __align__(256) unsigned long long scratch[kScratchParts];

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU, resolved once per device

struct HostFrame {
    int32_t W = 0, H = 0;
    double K[4];
    double Twk[16];  // column-major as given
    double Tkw[16];
    uint16_t *d_depth = nullptr;
    uint8_t *d_bgr = nullptr;
    std::vector<uint32_t> nbrs;  // raw Map::get_neighbors list
    bool selected = true;        // false: dropped by the selection rule (sfm_fusion_set_covisibility)
    uint32_t n_filtered = 0;     // neighbours left after the :72-89 filter
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    uint64_t n = 0;
};

struct HostCounters {  // pinned snapshot read back after the frame loop
    Counters ctr;
    alignas(64) unsigned long long cursor[kListShards * 8];
    alignas(64) StatShard stat[kStatShards];
    // range-scan run: index range of every fused point, per shard.  Last, and outside the per-run memset: a two-phase
    // run fills it in sfm_fusion_precompute, before sfm_fusion_run's own launches
    alignas(64) KeyRangeDev key_scan[kStatShards];
    // error words of sfm_fusion_precompute's verdict_kernel launches (phase A runs before sfm_fusion_run's memset and
    // performs every division of the two-phase path): zeroed by sfm_fusion_precompute, read with every run's results
    alignas(64) StatShard pa_stat[kStatShards];
};

}  // namespace

struct sfm_fusion_ctx {
    sfm_fusion_params prm;
    int device = 0;
    hipStream_t stream = nullptr;    // stream A
    hipStream_t stream_b = nullptr;  // stream B (table kernels)
    unsigned long long *stamp_ptr = nullptr;  // -DSFM_STAMPS=1 builds: timestamps of one chain launch
    size_t stamp_n = 0;
    bool own_stream = false;
    std::vector<HostFrame> frames;
    bool layout_dirty = true;
    // keyframe images live in a few large device chunks (one hipMalloc per keyframe image cost more
    // than its copy); host images are packed into one of two pinned staging buffers and copied on
    // the upload stream while the caller prepares the next keyframe
    struct Arena {
        std::vector<void *> chunks;
        unsigned char *cur = nullptr;
        size_t left = 0, next_chunk = 8u << 20;
    } arena;
    struct Stage {
        unsigned char *h = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool busy = false;
    } stage[2];
    int stage_i = 0;
    hipStream_t stream_up = nullptr;
    hipEvent_t ev_up = nullptr;
    bool up_pending = false;

    DevBuf<double> d_tables;         // x/y/covariance lookup tables
    DevBuf<FrameDev> d_nbrpack;      // per keyframe: a FrameDev copy of each filtered neighbour
    DevBuf<FrameDev> d_frames;       // every keyframe (fuse_chain_kernel reads them on the device)
    // one allocation, one memset per run: per-keyframe before/after counts, every group's barrier words of
    // fuse_chain_kernel, the kernel pair's queue cursors (a fill per buffer was a dispatch each, ~7 us)
    DevBuf<unsigned long long> d_zero;
    unsigned long long *z_fbefore = nullptr, *z_fafter = nullptr;
    unsigned int *z_bar = nullptr, *z_qcount = nullptr, *z_abort = nullptr;
    size_t z_bytes = 0;
    // keyframe chain: 1 = one persistent fuse_chain_kernel per group of keyframes (default when the
    // device can hold it), 0 = scan_frame_kernel + observe_frame_kernel per keyframe (SFM_FUSE_CHAIN)
    bool chain = true;
    // Plain launch by default: the grid is one workgroup per CU, each taking (nearly) the whole LDS
    // of its CU, so it is co-resident whenever the device is not shared (checked against the occupancy
    // query at create); every barrier spin is bounded, and a timeout sends the run to the kernel pair.
    // SFM_CHAIN_COOP=1 selects hipLaunchCooperativeKernel instead (about 4 % slower per launch here).
    // Known, outside this library: a process that made cooperative launches under `rocprofv3 --pmc` ends with
    // SIGSEGV inside its exit handlers -- libamdhip64's atexit handler -> libhsa-runtime64 shutdown, after the
    // profiler's "tool finalization", with no frame of this library on the stack and every ctx already
    // destroyed (tools/exit_probe.py maps the addresses; results are complete before it).  Plain launches
    // under the same profiler command exit cleanly.
    bool chain_coop = false;
    uint32_t chain_items = 0;        // (chunk, neighbour) LDS slots; 0 = chain kernel unusable here
    size_t chain_lds = 0;
    bool chain_ran = false;
    uint32_t chain_fallbacks = 0;    // runs that a barrier timeout (or a division-range error) sent to the kernel pair
    uint32_t chain_off_runs = 0;     // after a fallback: runs left on the kernel pair before the chain kernel is tried again
    unsigned long long chain_timeout_ticks = kChainTimeoutTicks;  // SFM_CHAIN_TIMEOUT_TICKS (tests force a timeout with it)
    std::vector<FrameDev> h_frames;  // device-side description of every keyframe (host copy)
    DevBuf<uint8_t> d_masks;
    DevBuf<Slot> d_table;
    DevBuf<unsigned long long> d_list_key, d_list_tag, d_out_key, d_out_tag, d_cand;
    DevBuf<double> d_list_xyz, d_out_xyz;
    DevBuf<uint32_t> d_out_bgr;
    DevBuf<unsigned long long> d_group_end, d_cand_end;
    uint64_t out_cap = 0;              // output slots of the last run (grown on kErrOutFull)
    DevBuf<uint32_t> d_qpix;           // survivor queue of the keyframe in flight
    DevBuf<double> d_qpw;
    uint32_t q_shard_cap = 0;
    int n_cu = 256;
    bool chain_xcd = true;           // XCD-aware piece mapping (needs 256 CUs; SFM_CHAIN_XCD=0 selects the generic kernel)
    // one allocation: Counters | cursors | stat shards (zeroed by a single memset per run)
    unsigned char *d_small = nullptr;
    HostCounters *h_small = nullptr;  // pinned
    StatShard *h_chain = nullptr;     // pinned: the chain's stat shards | phase A's, read by sfm_fusion_run_chain
    // a run split in two (sfm_fusion_run_chain, sfm_fusion_finish): the attempt in flight and its capacities
    struct RunState {
        bool launched = false;
        uint64_t table_cap = 0, shard_cap = 0, out_cap = 0;
        int attempt = 0, key_attempts = 0;
    } rs;
    bool have_masks = false;          // the chain of the run in flight has finished: the masks may be exported
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_init = nullptr, ev_b_done = nullptr;
    std::vector<hipEvent_t> gev;  // one per keyframe group
    uint32_t max_waves = 4;          // waves per observe workgroup (SFM_FUSE_WAVES overrides)
    // table_group_kernel workgroups per list shard (SFM_TABLE_WGS overrides).  Deliberately small:
    // the kernel is off the keyframe chain, and a chip-filling grid slows the chain kernels it
    // overlaps by more than it gains (r01v6 sweep: 64 -> 14.7 ms, 16 -> 13.2 ms, 4 -> 13.9 ms per C2 run)
    uint32_t table_wgs = 16;
    uint32_t table_wgs_alone = 32;   // ... and behind the chain (sfm_fusion_run_chain; SFM_TABLE_WGS_ALONE overrides)
    uint32_t table_threads = 256;    // threads per table_group_kernel workgroup (SFM_TABLE_THREADS: 64, 128 or 256)
    uint32_t waves_per_cu = 28;      // resident observe waves per CU (SFM_FUSE_RESIDENT overrides)
    bool profile = false;         // per-launch start/stop events on the frame kernels
    std::vector<hipEvent_t> kev;  // 2 per keyframe when profiling

    std::vector<uint32_t> global_order; // multi-GPU: global processing order of each keyframe
    DevBuf<uint32_t> d_gorder;
    bool merged = false;                // cloud tags already carry global orders
    DevBuf<unsigned long long> d_part; // [2 * n_parts] bucket counts, then scatter cursors
    // sfm_fusion_read_points: sorted keys, permutation (2n), sort scratch, the cloud in the caller's layout
    DevBuf<unsigned long long> d_rd_keys, d_rd_key3, d_rd_tag;
    DevBuf<uint32_t> d_rd_perm;
    DevBuf<unsigned char> d_rd_tmp;
    DevBuf<double> d_rd_xyz;
    DevBuf<int32_t> d_rd_bgr;
    std::vector<uint64_t> part_counts; // last sfm_fusion_partition_counts result
    unsigned long long *scratch = nullptr;   // this device's copy of the static scratch array (never freed)
    unsigned long long *h_part = nullptr;    // pinned, kScratchParts words: read-back of the no-alloc entry point
    uint64_t total_px = 0;             // pixels of the selected keyframes of the active range
    // One sequence over several GPUs (sfm_fusion_set_active_range): only keyframes [act_first, act_end) are fused
    // by this ctx, the others lend images and poses and hold masks.  mask_seed = the visibility masks the run
    // starts from (what the ranks before this one left, sfm_fusion_import_masks) instead of zeros.
    // the exact key packing in use (DevParams::key): voxel indices held per axis, and the code widths that follow
    KeyRangeDev key = {{-1048575, -1048575, -1048575}, {1048575, 1048575, 1048575}};
    uint32_t key_bits[3] = {21, 21, 21};
    bool key_scan = false;             // the attempt at hand is a range-scan run
    uint32_t act_first = 0, act_end = 0xFFFFFFFFu;
    std::vector<uint64_t> mask_off;    // per keyframe: offset of its mask in d_masks
    // d_mask_seed has the arena's layout, but only the keyframes named in seed_frames hold anything (never zeroed as a
    // whole): the first import of a keyframe copies, later ones OR.  masks_ready: the mask arena itself is already what
    // the next run starts from -- zeroed ahead of time by sfm_fusion_clear_imported_masks, the imports ORed in as they
    // came -- so that nothing but the imports stands between a predecessor's masks and this rank's chain.
    DevBuf<uint8_t> d_mask_seed;
    std::vector<uint32_t> seed_frames;
    bool have_seed = false;
    bool masks_ready = false;
    // two-phase chain (sfm_fusion_precompute): the mask-independent verdicts of the active keyframes
    std::vector<uint32_t> h_nbr;       // filtered neighbours of every keyframe, back to back (FrameDev::nbr_begin)
    DevBuf<unsigned char> d_verdict_arena;
    DevBuf<VerdictDev> d_verdicts;
    DevBuf<uint8_t *> d_later_masks;   // masks of every active keyframe's later neighbours (VerdictDev::later_begin)
    bool have_verdicts = false;
    bool verdict_plain = false;        // phase A met a division operand outside the shared range: verdict_kernel divides plainly
    bool big_k = false;                // a keyframe has more filtered neighbours than the fused kernels hold (kMaxK)
    bool auto_precompute = false;      // SFM_PRECOMPUTE=1: every run goes through the two phases (tests)
    bool verdicts_ran = false;
    hipEvent_t ev_pa[2] = {nullptr, nullptr};  // around phase A's launches
    bool pa_timed = false;
    uint64_t n_out = 0;
    bool have_result = false;
    sfm_fusion_stats stats;
    std::vector<uint64_t> fbefore, fafter;
    uint64_t *h_fcounts = nullptr;     // pinned: per-keyframe before | after counts of the last run (2 * h_fcounts_n words)
    size_t h_fcounts_n = 0;
    std::string err;
};

namespace {

constexpr size_t kSmallBytes = sizeof(HostCounters);
constexpr size_t kSmallZeroBytes = offsetof(HostCounters, key_scan);  // what every run zeroes

Counters *dev_ctr(sfm_fusion_ctx *c) { return reinterpret_cast<Counters *>(c->d_small); }
KeyRangeDev *dev_key_scan(sfm_fusion_ctx *c)
{
    return reinterpret_cast<KeyRangeDev *>(c->d_small + offsetof(HostCounters, key_scan));
}
uint32_t bits_for_range(long long lo, long long hi)
{
    const unsigned long long codes = axis_code_count(lo, hi);
    uint32_t b = 1;
    while (b < 64 && (1ull << b) < codes) b++;
    return b;
}
bool key_is_default(const KeyRangeDev &k)
{
    for (int a = 0; a < 3; a++)
        if (k.lo[a] != -1048575 || k.hi[a] != 1048575) return false;
    return true;
}
KeyPack key_pack_of(const KeyRangeDev &k)
{
    KeyPack kp;
    for (int a = 0; a < 3; a++) {
        kp.lo[a] = k.lo[a];
        kp.hi[a] = k.hi[a];
        kp.b[a] = bits_for_range(k.lo[a], k.hi[a]);
    }
    return kp;
}
KeyPack key_bits_of(const sfm_fusion_ctx *c) { return key_pack_of(c->key); }
// 0 when the ranges make an exact 64-bit key, else why not
const char *key_range_problem(const long long lo[3], const long long hi[3])
{
    uint32_t sum = 0;
    for (int a = 0; a < 3; a++) {
        if (lo[a] > hi[a]) return "empty index range";
        if (lo[a] < -(1ll << 62) || hi[a] > (1ll << 62) || (unsigned long long)(hi[a] - lo[a]) >= (1ull << 53))
            return "index range of an axis wider than 2^53";
        sum += bits_for_range(lo[a], hi[a]);
    }
    return sum > 63 ? "the three axes need more than 63 bits in all" : nullptr;
}
unsigned long long *dev_cursor(sfm_fusion_ctx *c)
{
    return reinterpret_cast<unsigned long long *>(c->d_small + offsetof(HostCounters, cursor));
}
StatShard *dev_stat(sfm_fusion_ctx *c)
{
    return reinterpret_cast<StatShard *>(c->d_small + offsetof(HostCounters, stat));
}
StatShard *dev_pa_stat(sfm_fusion_ctx *c)
{
    return reinterpret_cast<StatShard *>(c->d_small + offsetof(HostCounters, pa_stat));
}

int fail(sfm_fusion_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail((c), e_ == hipErrorOutOfMemory ? SFM_E_NOMEM : SFM_E_HIP,            \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

// the calls that change what a run computes are refused between sfm_fusion_run_chain and sfm_fusion_finish
int run_in_flight(sfm_fusion_ctx *c)
{
    return fail(c, SFM_E_INVALID, "a run is in flight: call sfm_fusion_finish first");
}

int arena_alloc(sfm_fusion_ctx *c, size_t bytes, void **out)
{
    bytes = (bytes + 255) & ~(size_t)255;
    auto &a = c->arena;
    if (a.left < bytes) {
        size_t chunk = std::max(bytes, a.next_chunk);
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, chunk);
        if (e != hipSuccess && chunk > bytes) {  // a large chunk may not fit where the image alone does
            (void)hipGetLastError();
            chunk = bytes;
            e = hipMalloc(&p, chunk);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, SFM_E_NOMEM, "hipMalloc keyframe images");
        }
        a.chunks.push_back(p);
        a.cur = (unsigned char *)p;
        a.left = chunk;
        a.next_chunk = std::min<size_t>(a.next_chunk * 2, (size_t)512 << 20);
    }
    *out = a.cur;
    a.cur += bytes;
    a.left -= bytes;
    return SFM_OK;
}

// rows of `row` bytes, `stride` apart -> contiguous
void pack_rows(unsigned char *dst, const void *src, size_t row, size_t stride, int32_t H)
{
    if (stride == row) {
        std::memcpy(dst, src, row * (size_t)H);
        return;
    }
    for (int32_t y = 0; y < H; y++) std::memcpy(dst + row * (size_t)y, (const unsigned char *)src + stride * (size_t)y, row);
}

template <typename T>
int ensure(sfm_fusion_ctx *c, DevBuf<T> &b, uint64_t n)
{
    if (b.n >= n && b.p) return SFM_OK;
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.n = 0;
    HIP_TRY(c, hipMalloc(&b.p, std::max<uint64_t>(n, 1) * sizeof(T)));
    b.n = n;
    return SFM_OK;
}

template <typename T>
void release(DevBuf<T> &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.n = 0;
}

uint64_t pow2_at_least(uint64_t n)
{
    uint64_t c = 1024;
    while (c < n) c <<= 1;
    return c;
}

// smallest s with sqrt(s) >= T  ("(p - p_w).norm() < T_dist", dense_reconstruction.cpp:224,
// evaluated on the squared norm so the kernel needs no sqrt).
double sqrt_threshold(double T)
{
    if (!(T > 0)) return 0.0;  // sqrt(s) >= 0 >= T for every s; NaN T: never "<"
    if (std::isinf(T)) return T;
    double s = T * T;
    while (std::sqrt(s) >= T && s > 0) s = std::nextafter(s, 0.0);
    while (std::sqrt(s) < T) s = std::nextafter(s, std::numeric_limits<double>::infinity());
    return s;
}

// dense_reconstruction.cpp:72-89: keep neighbour lf of frame f?
// norm(): (dx^2 + dy^2) + dz^2; R*(0,0,1): third column; dot: (a0b0 + a1b1) + a2b2.
bool neighbor_kept(const sfm_fusion_params &p, const HostFrame &f, const HostFrame &lf)
{
    const double dx = lf.Twk[12] - f.Twk[12];
    const double dy = lf.Twk[13] - f.Twk[13];
    const double dz = lf.Twk[14] - f.Twk[14];
    const double n = std::sqrt((dx * dx + dy * dy) + dz * dz);
    if (n > p.min_dist_filter) return true;
    const double dot = (f.Twk[8] * lf.Twk[8] + f.Twk[9] * lf.Twk[9]) + f.Twk[10] * lf.Twk[10];
    return dot < p.dir_dot;
}

void rows3(const double cm[16], double out[12])
{
    for (int r = 0; r < 3; r++)
        for (int col = 0; col < 4; col++) out[r * 4 + col] = cm[r + 4 * col];
}

double sqrt_threshold(double T);

int reset_key_scan(sfm_fusion_ctx *c)
{
    static const std::vector<KeyRangeDev> empty = [] {
        std::vector<KeyRangeDev> v(kStatShards);
        for (auto &r : v)
            for (int ax = 0; ax < 3; ax++) {
                r.lo[ax] = std::numeric_limits<long long>::max();
                r.hi[ax] = std::numeric_limits<long long>::min();
            }
        return v;
    }();
    HIP_TRY(c, hipMemcpyAsync(dev_key_scan(c), empty.data(), sizeof(KeyRangeDev) * kStatShards, hipMemcpyHostToDevice, c->stream));
    return SFM_OK;
}

DevParams make_dev_params(sfm_fusion_ctx *c)
{
    DevParams P;
    P.base_line = c->prm.base_line;
    P.T_cov = c->prm.T_cov;
    P.voxel_size = c->prm.voxel_size;
    P.tau_dist2 = sqrt_threshold(c->prm.T_dist);
    P.S0 = c->prm.S[0];
    P.S1 = c->prm.S[1];
    P.S2 = c->prm.S[2];
    P.depth_div = c->prm.depth_div;
    P.close_frac = c->prm.close_frac;
    P.key = c->key;
    for (int a = 0; a < 3; a++) P.key_bits[a] = c->key_bits[a];
    P.key_wide = key_is_default(c->key) ? 0u : 1u;
    P.key_scan = c->key_scan ? 1u : 0u;
    P.key_scan_out = dev_key_scan(c);
    return P;
}

int build_layout(sfm_fusion_ctx *c)
{
    const size_t n = c->frames.size();
    uint64_t total = 0;
    std::vector<uint64_t> &moff = c->mask_off;
    moff.assign(n, 0);
    c->have_seed = false;  // offsets may move: imported masks belong to the layout they were imported into
    c->seed_frames.clear();
    c->masks_ready = false;
    for (size_t i = 0; i < n; i++) {
        moff[i] = total;
        total += ((uint64_t)c->frames[i].W * c->frames[i].H + 255) / 256 * 256;
    }
    c->total_px = 0;
    for (size_t i = c->act_first; i < std::min<size_t>(n, c->act_end); i++)
        if (c->frames[i].selected) c->total_px += (uint64_t)c->frames[i].W * c->frames[i].H;
    int rc;
    if ((rc = ensure(c, c->d_masks, total))) return rc;
    {
        const size_t bar_words = (size_t)kBarStride * std::max<size_t>(1, (n + kGroup - 1) / kGroup);
        const size_t q_words = n * kQueueShards * 16;
        const size_t n2 = (2 * n + 31) / 32 * 32;  // barrier words start on a 256-byte boundary
        const size_t words64 = n2 + (bar_words + 1) / 2 + (q_words + 1) / 2 + 8;
        if ((rc = ensure(c, c->d_zero, words64))) return rc;
        c->z_fbefore = c->d_zero.p;
        c->z_fafter = c->d_zero.p + n;
        c->z_bar = reinterpret_cast<unsigned int *>(c->d_zero.p + n2);
        c->z_qcount = c->z_bar + (bar_words + 1) / 2 * 2;
        c->z_abort = reinterpret_cast<unsigned int *>(c->d_zero.p + words64 - 8);  // the slack words at the end
        c->z_bytes = words64 * sizeof(unsigned long long);
    }
    std::vector<FrameDev> &fd = c->h_frames;
    fd.assign(n, FrameDev());
    c->have_verdicts = false;
    c->verdict_plain = false;
    c->big_k = false;
    if (c->h_fcounts_n < n) {
        if (c->h_fcounts) HIP_TRY(c, hipHostFree(c->h_fcounts));
        c->h_fcounts = nullptr;
        c->h_fcounts_n = 0;
        HIP_TRY(c, hipHostMalloc((void **)&c->h_fcounts, std::max<size_t>(n, 1) * 2 * sizeof(uint64_t), hipHostMallocDefault));
        c->h_fcounts_n = n;
    }
    std::vector<uint32_t> &nbr = c->h_nbr;
    nbr.clear();
    // lookup tables, one per distinct (f, c, extent) / focal length
    struct Axis { double f, c; int n; size_t off; };
    std::vector<Axis> axes;
    std::vector<std::pair<double, size_t>> covs;
    size_t tab_doubles = 0;
    auto axis_of = [&](double f, double cc, int len) {
        for (auto &a : axes)
            if (std::memcmp(&a.f, &f, 8) == 0 && std::memcmp(&a.c, &cc, 8) == 0 && a.n == len) return a.off;
        axes.push_back({f, cc, len, tab_doubles});
        tab_doubles += ((size_t)std::max(len, 1) + 1) / 2 * 2;
        return axes.back().off;
    };
    auto cov_of = [&](double f) {
        for (auto &e : covs)
            if (std::memcmp(&e.first, &f, 8) == 0) return e.second;
        covs.push_back({f, tab_doubles});
        tab_doubles += 4 * 65536;
        return covs.back().second;
    };
    std::vector<size_t> xo(n), yo(n), co(n);
    for (size_t i = 0; i < n; i++) {
        const HostFrame &f = c->frames[i];
        co[i] = cov_of(f.K[0]);
        xo[i] = axis_of(f.K[0], f.K[2], f.W);
        yo[i] = axis_of(f.K[1], f.K[3], f.H);
    }
    if ((rc = ensure(c, c->d_tables, tab_doubles))) return rc;
    {
        const DevParams P = make_dev_params(c);
        for (auto &e : covs)
            hipLaunchKernelGGL(build_cov_table, dim3(256), dim3(256), 0, c->stream, c->d_tables.p + e.second, e.first, P);
        for (auto &a : axes)
            if (a.n > 0)
                hipLaunchKernelGGL(build_axis_table, dim3((a.n + 255) / 256), dim3(256), 0, c->stream,
                                   c->d_tables.p + a.off, a.n, a.c, a.f);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    for (size_t i = 0; i < n; i++) {
        HostFrame &f = c->frames[i];
        FrameDev &d = fd[i];
        std::memcpy(d.K, f.K, sizeof d.K);
        rows3(f.Twk, d.Twk);
        rows3(f.Tkw, d.Tkw);
        d.depth = f.d_depth;
        d.bgr = f.d_bgr;
        d.mask = c->d_masks.p + moff[i];
        d.xtab = c->d_tables.p + xo[i];
        d.ytab = c->d_tables.p + yo[i];
        if ((uint64_t)f.W * f.H == 0) {
            // an empty image is never visible and has no pixels, but the kernels' branch-free
            // gathers read element 0 of whatever they are given: give them readable memory
            d.depth = reinterpret_cast<const uint16_t *>(c->z_bar);
            d.bgr = reinterpret_cast<const uint8_t *>(c->z_bar);
            d.mask = reinterpret_cast<uint8_t *>(c->z_bar);
            d.xtab = c->d_tables.p;
            d.ytab = c->d_tables.p;
        }
        d.ctab = c->d_tables.p + co[i];
        d.W = f.W;
        d.H = f.H;
        d.nbr_begin = (uint32_t)nbr.size();
        for (uint32_t j : f.nbrs)
            if (neighbor_kept(c->prm, f, c->frames[j])) nbr.push_back(j);
        d.nbr_count = (uint32_t)nbr.size() - d.nbr_begin;
        f.n_filtered = d.nbr_count;
        d.order = (uint32_t)i;
        d.pad = 0;
        // more neighbours than the fused kernels hold in LDS take the two-phase chain, whose neighbours are a loop
        if (d.nbr_count > (uint32_t)kMaxK) c->big_k = true;
        if (d.nbr_count > (uint32_t)kMaxKTwoPhase)
            return fail(c, SFM_E_UNSUPPORTED,
                        "keyframe " + std::to_string(i) + " has " + std::to_string(d.nbr_count) +
                            " filtered neighbours; the per-pixel counters hold at most " + std::to_string(kMaxKTwoPhase));
    }
    std::vector<FrameDev> pack(nbr.size());
    for (size_t j = 0; j < nbr.size(); j++) pack[j] = fd[nbr[j]];
    if ((rc = ensure(c, c->d_nbrpack, pack.size()))) return rc;
    if ((rc = ensure(c, c->d_frames, n))) return rc;
    if (n) HIP_TRY(c, hipMemcpy(c->d_frames.p, fd.data(), n * sizeof(FrameDev), hipMemcpyHostToDevice));
    if ((rc = ensure(c, c->d_group_end, (n + kGroup - 1) / kGroup * kListShards * 8))) return rc;
    if ((rc = ensure(c, c->d_cand_end, (n + kGroup - 1) / kGroup))) return rc;
    uint64_t max_tiles = 1;
    for (auto &f : c->frames) max_tiles = std::max<uint64_t>(max_tiles, ((uint64_t)f.W * f.H + kTile - 1) / kTile);
    c->q_shard_cap = (uint32_t)((max_tiles + kQueueShards - 1) / kQueueShards * kTile);
    if ((rc = ensure(c, c->d_qpix, (uint64_t)kQueueShards * c->q_shard_cap))) return rc;
    if ((rc = ensure(c, c->d_qpw, 3ull * kQueueShards * c->q_shard_cap))) return rc;
    if (!pack.empty())
        HIP_TRY(c, hipMemcpy(c->d_nbrpack.p, pack.data(), pack.size() * sizeof(FrameDev), hipMemcpyHostToDevice));
    c->layout_dirty = false;
    return SFM_OK;
}

// Test hook of the div-checked build only (libsfm_fusion_checked.so): SFM_CHAIN_DIV_TRUST=1 skips the host-side operand
// bound, so that the kernels' own checks are what catches an operand outside the shared divisions' range.  The product
// build has no such switch: there the host-side bound is the only guard and cannot be turned off.
static inline bool div_trust()
{
#if SFM_DIV_CHECKED
    static const bool on = std::getenv("SFM_CHAIN_DIV_TRUST") && std::atoi(std::getenv("SFM_CHAIN_DIV_TRUST")) != 0;
    return on;
#else
    return false;
#endif
}

// One attempt with fixed capacities, first half: every launch of the attempt, the keyframe chain on stream A and the
// table kernels on stream B.  Nothing here waits for the device.  table_cap is a power of two.  *refused: the runtime
// would not take the cooperative grid (both streams are idle again; the caller reruns with the kernel pair).
// defer_tables (sfm_fusion_run_chain: somebody waits for this run's masks): the voxel filter does not run under the
// chain, which it slows by 12 % (fused) to 27 % (accept chain), but behind it -- under the successor's chain, on a
// device that has nothing else to do then, and with a grid sized for a device of its own (table_wgs_alone).
int run_launch(sfm_fusion_ctx *c, uint64_t table_cap, uint64_t shard_cap, uint64_t out_cap, bool defer_tables, bool *refused)
{
    *refused = false;
    const uint32_t table_wgs = defer_tables ? c->table_wgs_alone : c->table_wgs;
    const size_t n = c->frames.size();
    const uint64_t list_cap = shard_cap * kListShards;
    if (shard_cap > 0xFFF00000ull)  // table_group_kernel walks a shard with 32-bit offsets (and a grid-sized stride)
        return fail(c, SFM_E_UNSUPPORTED, "more than 2^32 list entries per shard");
    int rc;
    if ((rc = ensure(c, c->d_table, table_cap))) return rc;
    if ((rc = ensure(c, c->d_list_key, list_cap))) return rc;
    if ((rc = ensure(c, c->d_list_tag, list_cap))) return rc;
    // (with phase A's verdicts the fused points stay in the verdict records: finalize_kernel reads the winners there)
    if (!c->have_verdicts && (rc = ensure(c, c->d_list_xyz, 3 * list_cap))) return rc;
    // + the unused chunk tails of table_group_kernel's waves: at most two chunks per wave and launch
    if ((rc = ensure(c, c->d_cand, list_cap + (uint64_t)((c->frames.size() + kGroup - 1) / kGroup) *
                                                  std::max(c->table_wgs, c->table_wgs_alone) * kListShards * 4 * 2 * kCandChunk))) return rc;
    if ((rc = ensure(c, c->d_out_key, out_cap))) return rc;
    if ((rc = ensure(c, c->d_out_tag, out_cap))) return rc;
    if ((rc = ensure(c, c->d_out_xyz, 3 * out_cap))) return rc;
    if ((rc = ensure(c, c->d_out_bgr, out_cap))) return rc;
    // the keyframes this ctx fuses: all of them, or its block of the sequence (sfm_fusion_set_active_range)
    const size_t a0 = std::min<size_t>(c->act_first, n), a1 = std::max(a0, std::min<size_t>(c->act_end, n));
    const size_t n_groups = (a1 - a0 + kGroup - 1) / kGroup;
    while (c->gev.size() < n_groups) {
        hipEvent_t e;
        HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->gev.push_back(e);
    }
    if (c->profile)
        while (c->kev.size() < 4 * n) {
            hipEvent_t e;
            HIP_TRY(c, hipEventCreate(&e));
            c->kev.push_back(e);
        }
    hipStream_t sa = c->stream, sb = c->stream_b;
    HIP_TRY(c, hipEventRecord(c->ev[0], sa));
    if (c->masks_ready) {
        // zeros (:40-45) and the masks the earlier blocks of the sequence left are in already: made ahead of the run
        c->masks_ready = false;
    } else if (c->d_masks.n) {
        HIP_TRY(c, hipMemsetAsync(c->d_masks.p, 0, c->d_masks.n, sa));  // :40-45
        for (uint32_t i : c->seed_frames) {  // a run repeated on the same imports, or an attempt repeated
            const uint64_t px = (uint64_t)c->frames[i].W * c->frames[i].H;
            HIP_TRY(c, hipMemcpyAsync(c->d_masks.p + c->mask_off[i], c->d_mask_seed.p + c->mask_off[i], px,
                                      hipMemcpyDeviceToDevice, sa));
        }
    }
    // the table is first touched by the first group's table kernel: its fill runs on stream B, under the first chain launch
    HIP_TRY(c, hipMemsetAsync(c->d_table.p, 0xFF, table_cap * sizeof(Slot), sb));
    HIP_TRY(c, hipMemsetAsync(c->d_small, 0, kSmallZeroBytes, sa));
    // a range-scan run starts from empty ranges (a two-phase one has scanned in sfm_fusion_precompute already)
    if (c->key_scan && !c->have_verdicts && (rc = reset_key_scan(c))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_zero.p, 0, c->z_bytes, sa));  // counts, barrier words, queue cursors
    HIP_TRY(c, hipEventRecord(c->ev_init, sa));
    HIP_TRY(c, hipStreamWaitEvent(sb, c->ev_init, 0));

    ListDev list;
    list.key = c->d_list_key.p;
    list.tag = c->d_list_tag.p;
    list.xyz = c->d_list_xyz.p;
    list.shard_cap = shard_cap;
    list.cursor = dev_cursor(c);
    list.group_end = c->d_group_end.p;

    QueueDev q;
    q.pix = c->d_qpix.p;
    q.pw = c->d_qpw.p;
    q.shard_cap = c->q_shard_cap;
    q.count = nullptr;
    ScanArgs sc;
    FuseArgs a;
    a.list = list;
    a.stat = dev_stat(c);
    a.P = make_dev_params(c);

    TableArgs t;
    t.table = c->d_table.p;
    t.table_mask = table_cap - 1;
    t.list = list;
    t.ctr = dev_ctr(c);
    t.frame_before = c->z_fbefore;
    t.frame_after = c->z_fafter;
    t.cand = c->d_cand.p;

    FinalizeArgs fa;
    fa.table = c->d_table.p;
    fa.table_mask = table_cap - 1;
    fa.list = list;
    fa.cand = c->d_cand.p;
    fa.cand_end = c->d_cand_end.p;
    fa.frames = c->d_frames.p;
    fa.verdicts = c->have_verdicts ? c->d_verdicts.p : nullptr;  // two-phase chain: the winners' points are in the verdict records
    fa.v_first = (uint32_t)std::min<size_t>(c->act_first, n);
    fa.group = 0;
    fa.out_cap = out_cap;
    fa.ctr = dev_ctr(c);
    fa.out_xyz = c->d_out_xyz.p;
    fa.out_bgr = c->d_out_bgr.p;
    fa.out_key = c->d_out_key.p;
    fa.out_tag = c->d_out_tag.p;

    HIP_TRY(c, hipEventRecord(c->ev[1], sa));
    // the persistent chain kernel needs every keyframe's neighbour count to fit its LDS slots
    bool use_chain = c->chain && c->chain_items > 0;
    for (size_t i = 0; use_chain && i < n; i++)
        if (c->frames[i].n_filtered > c->chain_items) use_chain = false;
    // ... and its unchecked shared-reciprocal divisions need operands in their range (chain_div_ok)
    for (size_t i = 0; use_chain && i < n && !div_trust(); i++)
        if (!chain_div_ok(a.P, c->frames[i].K, c->frames[i].W, c->frames[i].H, c->h_frames[i].Twk, c->h_frames[i].Tkw))
            use_chain = false;
    // with the verdicts of sfm_fusion_precompute the chain is accept_chain_kernel: no LDS slots, no divisions -- all it
    // needs is the device-wide barrier; without that it is launched keyframe by keyframe
    const bool use_verdicts = c->have_verdicts;
    if (use_verdicts) use_chain = c->chain && c->chain_items > 0;
    c->chain_ran = use_chain;
    c->verdicts_ran = use_verdicts;
    // the voxel filter of one group of keyframes, on stream B
    auto launch_tables = [&](size_t g, size_t g0, size_t g1) -> int {
        t.f0 = (uint32_t)g0;
        t.f1 = (uint32_t)g1;
        t.group = (uint32_t)g;
        // diagnostic only (wrong or no cloud): 1 = chain without its co-runner, 2 / 3 = half / an eighth of the
        // shards, 4 = finalize_kernel only
        const char *diag = std::getenv("SFM_DIAG_SKIP_TABLE");
        if (diag && diag[0] == '1') return SFM_OK;
        const uint32_t shards = !diag ? kListShards : diag[0] == '2' ? kListShards / 2 : diag[0] == '3' ? kListShards / 8 : 0;
        if (shards) hipLaunchKernelGGL(table_group_kernel, dim3(table_wgs, shards), dim3(c->table_threads), 0, sb, t);
        // the group's winners are final now (later groups hold later keyframes): gather them
        HIP_TRY(c, hipMemcpyAsync(c->d_cand_end.p + g, &dev_ctr(c)->cand_count, sizeof(unsigned long long),
                                  hipMemcpyDeviceToDevice, sb));
        fa.group = (uint32_t)g;
        hipLaunchKernelGGL(finalize_kernel, dim3(table_wgs * kListShards), dim3(256), 0, sb, fa);
        return SFM_OK;
    };
    // diagnostic only (wrong cloud: the groups' list boundaries are not made): the whole chain as ONE launch, to price
    // the dispatch gaps and launch ramps between the groups' launches (needs the filter behind the chain)
    const bool diag_one_launch = defer_tables && std::getenv("SFM_DIAG_ONE_LAUNCH") != nullptr;
    for (size_t g = 0; g < n_groups; g++) {
        const size_t g0 = a0 + g * kGroup, g1 = std::min(a1, g0 + kGroup);
        bool group_event_recorded = false;
        if (use_verdicts && !use_chain) {
            // phase B without the device-wide barrier: one accept_chain_kernel launch per keyframe, the kernel
            // boundary orders the marks
            for (size_t i = g0; i < g1; i++) {
                AcceptArgs aa;
                aa.c.f0 = (uint32_t)i;
                aa.c.f1 = (uint32_t)i + 1;
                aa.c.item_cap = 0;
                aa.c.group = 0xFFFFFFFFu;
                aa.c.list = list;
                aa.c.stat = dev_stat(c);
                aa.c.bar = c->z_bar + g * kBarStride;
                aa.c.P = a.P;
                aa.c.abort_all = c->z_abort;
                aa.c.timeout_ticks = c->chain_timeout_ticks;
                aa.c.stamps = nullptr;
                aa.v_first = (uint32_t)a0;
                aa.pad = 0;
                hipLaunchKernelGGL(accept_chain_kernel, dim3(c->n_cu), dim3(kChainThreads), 0, sa,
                                   (const FrameDev *)c->d_frames.p, (const VerdictDev *)c->d_verdicts.p,
                                   (uint8_t *const *)c->d_later_masks.p, aa);
            }
        } else if (use_chain && diag_one_launch && g > 0) {
            // (diagnostic: the first launch took every keyframe)
        } else if (use_chain) {
            // :48-52 for keyframes [g0, g1) in one launch; order is kept by the in-kernel barrier
            ChainArgs ca;
            ca.f0 = (uint32_t)g0;
            ca.f1 = (uint32_t)(diag_one_launch ? a1 : g1);
            ca.item_cap = c->chain_items;
            ca.group = (uint32_t)g;
            ca.list = list;
            ca.stat = dev_stat(c);
            ca.bar = c->z_bar + g * kBarStride;
            ca.P = a.P;
            ca.abort_all = c->z_abort;
            ca.timeout_ticks = c->chain_timeout_ticks;
            ca.stamps = nullptr;
            const FrameDev *fr = c->d_frames.p;
            const FrameDev *np = c->d_nbrpack.p;
#if SFM_STAMPS
            static unsigned long long *d_cstamps = nullptr;
            const char *sg = std::getenv("SFM_STAMP_GROUP");
            const bool stamp_this = sg && (size_t)std::atoi(sg) == g && c->chain_xcd && !c->chain_coop;
            if (stamp_this) {
                const size_t bytes = (size_t)kGroup * c->n_cu * kStampSlots * 8;
                if (!d_cstamps) (void)hipMalloc(&d_cstamps, bytes);
                (void)hipMemsetAsync(d_cstamps, 0, bytes, sa);
                ca.stamps = d_cstamps;
                c->stamp_ptr = d_cstamps;
                c->stamp_n = (size_t)kGroup * c->n_cu * kStampSlots;
            }
#else
            const bool stamp_this = false;
            (void)stamp_this;
#endif
            if (c->profile) HIP_TRY(c, hipEventRecord(c->kev[4 * g], sa));
            if (use_verdicts) {
                // phase B of the two-phase chain: accept / scatter / emit from the stored verdicts
                AcceptArgs aa;
                aa.c = ca;
                const VerdictDev *vd = c->d_verdicts.p;
                uint8_t *const *lm = c->d_later_masks.p;
                aa.v_first = (uint32_t)a0;
                aa.pad = 0;
                hipEvent_t stop = c->profile ? nullptr : c->gev[g];
#if SFM_STAMPS
                if (stamp_this)
                    hipExtLaunchKernelGGL(accept_chain_kernel_stamps, dim3(c->n_cu), dim3(kChainThreads), 0, sa, nullptr, stop, 0, fr, vd, lm, aa);
                else
#endif
                hipExtLaunchKernelGGL(accept_chain_kernel, dim3(c->n_cu), dim3(kChainThreads), 0, sa, nullptr, stop, 0, fr, vd, lm, aa);
                group_event_recorded = !c->profile;
            } else if (c->chain_coop) {
                void *args[] = {(void *)&fr, (void *)&np, (void *)&ca};
                const void *kern = c->chain_xcd ? (const void *)fuse_chain_kernel : (const void *)fuse_chain_kernel_any;
                if (hipLaunchCooperativeKernel(kern, dim3(c->n_cu), dim3(kChainThreads), args,
                                               (unsigned int)c->chain_lds, sa) != hipSuccess) {
                    // the runtime refuses the grid (fewer CUs available than the device reports):
                    // same outcome as a barrier timeout -- the caller reruns with the kernel pair
                    (void)hipGetLastError();
                    HIP_TRY(c, hipStreamSynchronize(sa));
                    HIP_TRY(c, hipStreamSynchronize(sb));
                    *refused = true;
                    return SFM_OK;
                }
            } else {
                // the group's event rides on the dispatch itself: a separate event record is one more
                // packet behind every launch, and the next launch waits ~8 us longer for it
                hipEvent_t stop = c->profile ? nullptr : c->gev[g];
#if SFM_STAMPS
                if (stamp_this)
                    hipExtLaunchKernelGGL(fuse_chain_kernel_stamps, dim3(c->n_cu), dim3(kChainThreads), c->chain_lds, sa,
                                          nullptr, stop, 0, fr, np, ca);
                else
#endif
                if (c->chain_xcd)  // XCD-aware piece mapping (8 XCDs x 32 CUs)
                    hipExtLaunchKernelGGL(fuse_chain_kernel, dim3(c->n_cu), dim3(kChainThreads), c->chain_lds, sa, nullptr,
                                          stop, 0, fr, np, ca);
                else
                    hipExtLaunchKernelGGL(fuse_chain_kernel_any, dim3(c->n_cu), dim3(kChainThreads), c->chain_lds, sa,
                                          nullptr, stop, 0, fr, np, ca);
                group_event_recorded = !c->profile;
            }
            if (c->profile) HIP_TRY(c, hipEventRecord(c->kev[4 * g + 1], sa));
        } else
        for (size_t i = g0; i < g1 && !use_verdicts; i++) {  // :48-52, strictly in order
            const HostFrame &f = c->frames[i];
            const uint64_t px = (uint64_t)f.W * f.H;
            if (!px) continue;
            const uint32_t nw = std::min<uint32_t>(std::max<uint32_t>(f.n_filtered, 1), c->max_waves);
            q.count = c->z_qcount + i * kQueueShards * 16;
            sc.F = c->h_frames[i];
            sc.q = q;
            sc.P = a.P;
            a.F = c->h_frames[i];
            a.q = q;
            const FrameDev *nbp = c->d_nbrpack.p + a.F.nbr_begin;
            const uint32_t tiles = (uint32_t)((px + kTile - 1) / kTile);
            // every chunk of 64 survivors gets a workgroup; beyond what fits on the chip at
            // once the workgroups loop
            const uint32_t resident = (uint32_t)c->n_cu * std::max<uint32_t>(1, c->waves_per_cu / nw);
            const uint32_t wgs = (uint32_t)std::min<uint64_t>((px + kWave - 1) / kWave, resident);
            const size_t lds = fuse_lds_bytes(f.n_filtered);
            if (c->profile) {
                hipExtLaunchKernelGGL(scan_frame_kernel, dim3(tiles), dim3(kWave), 0, sa, c->kev[4 * i],
                                      c->kev[4 * i + 1], 0, sc);
                hipExtLaunchKernelGGL(observe_frame_kernel, dim3(wgs), dim3(nw * kWave), lds, sa,
                                      c->kev[4 * i + 2], c->kev[4 * i + 3], 0, nbp, a);
            } else {
                hipLaunchKernelGGL(scan_frame_kernel, dim3(tiles), dim3(kWave), 0, sa, sc);
                hipLaunchKernelGGL(observe_frame_kernel, dim3(wgs), dim3(nw * kWave), lds, sa, nbp, a);
            }
        }
        // list boundaries of the group: fuse_chain_kernel's last workgroup has written them; the kernel
        // pair needs a snapshot of the append cursors, in stream order
        if (!use_chain)
            HIP_TRY(c, hipMemcpyAsync(c->d_group_end.p + g * kListShards * 8, dev_cursor(c),
                                      kListShards * 8 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, sa));
        if (!group_event_recorded) HIP_TRY(c, hipEventRecord(c->gev[g], sa));
        if (!defer_tables) {
            HIP_TRY(c, hipStreamWaitEvent(sb, c->gev[g], 0));
            if ((rc = launch_tables(g, g0, g1))) return rc;
        }
    }
    if (defer_tables && n_groups) {
        HIP_TRY(c, hipStreamWaitEvent(sb, c->gev[n_groups - 1], 0));
        for (size_t g = 0; g < n_groups; g++) {
            const size_t g0 = a0 + g * kGroup, g1 = std::min(a1, g0 + kGroup);
            if ((rc = launch_tables(g, g0, g1))) return rc;
        }
    }
    HIP_TRY(c, hipGetLastError());
#if SFM_STAMPS
    if (c->stamp_ptr) {
        std::vector<unsigned long long> h(c->stamp_n);
        (void)hipStreamSynchronize(sa);
        (void)hipMemcpy(h.data(), c->stamp_ptr, c->stamp_n * 8, hipMemcpyDeviceToHost);
        if (const char *o = std::getenv("SFM_STAMP_OUT")) {
            if (FILE *fp = std::fopen(o, "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
        }
    }
#endif
    HIP_TRY(c, hipEventRecord(c->ev[2], sa));
    HIP_TRY(c, hipEventRecord(c->ev_b_done, sb));
    return SFM_OK;
}

// The chain's own error bits, read as soon as the chain has finished; the table kernels of the last groups may still
// run on stream B (sfm_fusion_run_chain: the visibility masks are final here, and a rank of a sequence split hands
// them on without waiting for its cloud).
int chain_status(sfm_fusion_ctx *c, uint32_t *err_bits)
{
    hipStream_t sa = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->h_chain, dev_stat(c), sizeof(StatShard) * kStatShards, hipMemcpyDeviceToHost, sa));
    HIP_TRY(c, hipMemcpyAsync(c->h_chain + kStatShards, dev_pa_stat(c), sizeof(StatShard) * kStatShards,
                              hipMemcpyDeviceToHost, sa));
    HIP_TRY(c, hipStreamSynchronize(sa));
    uint32_t err = 0;
    for (int s = 0; s < kStatShards; s++) err |= c->h_chain[s].err;
    if (c->verdicts_ran)
        for (int s = 0; s < kStatShards; s++)
            if (c->h_chain[kStatShards + s].err & kErrDivRange) err |= kErrDivRangeA;
    *err_bits = err;
    return SFM_OK;
}

// Second half of an attempt: wait for both streams, read the counters back.
int run_collect(sfm_fusion_ctx *c, uint64_t shard_cap, uint32_t *err_bits)
{
    const size_t n = c->frames.size();
    const bool use_verdicts = c->verdicts_ran;
    hipStream_t sa = c->stream;
    HIP_TRY(c, hipStreamWaitEvent(sa, c->ev_b_done, 0));
    HIP_TRY(c, hipEventRecord(c->ev[3], sa));
    HIP_TRY(c, hipMemcpyAsync(c->h_small, c->d_small, kSmallBytes, hipMemcpyDeviceToHost, sa));
    if (n)  // the per-keyframe counts ride behind the counters: z_fbefore | z_fafter are one block of 2n words
        HIP_TRY(c, hipMemcpyAsync(c->h_fcounts, c->z_fbefore, 2 * n * sizeof(uint64_t), hipMemcpyDeviceToHost, sa));
    HIP_TRY(c, hipStreamSynchronize(sa));  // the run's one host synchronisation
    uint32_t err = c->h_small->ctr.err;
    for (int s = 0; s < kStatShards; s++) err |= c->h_small->stat[s].err;
    if (use_verdicts)  // phase A's divisions happened in sfm_fusion_precompute
        for (int s = 0; s < kStatShards; s++)
            if (c->h_small->pa_stat[s].err & kErrDivRange) err |= kErrDivRangeA;
    uint64_t n_list = 0;
    for (int s = 0; s < kListShards; s++) {
        if (c->h_small->cursor[s * 8] > shard_cap) err |= kErrListFull;
        n_list += std::min<uint64_t>(c->h_small->cursor[s * 8], shard_cap);
    }
    *err_bits = err;  // the caller decides: grow and retry, or report
    (void)n_list;
    return SFM_OK;
}

}  // namespace

extern "C" {

void sfm_fusion_default_params(sfm_fusion_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->base_line = 0.50;
    p->T_cov = 0.5;
    p->voxel_size = 0.05;
    p->T_dist = 0.50;
    p->min_dist_filter = 0.0;
    p->dir_dot = 0.30;
    p->S[0] = 0.05;
    p->S[1] = 0.05;
    p->S[2] = 10.0;
    p->depth_div = 1000.0;
    p->close_frac = 0.50;
    p->table_capacity = 0;
    p->max_points = 0;
}

int sfm_fusion_create(const sfm_fusion_params *params, int device, void *stream, sfm_fusion_ctx **out)
{
    if (!out) return SFM_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SFM_E_HIP;
    sfm_fusion_ctx *c = new (std::nothrow) sfm_fusion_ctx();
    if (!c) return SFM_E_NOMEM;
    if (params) c->prm = *params; else sfm_fusion_default_params(&c->prm);
    c->device = device;
    if (const char *e = std::getenv("SFM_FUSE_WAVES")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= kMaxWaves) c->max_waves = (uint32_t)v;
    }
    if (const char *e = std::getenv("SFM_FUSE_RESIDENT")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 64) c->waves_per_cu = (uint32_t)v;
    }
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->n_cu = v;
    }
    std::memset(&c->stats, 0, sizeof c->stats);
    bool ok = hipSetDevice(device) == hipSuccess;
    if (ok && c->chain) {
        // fuse_chain_kernel: as many (chunk, neighbour) LDS slots as one workgroup may allocate
        int lds_max = 0, coop = 0;
        if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) lds_max = 0;
        if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) != hipSuccess) coop = 0;
        const size_t fixed = chain_lds_bytes(0);
        // 4 KiB of every CU's LDS stay free: the table kernels of the previous group (512 B per
        // workgroup) must be able to share the CUs, or each chain launch waits for them to drain
        const size_t budget = lds_max > 4096 ? (size_t)lds_max - 4096 : 0;
        uint32_t items = budget > fixed ? (uint32_t)((budget - fixed) / (chain_lds_bytes(1) - fixed)) : 0;
        items = std::min<uint32_t>(items, kChainMaxItems);
        if (items >= 8 && (coop || !c->chain_coop)) {
            c->chain_items = items;
            c->chain_lds = chain_lds_bytes(items);
            if (hipFuncSetAttribute((const void *)fuse_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)c->chain_lds) != hipSuccess ||
                hipFuncSetAttribute((const void *)fuse_chain_kernel_any, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)c->chain_lds) != hipSuccess) {
                (void)hipGetLastError();
                c->chain_items = 0;
            }
            // the device-wide barrier needs all n_cu workgroups resident at once: ask the runtime whether one
            // workgroup of this size and LDS fits a CU at all (a plain launch performs no such check)
            int per_cu = 0;
            if (c->chain_items &&
                (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fuse_chain_kernel, kChainThreads,
                                                              c->chain_lds) != hipSuccess || per_cu < 1)) {
                (void)hipGetLastError();
                c->chain_items = 0;
            }
        }
    }
    if (const char *e = std::getenv("SFM_CHAIN_TIMEOUT_TICKS")) {
        const long long v = std::atoll(e);
        if (v >= 1) c->chain_timeout_ticks = (unsigned long long)v;
    }
    c->chain_xcd = c->n_cu == 256;
    if (const char *e = std::getenv("SFM_CHAIN_XCD")) c->chain_xcd = c->chain_xcd && std::atoi(e) != 0;
    if (const char *e = std::getenv("SFM_FUSE_CHAIN")) c->chain = std::atoi(e) != 0;
    if (const char *e = std::getenv("SFM_CHAIN_COOP")) c->chain_coop = std::atoi(e) != 0;
    if (const char *e = std::getenv("SFM_PRECOMPUTE")) c->auto_precompute = std::atoi(e) != 0;
    if (const char *e = std::getenv("SFM_TABLE_THREADS")) {
        const int v = std::atoi(e);
        if (v == 64 || v == 128 || v == 256) c->table_threads = (uint32_t)v;
    }
    if (const char *e = std::getenv("SFM_TABLE_WGS")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 1024) c->table_wgs = (uint32_t)v;
    }
    if (const char *e = std::getenv("SFM_TABLE_WGS_ALONE")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 1024) c->table_wgs_alone = (uint32_t)v;
    }
    if (ok && stream) {
        c->stream = (hipStream_t)stream;
    } else if (ok) {
        ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
        c->own_stream = ok;
    }
    ok = ok && hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->stream_up, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming) == hipSuccess;
    for (auto &st : c->stage) ok = ok && hipEventCreateWithFlags(&st.ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(&c->d_small, kSmallBytes) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_part, kScratchParts * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess;
    if (ok && device < kMaxDevices) {
        // once per device, with that device current (hipSetDevice above): each GPU has its own copy of the array
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (!g_scratch_ptr[device]) {
            void *p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) == hipSuccess && p)
                g_scratch_ptr[device] = p;
            else
                (void)hipGetLastError();  // the no-alloc entry point reports SFM_E_UNSUPPORTED on this device
        }
        c->scratch = static_cast<unsigned long long *>(g_scratch_ptr[device]);
    }
    ok = ok && hipHostMalloc(&c->h_small, kSmallBytes, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_chain, 2 * kStatShards * sizeof(StatShard), hipHostMallocDefault) == hipSuccess;
    for (int i = 0; ok && i < 4; i++) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    for (int i = 0; ok && i < 2; i++) ok = hipEventCreate(&c->ev_pa[i]) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_init, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_b_done, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        sfm_fusion_destroy(c);
        return SFM_E_HIP;
    }
    *out = c;
    return SFM_OK;
}

void sfm_fusion_destroy(sfm_fusion_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream_b) (void)hipStreamSynchronize(c->stream_b);
    if (c->stream_up) (void)hipStreamSynchronize(c->stream_up);
    for (void *p : c->arena.chunks) (void)hipFree(p);
    for (auto &st : c->stage) {
        if (st.h) (void)hipHostFree(st.h);
        if (st.ev) (void)hipEventDestroy(st.ev);
    }
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    if (c->stream_up) (void)hipStreamDestroy(c->stream_up);
    release(c->d_rd_keys); release(c->d_rd_key3); release(c->d_rd_tag); release(c->d_rd_perm); release(c->d_rd_tmp);
    release(c->d_rd_xyz); release(c->d_rd_bgr);
    release(c->d_mask_seed); release(c->d_verdict_arena); release(c->d_verdicts); release(c->d_later_masks);
    release(c->d_part); release(c->d_gorder); release(c->d_tables); release(c->d_nbrpack); release(c->d_masks); release(c->d_table);
    release(c->d_list_key); release(c->d_list_tag); release(c->d_out_key);
    release(c->d_out_tag); release(c->d_cand); release(c->d_list_xyz); release(c->d_out_xyz);
    release(c->d_qpix); release(c->d_qpw); release(c->d_frames); release(c->d_zero);
    release(c->d_out_bgr); release(c->d_group_end); release(c->d_cand_end);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->h_chain) (void)hipHostFree(c->h_chain);
    if (c->h_part) (void)hipHostFree(c->h_part);
    if (c->h_fcounts) (void)hipHostFree(c->h_fcounts);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_pa)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_init) (void)hipEventDestroy(c->ev_init);
    if (c->ev_b_done) (void)hipEventDestroy(c->ev_b_done);
    for (auto &e : c->gev) (void)hipEventDestroy(e);
    for (auto &e : c->kev) (void)hipEventDestroy(e);
    if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sfm_fusion_add_keyframe(sfm_fusion_ctx *c, uint32_t order_idx, int32_t W, int32_t H,
                            const uint16_t *depth, size_t depth_stride, const uint8_t *bgr,
                            size_t bgr_stride, const double K[4], const double T_wk[16],
                            const double T_kw[16], int mem_kind)
{
    if (!c) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    if (order_idx != c->frames.size())
        return fail(c, SFM_E_INVALID, "keyframes must be added in order 0,1,2,...");
    if (W < 0 || H < 0 || !K || !T_wk || !T_kw) return fail(c, SFM_E_INVALID, "bad keyframe argument");
    if ((uint64_t)W * (uint64_t)H > 0xFFFFFF00ull) return fail(c, SFM_E_INVALID, "image too large");
    const uint64_t px = (uint64_t)W * H;
    if (px && (!depth || !bgr)) return fail(c, SFM_E_INVALID, "null image pointer");
    if (px && (depth_stride < 2ull * W || bgr_stride < 3ull * W))
        return fail(c, SFM_E_INVALID, "row stride smaller than a row");
    if (mem_kind != SFM_MEM_HOST && mem_kind != SFM_MEM_DEVICE) return fail(c, SFM_E_INVALID, "bad mem_kind");
    HIP_TRY(c, hipSetDevice(c->device));
    HostFrame f;
    f.W = W;
    f.H = H;
    std::memcpy(f.K, K, sizeof f.K);
    std::memcpy(f.Twk, T_wk, sizeof f.Twk);
    std::memcpy(f.Tkw, T_kw, sizeof f.Tkw);
    if (px) {
        // +8 bytes of slack: the kernel reads depth as ushort4 / colour per byte
        int rc;
        if ((rc = arena_alloc(c, px * 2 + 8, (void **)&f.d_depth))) return rc;
        if ((rc = arena_alloc(c, px * 3 + 8, (void **)&f.d_bgr))) return rc;
        if (mem_kind == SFM_MEM_HOST) {
            auto &st = c->stage[c->stage_i];
            c->stage_i ^= 1;
            if (st.busy) {
                HIP_TRY(c, hipEventSynchronize(st.ev));
                st.busy = false;
            }
            if (st.cap < px * 5) {
                if (st.h) HIP_TRY(c, hipHostFree(st.h));
                st.h = nullptr;
                st.cap = 0;
                HIP_TRY(c, hipHostMalloc((void **)&st.h, px * 5, hipHostMallocDefault));
                st.cap = px * 5;
            }
            // the caller's buffers are free again when this returns
            pack_rows(st.h, depth, 2ull * W, depth_stride, H);
            pack_rows(st.h + px * 2, bgr, 3ull * W, bgr_stride, H);
            HIP_TRY(c, hipMemcpyAsync(f.d_depth, st.h, px * 2, hipMemcpyHostToDevice, c->stream_up));
            HIP_TRY(c, hipMemcpyAsync(f.d_bgr, st.h + px * 2, px * 3, hipMemcpyHostToDevice, c->stream_up));
            HIP_TRY(c, hipEventRecord(st.ev, c->stream_up));
            st.busy = true;
            c->up_pending = true;
        } else {
            hipError_t e1 = hipMemcpy2D(f.d_depth, 2ull * W, depth, depth_stride, 2ull * W, H, hipMemcpyDeviceToDevice);
            hipError_t e2 = hipMemcpy2D(f.d_bgr, 3ull * W, bgr, bgr_stride, 3ull * W, H, hipMemcpyDeviceToDevice);
            if (e1 != hipSuccess || e2 != hipSuccess)
                return fail(c, SFM_E_HIP, std::string("hipMemcpy2D: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        }
    }
    c->frames.push_back(std::move(f));
    c->layout_dirty = true;
    c->have_result = c->have_masks = false;
    return SFM_OK;
}

// Sophus::SE3<double>::exp(tangent).matrix() and its inverse, column-major.  Tangent order as in Sophus:
// (upsilon = translational part, omega = rotation vector).  Restated from Sophus' published formulas
// (so3.hpp expAndTheta: unit quaternion from the half angle, Taylor branch below |omega| < 1e-10; se3.hpp exp:
// t = V * upsilon, V = I + (1 - cos th)/th^2 * Om + (th - sin th)/th^3 * Om^2; Eigen's Quaternion::toRotationMatrix).
void sfm_fusion_se3_exp(const double se3[6], double T[16], double T_inv[16])
{
    const double ux = se3[0], uy = se3[1], uz = se3[2], wx = se3[3], wy = se3[4], wz = se3[5];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    const double eps = 1e-10;
    double theta, imag, real;
    if (theta_sq < eps * eps) {
        theta = 0.0;
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        theta = std::sqrt(theta_sq);
        const double half = 0.5 * theta;
        imag = std::sin(half) / theta;
        real = std::cos(half);
    }
    const double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
    // Eigen::Quaternion::toRotationMatrix
    const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    double R[3][3];
    R[0][0] = 1.0 - (tyy + tzz); R[0][1] = txy - twz;         R[0][2] = txz + twy;
    R[1][0] = txy + twz;         R[1][1] = 1.0 - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy;         R[2][1] = tyz + twx;         R[2][2] = 1.0 - (txx + tyy);
    double V[3][3];
    if (std::fabs(theta) < eps) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) V[i][j] = R[i][j];  // "an accurate expansion" (se3.hpp)
    } else {
        const double Om[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
        double Om2[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Om2[i][j] = Om[i][0] * Om[0][j] + Om[i][1] * Om[1][j] + Om[i][2] * Om[2][j];
        const double a = (1.0 - std::cos(theta)) / theta_sq;
        const double b = (theta - std::sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) V[i][j] = (i == j ? 1.0 : 0.0) + a * Om[i][j] + b * Om2[i][j];
    }
    const double t[3] = {V[0][0] * ux + V[0][1] * uy + V[0][2] * uz, V[1][0] * ux + V[1][1] * uy + V[1][2] * uz,
                         V[2][0] * ux + V[2][1] * uy + V[2][2] * uz};
    if (T) {
        for (int col = 0; col < 3; col++)
            for (int r = 0; r < 3; r++) T[r + 4 * col] = R[r][col];
        T[12] = t[0]; T[13] = t[1]; T[14] = t[2];
        T[3] = T[7] = T[11] = 0.0; T[15] = 1.0;
    }
    if (T_inv) {  // rigid inverse [R^T | -R^T t]
        for (int col = 0; col < 3; col++)
            for (int r = 0; r < 3; r++) T_inv[r + 4 * col] = R[col][r];
        for (int r = 0; r < 3; r++) T_inv[12 + r] = -(R[0][r] * t[0] + R[1][r] * t[1] + R[2][r] * t[2]);
        T_inv[3] = T_inv[7] = T_inv[11] = 0.0; T_inv[15] = 1.0;
    }
}

int sfm_fusion_add_keyframe_se3(sfm_fusion_ctx *c, uint32_t order_idx, int32_t W, int32_t H, const uint16_t *depth,
                                size_t depth_stride, const uint8_t *bgr, size_t bgr_stride, const double K[4],
                                const double se3_kw[6], int mem_kind)
{
    if (!c) return SFM_E_INVALID;
    if (!se3_kw) return fail(c, SFM_E_INVALID, "null pose");
    double T_kw[16], T_wk[16];
    sfm_fusion_se3_exp(se3_kw, T_kw, T_wk);  // set_world_to_camera(tf): T_kw = tf, T_wk = tf.inverse()  (keyframe.cpp:56-60)
    return sfm_fusion_add_keyframe(c, order_idx, W, H, depth, depth_stride, bgr, bgr_stride, K, T_wk, T_kw, mem_kind);
}

int sfm_fusion_set_neighbors(sfm_fusion_ctx *c, uint32_t order_idx, const uint32_t *nbr, uint32_t n)
{
    if (!c) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    if (order_idx >= c->frames.size()) return fail(c, SFM_E_INVALID, "unknown keyframe");
    if (n && !nbr) return fail(c, SFM_E_INVALID, "null neighbour list");
    c->frames[order_idx].nbrs.assign(nbr, nbr + n);
    c->frames[order_idx].selected = true;
    c->layout_dirty = true;
    c->have_result = c->have_masks = false;
    return SFM_OK;
}

int sfm_fusion_set_covisibility(sfm_fusion_ctx *c, const uint32_t *edge_a, const uint32_t *edge_b,
                                const uint32_t *edge_count, uint64_t n_edges, uint32_t select_min,
                                uint32_t neighbor_min)
{
    if (!c) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    if (n_edges && (!edge_a || !edge_b || !edge_count)) return fail(c, SFM_E_INVALID, "null edge array");
    const size_t n = c->frames.size();
    for (uint64_t e = 0; e < n_edges; e++) {
        if (edge_a[e] >= n || edge_b[e] >= n) return fail(c, SFM_E_INVALID, "edge names an unknown keyframe");
        if (edge_a[e] == edge_b[e]) return fail(c, SFM_E_INVALID, "a keyframe cannot be its own neighbour");
    }
    for (auto &f : c->frames) {
        f.nbrs.clear();
        f.selected = false;
    }
    // The first insertion of an edge wins in the reference and freezes its count (covisibility_insert,
    // map.cpp:108-119; edges are keyed by the unordered pair, map.cpp:354-363): a repeated pair, in either
    // direction, is ignored -- whatever count it carries.
    std::vector<uint64_t> seen;
    seen.reserve(n_edges);
    for (uint64_t e = 0; e < n_edges; e++) {
        const uint32_t lo = std::min(edge_a[e], edge_b[e]), hi = std::max(edge_a[e], edge_b[e]);
        seen.push_back(((uint64_t)lo << 32) | hi);
    }
    std::vector<uint64_t> order(n_edges);
    std::iota(order.begin(), order.end(), (uint64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) { return seen[x] < seen[y]; });
    std::vector<bool> first(n_edges, false);
    for (uint64_t i = 0; i < n_edges; i++)
        if (i == 0 || seen[order[i]] != seen[order[i - 1]]) first[order[i]] = true;  // stable: smallest index of the pair
    for (uint64_t e = 0; e < n_edges; e++) {
        if (!first[e]) continue;
        HostFrame &fa = c->frames[edge_a[e]], &fb = c->frames[edge_b[e]];
        if (edge_count[e] >= select_min) fa.selected = fb.selected = true;  // dense_reconstruction.cpp:32-38
        if (edge_count[e] > neighbor_min) {                                 // map.cpp:157
            fa.nbrs.push_back(edge_b[e]);
            fb.nbrs.push_back(edge_a[e]);
        }
    }
    for (auto &f : c->frames) {
        // canonical neighbour order: ascending keyframe index
        std::sort(f.nbrs.begin(), f.nbrs.end());
        if (!f.selected) f.nbrs.clear();
    }
    c->layout_dirty = true;
    c->have_result = c->have_masks = false;
    return SFM_OK;
}

namespace {
// sfm_fusion_run and its two halves.  stop_after_chain: return as soon as the keyframe chain has finished without a
// chain error (the masks are final; the table kernels still run); the next call goes on from there.
int run_driver(sfm_fusion_ctx *c, bool stop_after_chain)
{
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->frames.size();
    int rc;
    auto &rs = c->rs;
    if (!rs.launched) {
        for (size_t i = 0; i < n; i++)
            for (uint32_t j : c->frames[i].nbrs) {
                if (j >= n) return fail(c, SFM_E_INVALID, "neighbour index out of range");
                if (j == i) return fail(c, SFM_E_INVALID, "a keyframe cannot be its own neighbour");
            }
        if (c->up_pending) {  // the images still in flight on the upload stream
            HIP_TRY(c, hipEventRecord(c->ev_up, c->stream_up));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_up, 0));
            c->up_pending = false;
        }
        if (c->layout_dirty && (rc = build_layout(c))) return rc;
        c->have_result = c->have_masks = false;
        // keyframes with more neighbours than the fused kernels hold always take the two-phase chain
        if ((c->big_k || (c->auto_precompute && c->chain)) && !c->have_verdicts && (rc = sfm_fusion_precompute(c))) return rc;
        c->part_counts.clear();  // sfm_fusion_export_records needs counts of THIS run's cloud
        // a fallback is not for life: after kChainRetryRuns runs on the kernel pair the chain kernel is tried again
        // (three fallbacks in a ctx's life make it permanent: something else owns CUs of this device for good)
        if (!c->chain && c->chain_off_runs > 0 && --c->chain_off_runs == 0 && c->chain_fallbacks < 3) c->chain = true;
    }
    const bool fixed_tab = c->prm.table_capacity != 0, fixed_list = c->prm.max_points != 0;
    if (!rs.launched) {
        // Sizes: explicit, else what the previous run ended with, else a first guess that the
        // retry loop below doubles until the run fits.
        rs.shard_cap = fixed_list ? (c->prm.max_points + kListShards - 1) / kListShards
                                  : std::max<uint64_t>(c->d_list_key.n / kListShards,
                                                       std::max<uint64_t>(c->total_px / 4, 1 << 16) / kListShards);
        rs.table_cap = fixed_tab ? pow2_at_least(c->prm.table_capacity)
                                 : pow2_at_least(std::max<uint64_t>(c->d_table.n,
                                                                    std::max<uint64_t>(c->total_px / 16, 1 << 16)));
        // output slots: what the previous run ended with, else one per 16 pixels (C2 needs one per 97, the
        // 5-mm C5 sample one per 21); the exact need is known after an overflowing run
        rs.out_cap = std::max<uint64_t>(c->out_cap, std::max<uint64_t>(c->total_px / 16, 1 << 12));
        rs.attempt = rs.key_attempts = 0;
        c->key_scan = false;
    }
    uint64_t &shard_cap = rs.shard_cap, &table_cap = rs.table_cap, &out_cap = rs.out_cap;
    int &key_attempts = rs.key_attempts;
    const uint64_t shard_max = c->total_px + kTile;  // every pixel live, all tiles in one shard
    for (;; rs.attempt++) {
        const int attempt = rs.attempt;
        uint32_t err = 0;
        if (!rs.launched) {
            bool refused = false;
            if ((rc = run_launch(c, table_cap, shard_cap, out_cap, stop_after_chain, &refused))) return rc;
            if (refused) {
                err = kErrBarrier;
            } else {
                rs.launched = true;
                if (stop_after_chain) {
                    uint32_t cerr = 0;
                    if ((rc = chain_status(c, &cerr))) return rc;
                    if (!(cerr & (kErrBarrier | kErrDivRange | kErrDivRangeA))) {
                        // every keyframe of the range has marked what it observed.  Whatever a later attempt may still
                        // have to repeat (a full list or table, an unfit key packing), it computes the same masks.
                        c->have_masks = true;
                        return SFM_OK;
                    }
                }
            }
        }
        if (rs.launched) {
            rs.launched = false;
            if ((rc = run_collect(c, shard_cap, &err))) return rc;
        }
        const bool crowded = !err && !fixed_tab && c->h_small->ctr.table_used * 2 > table_cap;
        if (!err && !crowded) break;
        if (err & kErrDivRangeA) {
            // phase A (verdict_kernel) met a division operand outside the shared-reciprocal range: its verdicts are void.
            // Recompute them with plain divisions and run again; counted like a fallback of the fused chain.
            if (c->verdict_plain) return fail(c, SFM_E_HIP, "division-range error from a kernel that divides plainly");
            c->verdict_plain = true;
            c->have_verdicts = false;
            c->chain_fallbacks++;
            if ((rc = sfm_fusion_precompute(c))) return rc;
            continue;
        }
        if (err & (kErrBarrier | kErrDivRange)) {
            // kErrBarrier: the persistent kernel's workgroups were not all resident within the time limit
            // (another process or stream holds CUs).  Nothing hangs: every workgroup has left.
            // kErrDivRange: a division operand outside the range of the chain kernel's shared-reciprocal
            // divisions (|x| < 2^-400 or > 2^400).  Either way this ctx goes on with the per-keyframe
            // kernel pair, which needs no co-residency and divides plainly.
            if (!c->chain) return fail(c, SFM_E_HIP, "device-wide barrier timed out");
            c->chain = false;
            c->chain_fallbacks++;
            c->chain_off_runs = 8;
            continue;
        }
        if (err & kErrKeyRange) {
            // Some voxel index does not fit the key packing.  Next attempt: a range-scan run (no records, every fused
            // point leaves its indices); the one after packs with exactly the ranges found.  The packing is exact while
            // the three code widths sum to <= 63 bits.
            if (!c->key_scan) {
                if (key_attempts++ >= 2) return fail(c, SFM_E_KEY_RANGE, "voxel indices outside the ranges a scan found");
                c->key_scan = true;
            } else {
                c->key_scan = false;
                KeyRangeDev r = c->h_small->key_scan[0];
                for (int sh = 1; sh < kStatShards; sh++)
                    for (int ax = 0; ax < 3; ax++) {
                        r.lo[ax] = std::min(r.lo[ax], c->h_small->key_scan[sh].lo[ax]);
                        r.hi[ax] = std::max(r.hi[ax], c->h_small->key_scan[sh].hi[ax]);
                    }
                for (int ax = 0; ax < 3; ax++)
                    if (r.lo[ax] > r.hi[ax]) r.lo[ax] = r.hi[ax] = 0;  // only indefinite indices on this axis
                if (const char *why = key_range_problem(r.lo, r.hi))
                    return fail(c, SFM_E_KEY_RANGE, std::string("no exact 64-bit voxel key for this cloud: ") + why);
                c->key = r;
                for (int ax = 0; ax < 3; ax++) c->key_bits[ax] = bits_for_range(r.lo[ax], r.hi[ax]);
            }
            if (c->have_verdicts) {  // their records hold keys of the old packing
                c->have_verdicts = false;
                if ((rc = sfm_fusion_precompute(c))) return rc;
            }
            continue;
        }
        if (c->key_scan) {  // a scan run that found nothing unfit cannot happen; never leave the mode on
            c->key_scan = false;
            return fail(c, SFM_E_KEY_RANGE, "range scan ended without a range");
        }
        bool grown = false;
        if (err & kErrOutFull) {
            const uint64_t need = c->h_small->ctr.out_count;  // the counter kept counting
            out_cap = std::max<uint64_t>(out_cap * 2, need + need / 16);
            grown = true;
        }
        if ((err & kErrListFull) && !fixed_list && shard_cap < shard_max) {
            uint64_t need = 0;
            for (int s = 0; s < kListShards; s++) need = std::max<uint64_t>(need, c->h_small->cursor[s * 8]);
            shard_cap = std::min<uint64_t>(shard_max, std::max<uint64_t>(shard_cap * 2, need + need / 8));
            grown = true;
        }
        if (((err & kErrTableFull) || crowded) && !fixed_tab && attempt < 16) {
            table_cap *= 2;  // keeps the load factor under 1/2
            grown = true;
        }
        if (!grown) {
            if (err & kErrListFull) return fail(c, SFM_E_CAPACITY, "candidate list capacity (max_points) exhausted");
            return fail(c, SFM_E_TABLE_FULL, "voxel table capacity exhausted");
        }
    }
    c->n_out = c->h_small->ctr.out_count;
    c->out_cap = out_cap;
    c->merged = false;
    sfm_fusion_stats &st = c->stats;
    std::memset(&st, 0, sizeof st);
    st.n_frames = n;
    st.n_px = c->total_px;
    for (int s = 0; s < kStatShards; s++) {
        st.n_gather += c->h_small->stat[s].n_gather;
        st.n_hit += c->h_small->stat[s].n_hit;
    }
    st.n_vox = c->n_out;
    st.global_after = c->n_out;
    st.table_capacity = table_cap;
    st.table_used = c->h_small->ctr.table_used;
    st.max_probe = c->h_small->ctr.max_probe;
    c->fbefore.assign(c->h_fcounts, c->h_fcounts + n);  // copied behind the counters, before the run's one synchronisation
    c->fafter.assign(c->h_fcounts + n, c->h_fcounts + 2 * n);
    st.n_acc = std::accumulate(c->fbefore.begin(), c->fbefore.end(), (uint64_t)0);
    st.global_before = std::accumulate(c->fafter.begin(), c->fafter.end(), (uint64_t)0);
    float ms = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[3]));
    st.last_run_ms = ms;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2]));
    st.frame_kernels_ms = ms;
    st.fuse_kernel_launches = 0;
    st.fuse_kernel_sum_ms = 0;
    st.scan_kernel_sum_ms = 0;
    st.fused_chain = c->chain_ran ? 1 : 0;
    st.chain_fallbacks = c->chain_fallbacks;
    st.precomputed = c->verdicts_ran ? 1 : 0;
    st.precompute_ms = 0;
    if (c->verdicts_ran && c->pa_timed) {  // the events completed before this run's kernels on the same stream
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pa[0], c->ev_pa[1]));
        st.precompute_ms = ms;
    }
    const size_t act0 = std::min<size_t>(c->act_first, n), act1 = std::max(act0, std::min<size_t>(c->act_end, n));
    if (c->profile && c->chain_ran) {
        for (size_t g = 0; g < (act1 - act0 + kGroup - 1) / kGroup; g++) {
            HIP_TRY(c, hipEventElapsedTime(&ms, c->kev[4 * g], c->kev[4 * g + 1]));
            st.fuse_kernel_sum_ms += ms;
            st.fuse_kernel_launches++;
        }
    } else if (c->profile) {
        for (size_t i = act0; i < act1; i++) {
            if ((uint64_t)c->frames[i].W * c->frames[i].H == 0) continue;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->kev[4 * i], c->kev[4 * i + 1]));
            st.scan_kernel_sum_ms += ms;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->kev[4 * i + 2], c->kev[4 * i + 3]));
            st.fuse_kernel_sum_ms += ms;
            st.fuse_kernel_launches++;
        }
    }
    c->have_result = c->have_masks = true;
    return SFM_OK;
}

}  // namespace

int sfm_fusion_run(sfm_fusion_ctx *c)
{
    if (!c) return SFM_E_INVALID;
    return run_driver(c, false);  // (after sfm_fusion_run_chain: the rest of that run)
}

int sfm_fusion_run_chain(sfm_fusion_ctx *c)
{
    if (!c) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    return run_driver(c, true);
}

int sfm_fusion_finish(sfm_fusion_ctx *c)
{
    if (!c) return SFM_E_INVALID;
    if (!c->rs.launched) return fail(c, SFM_E_INVALID, "no run in flight: call sfm_fusion_run_chain first");
    return run_driver(c, false);
}

int sfm_fusion_precompute(sfm_fusion_ctx *c)
{
    if (!c) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if (c->up_pending) {
        HIP_TRY(c, hipEventRecord(c->ev_up, c->stream_up));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_up, 0));
        c->up_pending = false;
    }
    if (c->layout_dirty && (rc = build_layout(c))) return rc;
    c->have_verdicts = false;
    const size_t n = c->frames.size();
    for (size_t i = 0; i < n; i++)
        for (uint32_t j : c->frames[i].nbrs)
            if (j >= n || j == i) return fail(c, SFM_E_INVALID, "neighbour index out of range");
    const size_t a0 = std::min<size_t>(c->act_first, n), a1 = std::max(a0, std::min<size_t>(c->act_end, n));
    const DevParams P = make_dev_params(c);
    // the shared-reciprocal divisions need operands in their range, for the keyframe and for its neighbours
    bool shared = !c->verdict_plain;
    for (size_t i = 0; shared && i < n && !div_trust(); i++)
        shared = chain_div_ok(P, c->frames[i].K, c->frames[i].W, c->frames[i].H, c->h_frames[i].Twk, c->h_frames[i].Tkw);
    // arena: per active keyframe info u32[P] | rec 32 B [P] | hits u32 [n_later][P], each 256-byte aligned
    auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
    std::vector<VerdictDev> hv(a1 - a0);
    std::vector<uint64_t> off(a1 - a0);
    std::vector<uint8_t *> later;
    uint64_t total = 0;
    for (size_t i = a0; i < a1; i++) {
        const FrameDev &F = c->h_frames[i];
        VerdictDev &V = hv[i - a0];
        std::memset(&V, 0, sizeof V);
        V.later_begin = (uint32_t)later.size();
        for (uint32_t k = 0; k < F.nbr_count; k++) {
            const uint32_t j = c->h_nbr[F.nbr_begin + k];
            if (j > i) {
                later.push_back(c->h_frames[j].mask);
                V.n_later++;
            }
        }
        const uint64_t px = (uint64_t)F.W * F.H;
        off[i - a0] = total;
        total += up(px * 4) + up(px * sizeof(VerdictRec)) + up(px * 4 * V.n_later);
    }
    // an earlier phase A or chain may still be reading the tables refilled below (the stream does not block the host)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if ((rc = ensure(c, c->d_verdict_arena, total))) return rc;
    if ((rc = ensure(c, c->d_verdicts, hv.size()))) return rc;
    if ((rc = ensure(c, c->d_later_masks, later.size()))) return rc;
    if (!later.empty())
        HIP_TRY(c, hipMemcpy(c->d_later_masks.p, later.data(), later.size() * sizeof(uint8_t *), hipMemcpyHostToDevice));
    for (size_t i = a0; i < a1; i++) {
        const FrameDev &F = c->h_frames[i];
        VerdictDev &V = hv[i - a0];
        const uint64_t px = (uint64_t)F.W * F.H;
        unsigned char *p = c->d_verdict_arena.p + off[i - a0];
        V.info = reinterpret_cast<uint32_t *>(p);
        V.rec = reinterpret_cast<VerdictRec *>(p + up(px * 4));
        V.hits = reinterpret_cast<uint32_t *>(p + up(px * 4) + up(px * sizeof(VerdictRec)));
    }
    if (!hv.empty())
        HIP_TRY(c, hipMemcpy(c->d_verdicts.p, hv.data(), hv.size() * sizeof(VerdictDev), hipMemcpyHostToDevice));
    if (c->key_scan && (rc = reset_key_scan(c))) return rc;  // the scan of a two-phase run happens here
    HIP_TRY(c, hipMemsetAsync(dev_pa_stat(c), 0, sizeof(StatShard) * kStatShards, c->stream));
    // phase A: no order between keyframes; one launch per 32 keyframes keeps the grids moderate
    HIP_TRY(c, hipEventRecord(c->ev_pa[0], c->stream));
    for (size_t g0 = a0; g0 < a1; g0 += kGroup) {
        const size_t g1 = std::min(a1, g0 + kGroup);
        uint64_t max_px = 0;
        for (size_t i = g0; i < g1; i++) max_px = std::max<uint64_t>(max_px, (uint64_t)c->frames[i].W * c->frames[i].H);
        if (!max_px) continue;
        const dim3 grid((uint32_t)((max_px + 255) / 256), (uint32_t)(g1 - g0));
        if (shared)
            hipLaunchKernelGGL(verdict_kernel<true>, grid, dim3(256), 0, c->stream, (const FrameDev *)c->d_frames.p,
                               (const FrameDev *)c->d_nbrpack.p, (const VerdictDev *)c->d_verdicts.p, (uint32_t)g0,
                               (uint32_t)(g0 - a0), P, dev_pa_stat(c));
        else
            hipLaunchKernelGGL(verdict_kernel<false>, grid, dim3(256), 0, c->stream, (const FrameDev *)c->d_frames.p,
                               (const FrameDev *)c->d_nbrpack.p, (const VerdictDev *)c->d_verdicts.p, (uint32_t)g0,
                               (uint32_t)(g0 - a0), P, dev_pa_stat(c));
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_pa[1], c->stream));
    c->pa_timed = true;
    c->have_verdicts = true;  // stream-ordered: sfm_fusion_run's launches follow on the same stream
    return SFM_OK;
}

int sfm_fusion_set_active_range(sfm_fusion_ctx *c, uint32_t first, uint32_t end)
{
    if (!c) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    if (first > end || end > c->frames.size())
        return fail(c, SFM_E_INVALID, "active range must satisfy first <= end <= number of keyframes added");
    c->act_first = first;
    c->act_end = end;
    c->layout_dirty = true;  // pixel totals (sizing, n_px) follow the range; imported masks are dropped with the layout
    c->have_result = c->have_masks = false;
    return SFM_OK;
}

namespace {
// shared argument check of the mask import/export: frames [first, first+count) packed back to back
int mask_span(sfm_fusion_ctx *c, uint32_t first, uint32_t count, uint64_t bytes, const void *ptr)
{
    if ((uint64_t)first + count > c->frames.size()) return fail(c, SFM_E_INVALID, "mask range names unknown keyframes");
    uint64_t need = 0;
    for (uint32_t i = first; i < first + count; i++) need += (uint64_t)c->frames[i].W * c->frames[i].H;
    if (bytes != need) return fail(c, SFM_E_INVALID, "mask buffer size must be the sum of W*H of the keyframes named");
    if (need && !ptr) return fail(c, SFM_E_INVALID, "null mask buffer");
    return SFM_OK;
}
}  // namespace

int sfm_fusion_import_masks(sfm_fusion_ctx *c, uint32_t first, uint32_t count, const void *src_dev, uint64_t bytes)
{
    if (!c) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    int rc;
    if ((rc = mask_span(c, first, count, bytes, src_dev))) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->layout_dirty && (rc = build_layout(c))) return rc;
    hipStream_t s = c->stream;
    if (!c->have_seed) {
        if ((rc = ensure(c, c->d_mask_seed, c->d_masks.n))) return rc;  // (same size for the life of a layout: made once)
        c->seed_frames.clear();
        c->have_seed = true;
    }
    const uint8_t *src = static_cast<const uint8_t *>(src_dev);
    for (uint32_t i = first; i < first + count; i++) {
        const uint64_t px = (uint64_t)c->frames[i].W * c->frames[i].H;
        if (!px) continue;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((px + 1023) / 1024, 2048);
        if (std::find(c->seed_frames.begin(), c->seed_frames.end(), i) == c->seed_frames.end()) {
            c->seed_frames.push_back(i);  // the keyframe's first import: its part of the seed holds nothing yet
            HIP_TRY(c, hipMemcpyAsync(c->d_mask_seed.p + c->mask_off[i], src, px, hipMemcpyDeviceToDevice, s));
        } else {
            hipLaunchKernelGGL(mask_or_kernel, dim3(blocks), dim3(256), 0, s, c->d_mask_seed.p + c->mask_off[i], src, px);
        }
        if (c->masks_ready)  // ... and straight into the arena the next run starts from
            hipLaunchKernelGGL(mask_or_kernel, dim3(blocks), dim3(256), 0, s, c->d_masks.p + c->mask_off[i], src, px);
        src += px;
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(s));  // the caller's buffer is free again on return
    c->have_result = c->have_masks = false;
    return SFM_OK;
}

int sfm_fusion_clear_imported_masks(sfm_fusion_ctx *c)
{
    if (!c) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    c->have_seed = false;
    c->seed_frames.clear();
    c->have_result = c->have_masks = false;
    // The next run starts from zeros (plus what is imported from here on): zero the arena NOW, asynchronously -- a rank
    // of a sequence split calls this before it starts waiting for its predecessors, and the 0.1-0.4 ms that the memset
    // (and, before round 3, a copy of a whole seed arena) took stood between their masks' arrival and its chain.
    c->masks_ready = false;
    if (!c->layout_dirty && c->d_masks.n) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipMemsetAsync(c->d_masks.p, 0, c->d_masks.n, c->stream));
        c->masks_ready = true;
    }
    return SFM_OK;
}

int sfm_fusion_export_masks(sfm_fusion_ctx *c, uint32_t first, uint32_t count, void *dst_dev, uint64_t bytes)
{
    if (!c) return SFM_E_INVALID;
    // (between sfm_fusion_run_chain and sfm_fusion_finish too: the chain has finished, stream A is idle, and the table
    // kernels still running on stream B do not touch the masks)
    if (!c->have_result && !c->have_masks) return fail(c, SFM_E_INVALID, "no result: call sfm_fusion_run first");
    int rc;
    if ((rc = mask_span(c, first, count, bytes, dst_dev))) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    uint8_t *dst = static_cast<uint8_t *>(dst_dev);
    for (uint32_t i = first; i < first + count; i++) {
        const uint64_t px = (uint64_t)c->frames[i].W * c->frames[i].H;
        if (!px) continue;
        HIP_TRY(c, hipMemcpyAsync(dst, c->d_masks.p + c->mask_off[i], px, hipMemcpyDeviceToDevice, c->stream));
        dst += px;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SFM_OK;
}

int sfm_fusion_set_profiling(sfm_fusion_ctx *c, int enable)
{
    if (!c) return SFM_E_INVALID;
    c->profile = enable != 0;
    return SFM_OK;
}

int sfm_fusion_set_global_order(sfm_fusion_ctx *c, const uint32_t *order, uint32_t n)
{
    if (!c) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    if (n != c->frames.size() || (n && !order))
        return fail(c, SFM_E_INVALID, "global order must cover every keyframe");
    for (uint32_t i = 1; i < n; i++)
        if (order[i] <= order[i - 1]) return fail(c, SFM_E_INVALID, "global order must be strictly increasing");
    c->global_order.assign(order, order + n);
    return SFM_OK;
}

int sfm_fusion_get_key_range(sfm_fusion_ctx *c, int64_t lo[3], int64_t hi[3])
{
    if (!c || !lo || !hi) return SFM_E_INVALID;
    for (int a = 0; a < 3; a++) {
        lo[a] = c->key.lo[a];
        hi[a] = c->key.hi[a];
    }
    return SFM_OK;
}

int sfm_fusion_set_key_range(sfm_fusion_ctx *c, const int64_t lo[3], const int64_t hi[3])
{
    if (!c || !lo || !hi) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    KeyRangeDev nk;
    for (int a = 0; a < 3; a++) {
        nk.lo[a] = lo[a];
        nk.hi[a] = hi[a];
    }
    if (const char *why = key_range_problem(nk.lo, nk.hi)) return fail(c, SFM_E_KEY_RANGE, std::string("key range: ") + why);
    if (std::memcmp(&nk, &c->key, sizeof nk) == 0) return SFM_OK;
    if (c->have_result && c->n_out) {  // the cloud at hand keeps its voxels: re-encode its keys
        HIP_TRY(c, hipSetDevice(c->device));
        hipStream_t s = c->stream;
        int rc;
        if ((rc = ensure(c, c->d_rd_keys, c->n_out))) return rc;  // (scratch of sfm_fusion_read_points)
        HIP_TRY(c, hipMemsetAsync(c->d_small, 0, sizeof(Counters), s));
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((c->n_out + 255) / 256, 4096);
        hipLaunchKernelGGL(rekey_kernel, dim3(blocks), dim3(256), 0, s, (const unsigned long long *)c->d_out_key.p,
                           c->d_rd_keys.p, (unsigned long long)c->n_out, key_bits_of(c), key_pack_of(nk), dev_ctr(c));
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(&c->h_small->ctr, c->d_small, sizeof(Counters), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (c->h_small->ctr.err)  // the new ranges do not hold this cloud: nothing was changed
            return fail(c, SFM_E_KEY_RANGE, "the cloud at hand has voxel indices outside the requested ranges");
        HIP_TRY(c, hipMemcpyAsync(c->d_out_key.p, c->d_rd_keys.p, c->n_out * sizeof(unsigned long long),
                                  hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        c->part_counts.clear();
    }
    c->key = nk;
    for (int a = 0; a < 3; a++) c->key_bits[a] = bits_for_range(nk.lo[a], nk.hi[a]);
    c->have_verdicts = false;  // their records hold packed keys
    return SFM_OK;
}

int sfm_fusion_get_cloud_index_range(sfm_fusion_ctx *c, int64_t lo[3], int64_t hi[3])
{
    if (!c || !lo || !hi) return SFM_E_INVALID;
    if (!c->have_result) return fail(c, SFM_E_INVALID, "no result: call sfm_fusion_run first");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = reset_key_scan(c))) return rc;  // (the scan words are free between runs) shard 0 = an empty range
    hipStream_t s = c->stream;
    if (c->n_out) {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((c->n_out + 255) / 256, 1024);
        hipLaunchKernelGGL(key_range_kernel, dim3(blocks), dim3(256), 0, s, (const unsigned long long *)c->d_out_key.p,
                           (unsigned long long)c->n_out, key_bits_of(c), dev_key_scan(c));
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipMemcpyAsync(&c->h_small->key_scan[0], dev_key_scan(c), sizeof(KeyRangeDev), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    for (int a = 0; a < 3; a++) {
        lo[a] = c->h_small->key_scan[0].lo[a];
        hi[a] = c->h_small->key_scan[0].hi[a];
    }
    return SFM_OK;
}

int sfm_fusion_partition_counts(sfm_fusion_ctx *c, uint32_t n_parts, uint64_t *counts)
{
    if (!c || !counts || n_parts == 0 || n_parts > 1024) return fail(c, SFM_E_INVALID, "bad partition argument");
    if (!c->have_result) return fail(c, SFM_E_INVALID, "no result: call sfm_fusion_run first");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->d_part, 2ull * n_parts))) return rc;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemsetAsync(c->d_part.p, 0, 2ull * n_parts * sizeof(unsigned long long), s));
    if (c->n_out) {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((c->n_out + 255) / 256, 1024);
        hipLaunchKernelGGL(partition_count_kernel, dim3(blocks), dim3(256), 0, s, c->d_out_key.p,
                           (unsigned long long)c->n_out, n_parts, key_bits_of(c), c->d_part.p);
        HIP_TRY(c, hipGetLastError());
    }
    c->part_counts.assign(n_parts, 0);
    HIP_TRY(c, hipMemcpyAsync(c->part_counts.data(), c->d_part.p, n_parts * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    std::copy(c->part_counts.begin(), c->part_counts.end(), counts);
    return SFM_OK;
}

// The same counts without any allocation: the counters live in this GPU's copy of the static `scratch` array, the
// read-back goes through a pinned buffer made at sfm_fusion_create.  No hipMalloc / hipFree / std::vector growth on
// this path.  Calls on one GPU share the array: it is cleared here, and calls that overlap on one GPU race.
int sfm_fusion_partition_counts_noalloc(sfm_fusion_ctx *c, uint32_t n_parts, uint64_t *counts)
{
    if (!c || !counts || n_parts == 0) return fail(c, SFM_E_INVALID, "bad partition argument");
    if (n_parts > kScratchParts)
        return fail(c, SFM_E_CAPACITY, "sfm_fusion_partition_counts_noalloc handles at most 1024 parts (static scratch)");
    if (!c->have_result) return fail(c, SFM_E_INVALID, "no result: call sfm_fusion_run first");
    if (!c->scratch || !c->h_part) return fail(c, SFM_E_UNSUPPORTED, "no static scratch on this device");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    unsigned long long *ptr = c->scratch;  // the device address, never the symbol
    HIP_TRY(c, hipMemsetAsync(ptr, 0, (size_t)n_parts * sizeof(unsigned long long), s));  // the last call's counts
    if (c->n_out) {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((c->n_out + 255) / 256, 1024);
        hipLaunchKernelGGL(partition_count_kernel, dim3(blocks), dim3(256), 0, s, c->d_out_key.p,
                           (unsigned long long)c->n_out, n_parts, key_bits_of(c), ptr);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_part, ptr, (size_t)n_parts * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    for (uint32_t p = 0; p < n_parts; p++) counts[p] = c->h_part[p];
    return SFM_OK;
}

int sfm_fusion_export_records(sfm_fusion_ctx *c, void *dst_dev, uint64_t dst_bytes)
{
    if (!c) return SFM_E_INVALID;
    if (!c->have_result || c->part_counts.empty())
        return fail(c, SFM_E_INVALID, "call sfm_fusion_run and sfm_fusion_partition_counts first");
    if (dst_bytes < c->n_out * sizeof(Record) || (c->n_out && !dst_dev))
        return fail(c, SFM_E_INVALID, "export buffer too small");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t n_parts = (uint32_t)c->part_counts.size();
    std::vector<uint64_t> start(n_parts, 0);
    for (uint32_t p = 1; p < n_parts; p++) start[p] = start[p - 1] + c->part_counts[p - 1];
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->d_part.p + n_parts, start.data(), n_parts * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    std::vector<uint32_t> go = c->global_order;
    if (go.size() != c->frames.size()) {
        go.resize(c->frames.size());
        std::iota(go.begin(), go.end(), 0u);
    }
    int rc;
    if ((rc = ensure(c, c->d_gorder, go.size()))) return rc;
    if (!go.empty())
        HIP_TRY(c, hipMemcpyAsync(c->d_gorder.p, go.data(), go.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (c->n_out) {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((c->n_out + 255) / 256, 2048);
        hipLaunchKernelGGL(partition_scatter_kernel, dim3(blocks), dim3(256), 0, s, c->d_out_key.p, c->d_out_tag.p,
                           c->d_out_xyz.p, c->d_out_bgr.p, (unsigned long long)c->n_out, n_parts, key_bits_of(c),
                           (const uint32_t *)c->d_gorder.p, c->d_part.p + n_parts, (Record *)dst_dev);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(s));  // start[] and go[] live on this stack frame
    return SFM_OK;
}

int sfm_fusion_merge(sfm_fusion_ctx *c, const void *records_dev, uint64_t n)
{
    if (!c || (n && !records_dev)) return SFM_E_INVALID;
    if (c->rs.launched) return run_in_flight(c);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    const uint64_t table_cap = pow2_at_least(std::max<uint64_t>(2 * n, 1 << 16));
    if ((rc = ensure(c, c->d_table, table_cap))) return rc;
    if ((rc = ensure(c, c->d_out_key, n))) return rc;
    if ((rc = ensure(c, c->d_out_tag, n))) return rc;
    if ((rc = ensure(c, c->d_out_xyz, 3 * n))) return rc;
    if ((rc = ensure(c, c->d_out_bgr, n))) return rc;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemsetAsync(c->d_table.p, 0xFF, table_cap * sizeof(Slot), s));
    HIP_TRY(c, hipMemsetAsync(c->d_small, 0, sizeof(Counters), s));
    if (n) {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(merge_insert_kernel, dim3(blocks), dim3(256), 0, s, (const Record *)records_dev,
                           (unsigned long long)n, c->d_table.p, table_cap - 1, dev_ctr(c));
        hipLaunchKernelGGL(merge_finalize_kernel, dim3(blocks), dim3(256), 0, s, (const Record *)records_dev,
                           (unsigned long long)n, (const Slot *)c->d_table.p, table_cap - 1, dev_ctr(c),
                           c->d_out_xyz.p, c->d_out_bgr.p, c->d_out_key.p, c->d_out_tag.p);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipMemcpyAsync(&c->h_small->ctr, c->d_small, sizeof(Counters), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (c->h_small->ctr.err) return fail(c, SFM_E_TABLE_FULL, "merge table overflow");
    c->n_out = c->h_small->ctr.out_count;
    c->stats.n_vox = c->n_out;
    c->stats.global_after = c->n_out;
    c->part_counts.clear();
    c->merged = true;
    c->have_result = true;
    return SFM_OK;
}

int sfm_fusion_num_points(sfm_fusion_ctx *c, uint64_t *n)
{
    if (!c || !n) return SFM_E_INVALID;
    if (!c->have_result) return fail(c, SFM_E_INVALID, "no result: call sfm_fusion_run first");
    *n = c->n_out;
    return SFM_OK;
}

int sfm_fusion_read_points(sfm_fusion_ctx *c, double *xyz, int32_t *bgr, uint64_t *key, uint64_t *tag)
{
    if (!c) return SFM_E_INVALID;
    if (!c->have_result) return fail(c, SFM_E_INVALID, "no result: call sfm_fusion_run first");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t n = c->n_out;
    if (n == 0) return SFM_OK;
    if (n > 0xFFFFFFFFull) return fail(c, SFM_E_UNSUPPORTED, "more than 2^32 output points");
    // sort by packed key and bring the cloud into the caller's layout on the device; the host only copies
    int rc;
    if ((rc = ensure(c, c->d_rd_keys, n))) return rc;
    if ((rc = ensure(c, c->d_rd_perm, 2 * n))) return rc;
    if ((rc = ensure(c, c->d_rd_xyz, 3 * n))) return rc;
    if ((rc = ensure(c, c->d_rd_bgr, 3 * n))) return rc;
    if ((rc = ensure(c, c->d_rd_key3, 3 * n))) return rc;
    if ((rc = ensure(c, c->d_rd_tag, n))) return rc;
    hipStream_t s = c->stream;
    const bool remap = !c->merged && !c->global_order.empty() && c->global_order.size() == c->frames.size();
    if (remap) {
        if ((rc = ensure(c, c->d_gorder, c->global_order.size()))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->d_gorder.p, c->global_order.data(), c->global_order.size() * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, s));
    }
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(256), 0, s, c->d_rd_perm.p, (unsigned long long)n);
    HIP_TRY(c, hipGetLastError());
    size_t tmp_bytes = 0;
    HIP_TRY(c, sort_keys_with_index(nullptr, tmp_bytes, c->d_out_key.p, c->d_rd_keys.p, c->d_rd_perm.p,
                                    c->d_rd_perm.p + n, n, s));
    if ((rc = ensure(c, c->d_rd_tmp, tmp_bytes))) return rc;
    HIP_TRY(c, sort_keys_with_index(c->d_rd_tmp.p, tmp_bytes, c->d_out_key.p, c->d_rd_keys.p, c->d_rd_perm.p,
                                    c->d_rd_perm.p + n, n, s));
    hipLaunchKernelGGL(cloud_gather_kernel, dim3(blocks), dim3(256), 0, s, (const uint32_t *)(c->d_rd_perm.p + n),
                       (unsigned long long)n, (const unsigned long long *)c->d_out_key.p,
                       (const unsigned long long *)c->d_out_tag.p, (const double *)c->d_out_xyz.p,
                       (const uint32_t *)c->d_out_bgr.p, (const uint32_t *)c->d_gorder.p,
                       remap ? (uint32_t)c->global_order.size() : 0u, key_bits_of(c), c->d_rd_xyz.p, c->d_rd_bgr.p,
                       c->d_rd_key3.p, c->d_rd_tag.p);
    HIP_TRY(c, hipGetLastError());
    if (xyz) HIP_TRY(c, hipMemcpyAsync(xyz, c->d_rd_xyz.p, n * 24, hipMemcpyDeviceToHost, s));
    if (bgr) HIP_TRY(c, hipMemcpyAsync(bgr, c->d_rd_bgr.p, n * 12, hipMemcpyDeviceToHost, s));
    if (key) HIP_TRY(c, hipMemcpyAsync(key, c->d_rd_key3.p, n * 24, hipMemcpyDeviceToHost, s));
    if (tag) HIP_TRY(c, hipMemcpyAsync(tag, c->d_rd_tag.p, n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return SFM_OK;
}

int sfm_fusion_get_stats(sfm_fusion_ctx *c, sfm_fusion_stats *out)
{
    if (!c || !out) return SFM_E_INVALID;
    if (!c->have_result) return fail(c, SFM_E_INVALID, "no result: call sfm_fusion_run first");
    *out = c->stats;
    return SFM_OK;
}

int sfm_fusion_get_frame_counts(sfm_fusion_ctx *c, uint64_t *before, uint64_t *after)
{
    if (!c) return SFM_E_INVALID;
    if (!c->have_result) return fail(c, SFM_E_INVALID, "no result: call sfm_fusion_run first");
    if (before) std::copy(c->fbefore.begin(), c->fbefore.end(), before);
    if (after) std::copy(c->fafter.begin(), c->fafter.end(), after);
    return SFM_OK;
}

const char *sfm_fusion_last_error(sfm_fusion_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

const char *sfm_fusion_version(void)
{
#if SFM_DIV_CHECKED
    return "sfm_fusion 0.3 (gfx950) div-checked";
#else
    return "sfm_fusion 0.3 (gfx950)";
#endif
}

}  // extern "C"
