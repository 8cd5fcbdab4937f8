// dist_fusion.hpp -- C++ host of the multi-GPU path: one process per GPU, RCCL over xGMI, the C ABI underneath.
// The C++ twin of sfm_amd/dist_fusion.py (SequenceSplit + VoxelExchange); no reference counterpart (the reference is
// one process).  One keyframe sequence is cut into contiguous blocks, one per rank:
//   1. every rank but the first computes its block's mask-independent verdicts (sfm_fusion_precompute) while its
//      predecessors chain;
//   2. a rank receives the visibility masks its predecessors left on its keyframes (ncclRecv, rank order), fuses its
//      block (sfm_fusion_run_chain), and sends the masks of later ranks' keyframes on (ncclSend);
//   3. voxel exchange: winners bucketed by owner = hash(voxel indices) mod N, counts and key ranges by ncclAllGather,
//      48-byte records by grouped ncclSend/ncclRecv (every peer pair on its own xGMI link), sfm_fusion_merge.
// The union of the ranks' clouds is bit-identical to DenseReconstruction::fuse() on one GPU.
// Header-only; build with -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ and link -lsfm_fusion -lrccl -lamdhip64.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "dense_reconstruction.hpp"

namespace sfm_amd {

struct RankPlan {
    std::size_t first = 0, end = 0;                         // block [first, end) of the selected-keyframe order
    std::vector<std::size_t> frames;                        // ascending: the block and every neighbour of a keyframe in it
    std::map<int, std::vector<std::size_t>> sends, recvs;   // rank -> keyframes whose masks travel
    std::map<std::size_t, std::uint32_t> local;             // keyframe -> ctx order index
};

// Relative cost of a keyframe (per pixel) as rank 0's fused chain, as a later rank's phase A (verdicts) and as its
// phase B (accept chain), measured on one MI355X.  Only the ratios matter, and only for speed.
struct SplitCost {
    // microseconds per C3 keyframe on one MI355X: fused chain, phase A, phase B -- both chains as
    // sfm_fusion_run_chain runs them, the voxel filter behind the chain -- and that voxel filter alone on the device
    double fused = 34.6, verdict = 44.5, accept = 9.3, filter = 9.3;
    bool equal_blocks = false;
};

// The pipeline model behind block_shares: rank 0 chains from time 0, rank r > 0 computes its verdicts from time 0 and
// chains when they are done and its predecessor has stopped chaining; a rank's voxel filter runs behind its own chain.
inline double modelled_makespan(const std::vector<double> &shares, const SplitCost &c, double units = 1.0)
{
    double end = 0, worst = 0;
    for (std::size_t r = 0; r < shares.size(); r++) {
        const double n = shares[r] * units;
        const double start = r == 0 ? end : std::max(end, c.verdict * n);
        end = start + (r == 0 ? c.fused : c.accept) * n;
        worst = std::max(worst, end + c.filter * n);
    }
    return worst;
}

// Share of the sequence each rank's block gets: every rank's phase A ends just when its predecessor's chain does
// (rank 1's block is fused/verdict of rank 0's, every later one (1 + accept/verdict) times the one before).  The voxel
// filter behind a rank's chain is hidden under its successors' chains on every rank but the last: the last block is
// shortened (the others keep their ratios) as far as the model gains from it.
inline std::vector<double> block_shares(int world, const SplitCost &c = SplitCost{})
{
    std::vector<double> s((std::size_t)world, 1.0);
    if (!c.equal_blocks)
        for (int r = 1; r < world; r++) s[r] = r == 1 ? c.fused / c.verdict : s[r - 1] * (1.0 + c.accept / c.verdict);
    auto normalise = [](std::vector<double> &v) {
        double total = 0;
        for (double x : v) total += x;
        for (double &x : v) x /= total;
    };
    normalise(s);
    if (!c.equal_blocks && c.filter > 0 && world > 1) {
        std::vector<double> best = s;
        double best_t = modelled_makespan(s, c);
        for (int k = 1; k < 40; k++) {
            std::vector<double> cand = s;
            cand.back() *= 1.0 - k / 40.0;
            normalise(cand);
            const double t = modelled_makespan(cand, c);
            if (t < best_t) { best = cand; best_t = t; }
        }
        s = best;
    }
    return s;
}

// nbrs[i] = neighbours of keyframe i (positions in the processing order); px[i] = its pixels (the block weight)
inline std::vector<RankPlan> plan_sequence(const std::vector<std::vector<std::uint32_t>> &nbrs,
                                           const std::vector<std::uint64_t> &px, int world,
                                           const SplitCost &cost = SplitCost{})
{
    const std::size_t n = nbrs.size();
    std::uint64_t total = 0;
    for (auto w : px) total += w;
    const std::vector<double> shares = block_shares(world, cost);
    std::vector<std::size_t> cuts{0};
    std::uint64_t acc = 0;
    std::size_t i = 0;
    double upto = 0;
    for (int r = 1; r < world; r++) {
        upto += shares[r - 1];
        const double want = (double)total * upto;
        while (i < n && (double)acc + px[i] / 2.0 <= want) acc += px[i++];
        cuts.push_back(i);
    }
    cuts.push_back(n);
    std::vector<int> owner(n, 0);
    for (int r = 0; r < world; r++)
        for (std::size_t g = cuts[r]; g < cuts[r + 1]; g++) owner[g] = r;
    std::vector<RankPlan> plans(world);
    std::vector<std::map<int, std::set<std::size_t>>> marks(world);
    for (int r = 0; r < world; r++) {
        RankPlan &p = plans[r];
        p.first = cuts[r];
        p.end = cuts[r + 1];
        std::set<std::size_t> fr;
        for (std::size_t g = p.first; g < p.end; g++) {
            fr.insert(g);
            for (auto j : nbrs[g]) {
                fr.insert(j);
                if (j >= p.end) marks[r][owner[j]].insert(j);  // a keyframe marks its LATER neighbours only
            }
        }
        p.frames.assign(fr.begin(), fr.end());
        for (std::size_t k = 0; k < p.frames.size(); k++) p.local[p.frames[k]] = (std::uint32_t)k;
    }
    for (int q = 0; q < world; q++)
        for (auto &[t, s] : marks[q]) {
            plans[q].sends[t].assign(s.begin(), s.end());
            plans[t].recvs[q].assign(s.begin(), s.end());
        }
    return plans;
}

class DistributedDenseReconstruction {
public:
    DistributedDenseReconstruction(const Map &map, int rank, int world, ncclComm_t comm, int device)
        : map_(map), rank_(rank), world_(world), comm_(comm), device_(device)
    {
        sfm_fusion_default_params(&params_);
    }
    sfm_fusion_params &params() { return params_; }
    const sfm_fusion_stats &stats() const { return stats_; }

    // This rank's share of the cloud (the voxels it owns), sorted by voxel key; tags carry positions in the
    // selected-keyframe order, as DenseReconstruction::fuse() gives them.
    Cloud fuse()
    {
        hip(hipSetDevice(device_), "hipSetDevice");
        hipStream_t stream = nullptr;
        hip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
        // selection and neighbours exactly as the one-GPU facade (dense_reconstruction.cpp:32-38, map.cpp:152-160)
        std::set<std::size_t> sel;
        for (const auto &[e, count] : map_.covisibility_edge)
            if (count >= 100) { sel.insert(e.first); sel.insert(e.second); }
        const std::vector<std::size_t> order(sel.begin(), sel.end());
        std::map<std::size_t, std::uint32_t> pos;
        for (std::size_t i = 0; i < order.size(); i++) pos[order[i]] = (std::uint32_t)i;
        std::vector<std::vector<std::uint32_t>> nb(order.size());
        for (const auto &[e, count] : map_.covisibility_edge) {
            if (count <= 100) continue;
            nb[pos.at(e.first)].push_back(pos.at(e.second));
            nb[pos.at(e.second)].push_back(pos.at(e.first));
        }
        std::vector<std::uint64_t> px(order.size());
        for (std::size_t i = 0; i < order.size(); i++) {
            std::sort(nb[i].begin(), nb[i].end());
            const KeyFrame &k = map_.keyframes.at(order[i]);
            px[i] = (std::uint64_t)k.width * k.height;
        }
        const RankPlan plan = plan_sequence(nb, px, world_)[rank_];

        sfm_fusion_ctx *ctx = nullptr;
        struct Guard {  // the ctx and its stream go on every way out, exceptions included
            sfm_fusion_ctx *&ctx;
            hipStream_t &stream;
            ~Guard()
            {
                if (ctx) sfm_fusion_destroy(ctx);
                if (stream) (void)hipStreamDestroy(stream);
            }
        } guard{ctx, stream};
        if (sfm_fusion_create(&params_, device_, stream, &ctx) != SFM_OK)
            throw std::runtime_error("sfm_fusion_create failed (no HIP device?)");
        auto check = [&](int rc) {
            if (rc != SFM_OK) throw std::runtime_error(std::string("sfm_fusion: ") + sfm_fusion_last_error(ctx));
        };
        std::vector<std::uint32_t> global_order;
        for (std::size_t g : plan.frames) {
            const KeyFrame &k = map_.keyframes.at(order[g]);
            const double K[4] = {k.K.fx, k.K.fy, k.K.cx, k.K.cy};
            check(sfm_fusion_add_keyframe(ctx, plan.local.at(g), k.width, k.height, k.depth, k.depth_step, k.img, k.img_step, K,
                                          k.T_wk, k.T_kw, SFM_MEM_HOST));
            global_order.push_back((std::uint32_t)g);
        }
        for (std::size_t g = plan.first; g < plan.end; g++) {
            std::vector<std::uint32_t> loc;
            for (auto j : nb[g]) loc.push_back(plan.local.at(j));
            check(sfm_fusion_set_neighbors(ctx, plan.local.at(g), loc.data(), (std::uint32_t)loc.size()));
        }
        check(sfm_fusion_set_global_order(ctx, global_order.data(), (std::uint32_t)global_order.size()));
        const std::uint32_t a0 = plan.first < plan.end ? plan.local.at(plan.first) : 0;
        check(sfm_fusion_set_active_range(ctx, a0, a0 + (std::uint32_t)(plan.end - plan.first)));

        // every mask buffer of the plan is made before anything runs: nothing is allocated between a predecessor's
        // send and this rank's chain, or between the chain and the sends that the next rank waits for
        auto bytes_of = [&](const std::vector<std::size_t> &fr) {
            std::uint64_t b = 0;
            for (auto g : fr) b += px[g];
            return b;
        };
        std::map<int, std::unique_ptr<DevBuf>> recv_buf, send_buf;
        for (const auto &[q, fr] : plan.recvs) recv_buf[q] = std::make_unique<DevBuf>(bytes_of(fr));
        for (const auto &[t, fr] : plan.sends) send_buf[t] = std::make_unique<DevBuf>(bytes_of(fr));
        // 1. phase A while the predecessors chain (the first rank has nobody to wait for: fused chain)
        if (rank_ > 0 && plan.end > plan.first) check(sfm_fusion_precompute(ctx));
        // 2. masks in (ascending ranks: the order the senders finish in), this block's chain, masks out
        for (const auto &[q, fr] : plan.recvs) {
            DevBuf &buf = *recv_buf[q];
            nccl(ncclRecv(buf.p, buf.n, ncclUint8, q, comm_, stream), "ncclRecv masks");
            hip(hipStreamSynchronize(stream), "sync");
            std::uint64_t off = 0;
            for_runs(plan, fr, [&](std::uint32_t first, std::uint32_t count, std::uint64_t nbytes) {
                check(sfm_fusion_import_masks(ctx, first, count, (const char *)buf.p + off, nbytes));
                off += nbytes;
            }, px);
        }
        // a rank with successors hands its masks on as soon as the frame loop has finished; the voxel filter of its last
        // keyframe groups runs on (second stream of the ctx) under the sends and is collected by sfm_fusion_finish
        const bool split = !plan.sends.empty();
        check(split ? sfm_fusion_run_chain(ctx) : sfm_fusion_run(ctx));
        for (const auto &[t, fr] : plan.sends) {
            std::uint64_t off = 0;
            for_runs(plan, fr, [&](std::uint32_t first, std::uint32_t count, std::uint64_t nbytes) {
                check(sfm_fusion_export_masks(ctx, first, count, (char *)send_buf[t]->p + off, nbytes));
                off += nbytes;
            }, px);
        }
        if (!plan.sends.empty()) {  // all outgoing masks as one group: every peer on its own xGMI link at once
            nccl(ncclGroupStart(), "ncclGroupStart");
            for (const auto &[t, fr] : plan.sends)
                nccl(ncclSend(send_buf[t]->p, send_buf[t]->n, ncclUint8, t, comm_, stream), "ncclSend masks");
            nccl(ncclGroupEnd(), "ncclGroupEnd");
            hip(hipStreamSynchronize(stream), "sync");
        }
        if (split) check(sfm_fusion_finish(ctx));
        // 3. voxel exchange
        std::vector<std::uint64_t> counts(world_);
        check(sfm_fusion_partition_counts(ctx, (std::uint32_t)world_, counts.data()));
        const int row = world_ + 6;  // my record count for every owner, then my key packing (lo[3], hi[3])
        std::vector<std::int64_t> mine(row), all((std::size_t)row * world_);
        for (int p = 0; p < world_; p++) mine[p] = (std::int64_t)counts[p];
        check(sfm_fusion_get_key_range(ctx, &mine[world_], &mine[world_ + 3]));
        auto all_gather = [&](const std::vector<std::int64_t> &in, std::vector<std::int64_t> &out) {
            DevBuf s(in.size() * 8), r((std::uint64_t)in.size() * 8 * world_);
            hip(hipMemcpyAsync(s.p, in.data(), in.size() * 8, hipMemcpyHostToDevice, stream), "H2D");
            nccl(ncclAllGather(s.p, r.p, in.size(), ncclInt64, comm_, stream), "ncclAllGather");
            hip(hipMemcpyAsync(out.data(), r.p, in.size() * 8 * world_, hipMemcpyDeviceToHost, stream), "D2H");
            hip(hipStreamSynchronize(stream), "sync");
        };
        all_gather(mine, all);
        bool differ = false;
        for (int p = 0; p < world_ && !differ; p++)
            for (int k = 0; k < 6; k++) differ = differ || all[(std::size_t)p * row + world_ + k] != mine[world_ + k];
        if (differ) {
            // One packing on every rank: the hull of the index ranges the CLOUDS really have -- not of the packings (a
            // rank still on the default packing holds +-(2^20 - 1) per axis whatever its cloud is, and the hull of that
            // with a scanned range far from the origin can need more than 63 bits where the clouds' own hull does not).
            std::vector<std::int64_t> real(6), reals((std::size_t)6 * world_);
            check(sfm_fusion_get_cloud_index_range(ctx, &real[0], &real[3]));
            all_gather(real, reals);
            std::int64_t lo[3], hi[3];
            for (int a = 0; a < 3; a++) {
                bool any = false;
                lo[a] = hi[a] = 0;
                for (int p = 0; p < world_; p++) {
                    const std::int64_t l = reals[(std::size_t)p * 6 + a], h = reals[(std::size_t)p * 6 + 3 + a];
                    if (l > h) continue;  // no index on this axis in that rank's cloud
                    lo[a] = any ? std::min(lo[a], l) : l;
                    hi[a] = any ? std::max(hi[a], h) : h;
                    any = true;
                }
            }
            check(sfm_fusion_set_key_range(ctx, lo, hi));  // re-encodes this rank's cloud
            check(sfm_fusion_partition_counts(ctx, (std::uint32_t)world_, counts.data()));  // (arms export_records again)
        }
        std::uint64_t n_local = 0, n_recv = 0;
        for (auto c : counts) n_local += c;
        for (int p = 0; p < world_; p++) n_recv += (std::uint64_t)all[(std::size_t)p * row + rank_];
        DevBuf send(n_local * SFM_RECORD_BYTES), recv(n_recv * SFM_RECORD_BYTES);
        check(sfm_fusion_export_records(ctx, send.p, n_local * SFM_RECORD_BYTES));
        nccl(ncclGroupStart(), "ncclGroupStart");
        std::uint64_t so = 0, ro = 0;
        for (int p = 0; p < world_; p++) {
            const std::uint64_t sb = counts[p] * SFM_RECORD_BYTES, rb = (std::uint64_t)all[(std::size_t)p * row + rank_] * SFM_RECORD_BYTES;
            if (sb) nccl(ncclSend((char *)send.p + so, sb, ncclUint8, p, comm_, stream), "ncclSend records");
            if (rb) nccl(ncclRecv((char *)recv.p + ro, rb, ncclUint8, p, comm_, stream), "ncclRecv records");
            so += sb;
            ro += rb;
        }
        nccl(ncclGroupEnd(), "ncclGroupEnd");
        hip(hipStreamSynchronize(stream), "sync");
        check(sfm_fusion_merge(ctx, recv.p, n_recv));
        std::uint64_t n = 0;
        check(sfm_fusion_num_points(ctx, &n));
        Cloud c;
        c.xyz.resize(3 * n); c.bgr.resize(3 * n); c.key.resize(3 * n); c.tag.resize(n);
        check(sfm_fusion_read_points(ctx, c.xyz.data(), c.bgr.data(), c.key.data(), c.tag.data()));
        check(sfm_fusion_get_stats(ctx, &stats_));
        return c;
    }

private:
    struct DevBuf {
        void *p = nullptr;
        std::uint64_t n;
        explicit DevBuf(std::uint64_t bytes) : n(bytes)
        {
            if (hipMalloc(&p, std::max<std::uint64_t>(bytes, 16)) != hipSuccess) throw std::runtime_error("hipMalloc");
        }
        ~DevBuf() { (void)hipFree(p); }
        DevBuf(const DevBuf &) = delete;
    };
    static void hip(hipError_t e, const char *what)
    {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }
    static void nccl(ncclResult_t e, const char *what)
    {
        if (e != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(e));
    }
    // consecutive runs of ctx order indices among the keyframes `fr`
    template <typename F>
    static void for_runs(const RankPlan &plan, const std::vector<std::size_t> &fr, F &&f, const std::vector<std::uint64_t> &px)
    {
        std::size_t i = 0;
        while (i < fr.size()) {
            const std::uint32_t first = plan.local.at(fr[i]);
            std::uint32_t count = 0;
            std::uint64_t nbytes = 0;
            while (i < fr.size() && plan.local.at(fr[i]) == first + count) {
                nbytes += px[fr[i]];
                count++;
                i++;
            }
            f(first, count, nbytes);
        }
    }

    const Map &map_;
    int rank_, world_;
    ncclComm_t comm_;
    int device_;
    sfm_fusion_params params_{};
    sfm_fusion_stats stats_{};
};

}  // namespace sfm_amd
