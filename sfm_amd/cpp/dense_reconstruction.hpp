// dense_reconstruction.hpp -- C++17 facade over the C ABI (include/sfm_fusion.h) with the
// reference's two-call shape:
//
//     sfm_amd::DenseReconstruction dense{map};      // reference: src/reconstruct.cpp:72
//     dense.reconstruct(dense_file);                // reference: src/reconstruct.cpp:73
//
// `Map` / `KeyFrame` are POD stand-ins for the parts of sfm::Map / sfm::KeyFrame the fusion
// reads (include/reconstruction/map.hpp:140-151, keyframe.hpp:147-178); a build of the
// reference fills them with the adapter shown in INTEGRATION.md.  Header-only; link with
// -lsfm_fusion.  The facade keeps the host-side rules of reconstruct(): keyframe selection
// (count >= 100, src/dense_reconstruction.cpp:32-38), neighbours (count > 100,
// src/map.cpp:152-160) and the text writer (:61-64).
#pragma once
#include <algorithm>
#include <cstdint>
#include <iostream>
#include <map>
#include "cloud_writer.hpp"
#include <ostream>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sfm_fusion.h"

namespace sfm_amd {

struct PinholeModel {  // include/reconstruction/pinhole.hpp:15-28
    double fx, fy, cx, cy;
};

struct KeyFrame {
    int width = 0, height = 0;
    const std::uint16_t *depth = nullptr;  // KeyFrame::get_depth_image(): CV_16UC1
    std::size_t depth_step = 0;            // cv::Mat::step, bytes
    const std::uint8_t *img = nullptr;     // KeyFrame::img: CV_8UC3, B,G,R
    std::size_t img_step = 0;
    PinholeModel K{};                      // camera_calibration()
    double T_wk[16]{};                     // transform().data()        (column-major)
    double T_kw[16]{};                     // world_to_camera().data()
};

struct Map {
    std::vector<KeyFrame> keyframes;  // insertion order = canonical processing order
    // (i, j) with i < j -> shared map-point count, frozen at first insertion (src/map.cpp:108-119)
    std::map<std::pair<std::size_t, std::size_t>, std::size_t> covisibility_edge;

    void insert_edge(std::size_t a, std::size_t b, std::size_t count)
    {
        if (a == b) return;
        covisibility_edge.emplace(std::make_pair(std::min(a, b), std::max(a, b)), count);
    }
    // strictly more than min_shared_features (src/map.cpp:157); ascending index
    std::vector<std::size_t> get_neighbors(std::size_t kf, std::size_t min_shared_features) const
    {
        std::vector<std::size_t> out;
        for (const auto &[e, count] : covisibility_edge) {
            if (count <= min_shared_features) continue;
            if (e.first == kf) out.push_back(e.second);
            else if (e.second == kf) out.push_back(e.first);
        }
        std::sort(out.begin(), out.end());
        return out;
    }
};

struct Cloud {
    std::vector<double> xyz;         // N*3
    std::vector<std::int32_t> bgr;   // N*3, B,G,R like PointInfo::color
    std::vector<std::uint64_t> key;  // N*3
    std::vector<std::uint64_t> tag;  // N
    std::size_t size() const { return tag.size(); }
};

class DenseReconstruction {
public:
    explicit DenseReconstruction(const Map &map, int device = 0) : map_(map), device_(device)
    {
        sfm_fusion_default_params(&params_);
    }
    sfm_fusion_params &params() { return params_; }

    // Selection + upload + GPU fusion; cloud sorted by voxel key.
    Cloud fuse()
    {
        std::set<std::size_t> sel;  // dense_reconstruction.cpp:32-38
        for (const auto &[e, count] : map_.covisibility_edge)
            if (count >= 100) { sel.insert(e.first); sel.insert(e.second); }
        std::vector<std::size_t> order(sel.begin(), sel.end());
        std::map<std::size_t, std::uint32_t> pos;
        for (std::size_t i = 0; i < order.size(); i++) pos[order[i]] = (std::uint32_t)i;

        sfm_fusion_ctx *ctx = nullptr;
        if (sfm_fusion_create(&params_, device_, nullptr, &ctx) != SFM_OK)
            throw std::runtime_error("sfm_fusion_create failed (no HIP device?)");
        auto check = [&](int rc) {
            if (rc != SFM_OK) {
                std::string msg = sfm_fusion_last_error(ctx);
                sfm_fusion_destroy(ctx);
                throw std::runtime_error("sfm_fusion: " + msg);
            }
        };
        for (std::size_t i = 0; i < order.size(); i++) {
            const KeyFrame &k = map_.keyframes.at(order[i]);
            const double K[4] = {k.K.fx, k.K.fy, k.K.cx, k.K.cy};
            check(sfm_fusion_add_keyframe(ctx, (std::uint32_t)i, k.width, k.height, k.depth, k.depth_step,
                                          k.img, k.img_step, K, k.T_wk, k.T_kw, SFM_MEM_HOST));
        }
        // process_frame: map.get_neighbors(frame, 100) (:70) for every selected keyframe, in one
        // pass over the edges (count > 100 implies both endpoints are selected); ascending order
        std::vector<std::vector<std::uint32_t>> nb(order.size());
        for (const auto &[e, count] : map_.covisibility_edge) {
            if (count <= 100) continue;
            nb[pos.at(e.first)].push_back(pos.at(e.second));
            nb[pos.at(e.second)].push_back(pos.at(e.first));
        }
        for (std::size_t i = 0; i < order.size(); i++) {
            std::sort(nb[i].begin(), nb[i].end());
            check(sfm_fusion_set_neighbors(ctx, (std::uint32_t)i, nb[i].data(), (std::uint32_t)nb[i].size()));
        }
        check(sfm_fusion_run(ctx));
        std::uint64_t n = 0;
        check(sfm_fusion_num_points(ctx, &n));
        Cloud c;
        c.xyz.resize(3 * n); c.bgr.resize(3 * n); c.key.resize(3 * n); c.tag.resize(n);
        check(sfm_fusion_read_points(ctx, c.xyz.data(), c.bgr.data(), c.key.data(), c.tag.data()));
        check(sfm_fusion_get_stats(ctx, &stats_));
        frame_before_.assign(order.size(), 0);
        frame_after_.assign(order.size(), 0);
        check(sfm_fusion_get_frame_counts(ctx, frame_before_.data(), frame_after_.data()));
        sfm_fusion_destroy(ctx);
        return c;
    }

    // The reference's progress log (dense_reconstruction.cpp:49, :54-58, :118-122): same lines, same order, written
    // after the fusion (the GPU fuses all keyframes in one call).  On by default, to std::cout like the reference;
    // set_progress_log(nullptr) silences it, any other stream redirects it.
    void set_progress_log(std::ostream *log) { log_ = log; }
    void write_progress_log(std::ostream &log) const
    {
        const std::size_t n = frame_before_.size();
        for (std::size_t i = 0; i < n; i++) {
            log << "Processing keyframe " << i << "/" << n << std::endl;           // :49
            log << "Points before voxel filter: " << frame_before_[i] << std::endl;  // :118
            log << "Points after voxel filter: " << frame_after_[i] << std::endl;    // :120
            log << "\n";                                                           // :122
        }
        log << "processed all map frames" << std::endl;                              // :54
        log << "Points before voxel removal: " << stats_.global_before << std::endl;  // :56
        log << "Points after voxel removal: " << stats_.global_after << std::endl;    // :58
    }

    // dense_reconstruction.cpp:61-64: "x y z R G B" with default ostream formatting
    void reconstruct(std::ostream &os)
    {
        const Cloud c = fuse();
        if (log_) write_progress_log(*log_);
        write_text_fast(os, c.xyz.data(), c.bgr.data(), c.size());  // same bytes as write_text (cloud_writer.hpp)
    }
    // binary PLY instead of text (no reference counterpart)
    void reconstruct_ply(std::ostream &os)
    {
        const Cloud c = fuse();
        if (log_) write_progress_log(*log_);
        write_ply_binary(os, c.xyz.data(), c.bgr.data(), c.size());
    }
    const sfm_fusion_stats &stats() const { return stats_; }
    // per selected keyframe: "Points before / after voxel filter" (:118-120)
    const std::vector<std::uint64_t> &frame_before() const { return frame_before_; }
    const std::vector<std::uint64_t> &frame_after() const { return frame_after_; }

private:
    const Map &map_;
    int device_;
    sfm_fusion_params params_{};
    sfm_fusion_stats stats_{};
    std::vector<std::uint64_t> frame_before_, frame_after_;
    std::ostream *log_ = &std::cout;
};

}  // namespace sfm_amd
