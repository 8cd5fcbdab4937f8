// replay.hpp -- keyframe replay dump; the C++ twin of sfm_amd/replay.py, where
// the layout is specified.  A dump holds what DenseReconstruction::reconstruct reads from the
// reference's Map after bundle adjustment: per keyframe img / depth / K / T_wk / T_kw
// (src/keyframe.cpp:31-54) and the covisibility edges with their counts (map.hpp:143).
//
//   sfm_amd::write_replay(path, map)   from an sfm_amd::Map (plain pointers into caller memory)
//   sfm_amd::Replay r(path); r.map     memory-mapped; keyframes point into the mapping
#pragma once
#include "dense_reconstruction.hpp"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <stdexcept>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace sfm_amd {

namespace replay_detail {
constexpr char kMagic[9] = "SFMREPL1";
constexpr std::uint32_t kVersion = 1;
#pragma pack(push, 1)
struct Header {
    char magic[8];
    std::uint32_t version, n_keyframes, n_edges, reserved;
    std::uint64_t kf_off, edge_off, file_size;
};
struct KfRecord {
    std::uint32_t W, H;
    double K[4], T_wk[16], T_kw[16];
    std::uint64_t depth_off, bgr_off, reserved;
};
struct EdgeRecord {
    std::uint32_t a, b, count;
};
#pragma pack(pop)
static_assert(sizeof(Header) == 48 && sizeof(KfRecord) == 320 && sizeof(EdgeRecord) == 12, "replay layout");
inline std::uint64_t align64(std::uint64_t n) { return (n + 63) / 64 * 64; }
}  // namespace replay_detail

inline void write_replay(const std::string &path, const Map &map)
{
    using namespace replay_detail;
    const std::size_t n = map.keyframes.size();
    Header h{};
    std::memcpy(h.magic, kMagic, 8);
    h.version = kVersion;
    h.n_keyframes = (std::uint32_t)n;
    h.n_edges = (std::uint32_t)map.covisibility_edge.size();
    h.kf_off = 64;
    h.edge_off = align64(h.kf_off + sizeof(KfRecord) * n);
    std::uint64_t pos = align64(h.edge_off + sizeof(EdgeRecord) * h.n_edges);
    std::vector<KfRecord> recs(n);
    for (std::size_t i = 0; i < n; i++) {
        const KeyFrame &kf = map.keyframes[i];
        KfRecord &r = recs[i];
        std::memset(&r, 0, sizeof r);
        r.W = (std::uint32_t)kf.width;
        r.H = (std::uint32_t)kf.height;
        r.K[0] = kf.K.fx; r.K[1] = kf.K.fy; r.K[2] = kf.K.cx; r.K[3] = kf.K.cy;
        std::memcpy(r.T_wk, kf.T_wk, sizeof r.T_wk);
        std::memcpy(r.T_kw, kf.T_kw, sizeof r.T_kw);
        r.depth_off = pos;
        pos = align64(pos + 2ull * r.W * r.H);
        r.bgr_off = pos;
        pos = align64(pos + 3ull * r.W * r.H);
    }
    h.file_size = pos;
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path);
    auto put = [&](std::uint64_t off, const void *p, std::size_t bytes) {
        if (std::fseek(f, (long)off, SEEK_SET) != 0 || (bytes && std::fwrite(p, 1, bytes, f) != bytes)) {
            std::fclose(f);
            throw std::runtime_error("write error on " + path);
        }
    };
    put(0, &h, sizeof h);
    put(h.kf_off, recs.data(), sizeof(KfRecord) * n);
    std::vector<EdgeRecord> edges;
    for (const auto &e : map.covisibility_edge)
        edges.push_back({(std::uint32_t)e.first.first, (std::uint32_t)e.first.second, (std::uint32_t)e.second});
    put(h.edge_off, edges.data(), sizeof(EdgeRecord) * edges.size());
    for (std::size_t i = 0; i < n; i++) {
        const KeyFrame &kf = map.keyframes[i];
        for (std::uint32_t y = 0; y < recs[i].H; y++) {  // rows may be strided in the source (cv::Mat::step)
            put(recs[i].depth_off + 2ull * recs[i].W * y,
                reinterpret_cast<const unsigned char *>(kf.depth) + kf.depth_step * y, 2ull * recs[i].W);
            put(recs[i].bgr_off + 3ull * recs[i].W * y, kf.img + kf.img_step * y, 3ull * recs[i].W);
        }
    }
    if (pos) {
        const unsigned char zero = 0;  // the file ends on its last (aligned) byte
        std::fseek(f, 0, SEEK_END);
        if ((std::uint64_t)std::ftell(f) < pos) put(pos - 1, &zero, 1);
    }
    std::fclose(f);
}

class Replay {
public:
    Map map;

    explicit Replay(const std::string &path)
    {
        using namespace replay_detail;
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (::fstat(fd_, &st) != 0 || st.st_size < 64) fail(path + ": not a keyframe replay dump");
        size_ = (std::size_t)st.st_size;
        base_ = static_cast<const unsigned char *>(::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0));
        if (base_ == MAP_FAILED) { base_ = nullptr; fail("mmap failed for " + path); }
        Header h;
        std::memcpy(&h, base_, sizeof h);
        if (std::memcmp(h.magic, kMagic, 8) != 0) fail(path + ": not a keyframe replay dump");
        if (h.version != kVersion) fail(path + ": unsupported replay version");
        if (h.file_size != size_ || h.kf_off + sizeof(KfRecord) * (std::uint64_t)h.n_keyframes > size_ ||
            h.edge_off + sizeof(EdgeRecord) * (std::uint64_t)h.n_edges > size_)
            fail(path + ": truncated replay dump");
        map.keyframes.resize(h.n_keyframes);
        for (std::uint32_t i = 0; i < h.n_keyframes; i++) {
            KfRecord r;
            std::memcpy(&r, base_ + h.kf_off + sizeof(KfRecord) * i, sizeof r);
            const std::uint64_t px = (std::uint64_t)r.W * r.H;
            if (r.depth_off + 2 * px > size_ || r.bgr_off + 3 * px > size_)
                fail(path + ": keyframe " + std::to_string(i) + " points outside the file");
            KeyFrame &kf = map.keyframes[i];
            kf.width = (int)r.W;
            kf.height = (int)r.H;
            kf.depth = reinterpret_cast<const std::uint16_t *>(base_ + r.depth_off);
            kf.depth_step = 2ull * r.W;
            kf.img = base_ + r.bgr_off;
            kf.img_step = 3ull * r.W;
            kf.K = PinholeModel{r.K[0], r.K[1], r.K[2], r.K[3]};
            std::memcpy(kf.T_wk, r.T_wk, sizeof kf.T_wk);
            std::memcpy(kf.T_kw, r.T_kw, sizeof kf.T_kw);
        }
        for (std::uint32_t i = 0; i < h.n_edges; i++) {
            EdgeRecord e;
            std::memcpy(&e, base_ + h.edge_off + sizeof(EdgeRecord) * i, sizeof e);
            if (e.a >= h.n_keyframes || e.b >= h.n_keyframes) fail(path + ": edge names an unknown keyframe");
            map.insert_edge(e.a, e.b, e.count);
        }
    }
    Replay(const Replay &) = delete;
    Replay &operator=(const Replay &) = delete;
    ~Replay() { release(); }

private:
    void release()
    {
        if (base_) ::munmap(const_cast<unsigned char *>(base_), size_);
        if (fd_ >= 0) ::close(fd_);
        base_ = nullptr;
        fd_ = -1;
    }
    [[noreturn]] void fail(const std::string &msg)
    {
        release();
        throw std::runtime_error(msg);
    }
    int fd_ = -1;
    const unsigned char *base_ = nullptr;
    std::size_t size_ = 0;
};

}  // namespace sfm_amd
