// cloud_writer.hpp -- writers for the fused cloud.
//
//   write_text        the reference's writer loop, src/dense_reconstruction.cpp:61-64, byte for
//                     byte: `os << x << " " << y << " " << z << " " << R << " " << G << " " << B`
//                     with the stream's default double formatting (precision 6, floatfield unset
//                     == printf("%g")) and one line per point.  The reference ends every line with
//                     std::endl, i.e. '\n' plus a flush; the bytes are the same.
//   write_text_fast   the same bytes, formatted with snprintf("%g") by up to 16 host threads: no
//                     per-point stream call, no per-line flush (the flush is the reference's next
//                     bottleneck once fusion runs on the GPU).
//   write_ply_binary  binary little-endian PLY, lossless (double coordinates), colour as uchar
//                     red/green/blue.  No counterpart in the reference.
//   write_sparse_csv  the comma-separated variant the reference uses for its sparse cloud,
//                     src/map.cpp:365-374 ("x, y, z, r, g, b").
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ostream>
#include <algorithm>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace sfm_amd {

// xyz: n*3 doubles; bgr: n*3 ints in B,G,R order (PointInfo::color)
inline void write_text(std::ostream &os, const double *xyz, const int32_t *bgr, std::size_t n)
{
    for (std::size_t i = 0; i < n; i++)
        os << xyz[3 * i] << " " << xyz[3 * i + 1] << " " << xyz[3 * i + 2] << " " << bgr[3 * i + 2] << " "
           << bgr[3 * i + 1] << " " << bgr[3 * i] << '\n';
}

inline void format_text_range(std::string &buf, const double *xyz, const int32_t *bgr, std::size_t i0, std::size_t i1)
{
    char line[160];
    for (std::size_t i = i0; i < i1; i++) {
        const int len = std::snprintf(line, sizeof line, "%g %g %g %d %d %d\n", xyz[3 * i], xyz[3 * i + 1],
                                      xyz[3 * i + 2], bgr[3 * i + 2], bgr[3 * i + 1], bgr[3 * i]);
        buf.append(line, (std::size_t)len);
    }
}

// threads = 0: one per hardware thread, at most 16; blocks of 64 Ki points are formatted in
// parallel and written in order
inline void write_text_fast(std::ostream &os, const double *xyz, const int32_t *bgr, std::size_t n,
                            unsigned threads = 0)
{
    constexpr std::size_t kBlock = 1 << 16;
    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    const std::size_t blocks = (n + kBlock - 1) / kBlock;
    threads = (unsigned)std::min<std::size_t>(threads, std::max<std::size_t>(blocks, 1));
    std::vector<std::string> buf(threads);
    for (std::size_t b0 = 0; b0 < blocks; b0 += threads) {
        const std::size_t nb = std::min<std::size_t>(threads, blocks - b0);
        std::vector<std::thread> pool;
        for (std::size_t t = 0; t < nb; t++) {
            buf[t].clear();
            const std::size_t i0 = (b0 + t) * kBlock, i1 = std::min(n, i0 + kBlock);
            if (nb == 1) format_text_range(buf[t], xyz, bgr, i0, i1);
            else pool.emplace_back(format_text_range, std::ref(buf[t]), xyz, bgr, i0, i1);
        }
        for (auto &th : pool) th.join();
        for (std::size_t t = 0; t < nb; t++) os.write(buf[t].data(), (std::streamsize)buf[t].size());
    }
}

inline void write_ply_binary(std::ostream &os, const double *xyz, const int32_t *bgr, std::size_t n)
{
    os << "ply\nformat binary_little_endian 1.0\ncomment fused by sfm_amd (MI355X dense fusion)\n"
       << "element vertex " << n << "\nproperty double x\nproperty double y\nproperty double z\n"
       << "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
    std::vector<unsigned char> rec(27 * std::min<std::size_t>(n, 1 << 15));
    for (std::size_t i0 = 0; i0 < n; i0 += 1 << 15) {
        const std::size_t m = std::min<std::size_t>(n - i0, 1 << 15);
        for (std::size_t j = 0; j < m; j++) {
            const std::size_t i = i0 + j;
            std::memcpy(&rec[27 * j], &xyz[3 * i], 24);  // x86-64 and gfx950 hosts are little-endian
            rec[27 * j + 24] = (unsigned char)bgr[3 * i + 2];
            rec[27 * j + 25] = (unsigned char)bgr[3 * i + 1];
            rec[27 * j + 26] = (unsigned char)bgr[3 * i];
        }
        os.write(reinterpret_cast<const char *>(rec.data()), (std::streamsize)(27 * m));
    }
}

inline void write_sparse_csv(std::ostream &os, const double *xyz, const int32_t *bgr, std::size_t n)
{
    for (std::size_t i = 0; i < n; i++)
        os << xyz[3 * i] << ", " << xyz[3 * i + 1] << ", " << xyz[3 * i + 2] << ", " << bgr[3 * i + 2] << ", "
           << bgr[3 * i + 1] << ", " << bgr[3 * i] << '\n';
}

}  // namespace sfm_amd
