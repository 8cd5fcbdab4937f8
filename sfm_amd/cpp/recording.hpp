// recording.hpp -- RecordingReader without OpenCV.
//
// The reference reads a recorded log of numbered files `color_N.png` / `depth_N.png`
// (src/recording.cpp:12-38) with cv::imread(IMREAD_COLOR) -> CV_8UC3 in B,G,R order and
// cv::imread(IMREAD_ANYDEPTH) -> CV_16UC1.  This header gives the same class shape
// (RecordingReader(directory, start_frame), read_frames() -> optional pair) over a small PNG
// decoder of its own: zlib's inflate plus the five PNG row filters.  Handled, like imread does for
// these two calls: colour from 8-bit RGB / RGBA (alpha dropped) / grey (replicated) / palette
// PNGs; depth from 16-bit greyscale PNGs (big-endian samples -> host order).  Anything else
// (interlaced files, 16-bit colour, 8-bit depth) is reported as "not readable", which is what a
// failed imread is to the reference: read_frames() returns nullopt and the caller stops.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <optional>
#include <string>
#include <utility>
#include <vector>
#include <zlib.h>

namespace sfm_amd {

struct Image8 {   // CV_8UC3, B,G,R
    int width = 0, height = 0;
    std::vector<std::uint8_t> data;  // height * width * 3
};
struct Image16 {  // CV_16UC1
    int width = 0, height = 0;
    std::vector<std::uint16_t> data;  // height * width
};

namespace png_detail {

struct Raw {
    std::uint32_t width = 0, height = 0;
    int bit_depth = 0, color_type = 0, channels = 0;
    std::vector<std::uint8_t> pixels;   // unfiltered scanlines, height * stride
    std::vector<std::uint8_t> palette;  // RGB triples
    std::size_t stride = 0;
};

inline std::uint32_t be32(const std::uint8_t *p)
{
    return ((std::uint32_t)p[0] << 24) | ((std::uint32_t)p[1] << 16) | ((std::uint32_t)p[2] << 8) | p[3];
}

inline bool read_file(const std::string &path, std::vector<std::uint8_t> &out)
{
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (std::size_t)n : 0);
    const bool ok = n > 0 && std::fread(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

inline bool decode(const std::vector<std::uint8_t> &file, Raw &r)
{
    static const std::uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) return false;
    std::vector<std::uint8_t> idat;
    bool have_ihdr = false;
    int interlace = 0;
    for (std::size_t pos = 8; pos + 12 <= file.size();) {
        const std::uint32_t len = be32(&file[pos]);
        const std::uint8_t *type = &file[pos + 4];
        if (pos + 12 + (std::size_t)len > file.size()) return false;
        const std::uint8_t *body = &file[pos + 8];
        if (std::memcmp(type, "IHDR", 4) == 0) {
            if (len != 13) return false;
            r.width = be32(body);
            r.height = be32(body + 4);
            r.bit_depth = body[8];
            r.color_type = body[9];
            interlace = body[12];
            have_ihdr = true;
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            r.palette.assign(body, body + len);
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), body, body + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (std::size_t)len;
    }
    if (!have_ihdr || interlace != 0 || r.width == 0 || r.height == 0) return false;
    switch (r.color_type) {
        case 0: r.channels = 1; break;
        case 2: r.channels = 3; break;
        case 3: r.channels = 1; break;
        case 4: r.channels = 2; break;
        case 6: r.channels = 4; break;
        default: return false;
    }
    if (r.bit_depth != 8 && r.bit_depth != 16) return false;  // sub-byte depths are not what the recorder writes
    const std::size_t bpp = (std::size_t)r.channels * r.bit_depth / 8;
    r.stride = (std::size_t)r.width * bpp;
    std::vector<std::uint8_t> raw((r.stride + 1) * r.height);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return false;
    r.pixels.assign(r.stride * r.height, 0);
    const std::vector<std::uint8_t> zero(r.stride, 0);
    for (std::uint32_t y = 0; y < r.height; y++) {
        const std::uint8_t *src = &raw[(r.stride + 1) * y];
        const int filter = src[0];
        src++;
        std::uint8_t *dst = &r.pixels[r.stride * y];
        const std::uint8_t *up = y ? dst - r.stride : zero.data();
        for (std::size_t i = 0; i < r.stride; i++) {
            const int a = i >= bpp ? dst[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int pred;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p,
                              pc = p > c ? p - c : c - p;
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return false;
            }
            dst[i] = (std::uint8_t)(src[i] + pred);
        }
    }
    return true;
}

}  // namespace png_detail

// cv::imread(path, cv::IMREAD_COLOR) for the PNGs a recorder writes
inline bool read_png_bgr(const std::string &path, Image8 &img)
{
    std::vector<std::uint8_t> file;
    png_detail::Raw r;
    if (!png_detail::read_file(path, file) || !png_detail::decode(file, r) || r.bit_depth != 8) return false;
    img.width = (int)r.width;
    img.height = (int)r.height;
    img.data.resize((std::size_t)r.width * r.height * 3);
    const std::size_t n = (std::size_t)r.width * r.height;
    for (std::size_t i = 0; i < n; i++) {
        std::uint8_t R, G, B;
        const std::uint8_t *p = &r.pixels[i * r.channels];
        switch (r.color_type) {
            case 0: case 4: R = G = B = p[0]; break;
            case 3:
                if ((std::size_t)p[0] * 3 + 2 >= r.palette.size()) return false;
                R = r.palette[p[0] * 3]; G = r.palette[p[0] * 3 + 1]; B = r.palette[p[0] * 3 + 2];
                break;
            default: R = p[0]; G = p[1]; B = p[2]; break;
        }
        img.data[3 * i] = B;
        img.data[3 * i + 1] = G;
        img.data[3 * i + 2] = R;
    }
    return true;
}

// cv::imread(path, cv::IMREAD_ANYDEPTH) for 16-bit greyscale depth PNGs
inline bool read_png_depth16(const std::string &path, Image16 &img)
{
    std::vector<std::uint8_t> file;
    png_detail::Raw r;
    if (!png_detail::read_file(path, file) || !png_detail::decode(file, r)) return false;
    if (r.bit_depth != 16 || r.color_type != 0) return false;
    img.width = (int)r.width;
    img.height = (int)r.height;
    const std::size_t n = (std::size_t)r.width * r.height;
    img.data.resize(n);
    for (std::size_t i = 0; i < n; i++) img.data[i] = (std::uint16_t)((r.pixels[2 * i] << 8) | r.pixels[2 * i + 1]);
    return true;
}

// sfm::utils::RecordingReader, src/recording.cpp:9-38
class RecordingReader {
public:
    explicit RecordingReader(const std::string &directory, std::size_t start_frame = 0)
        : frame_counter_(start_frame), directory_(directory) {}

    // {colour image, depth image} of the next frame, or nullopt when a file is missing or
    // unreadable (:19-33); the counter only advances on success (:35)
    std::optional<std::pair<Image8, Image16>> read_frames()
    {
        const std::string c = directory_ + "/color_" + std::to_string(frame_counter_) + ".png";
        const std::string d = directory_ + "/depth_" + std::to_string(frame_counter_) + ".png";
        std::pair<Image8, Image16> out;
        if (!read_png_bgr(c, out.first) || !read_png_depth16(d, out.second)) return std::nullopt;
        frame_counter_ += 1;
        return out;
    }

private:
    std::size_t frame_counter_;
    std::string directory_;
};

}  // namespace sfm_amd
