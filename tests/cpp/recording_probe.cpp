// recording_probe.cpp -- host-only test helper: reads a recording directory with
// sfm_amd::RecordingReader and writes every frame raw (u32 W, u32 H, BGR bytes, u32 W, u32 H,
// u16 depth) to stdout; tests/test_recording_reader.py compares with PIL's decode.
#include "../../sfm_amd/cpp/recording.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    sfm_amd::RecordingReader reader(argv[1], argc > 2 ? (std::size_t)std::atoll(argv[2]) : 0);
    unsigned frames = 0;
    while (auto f = reader.read_frames()) {
        const std::uint32_t h1[2] = {(std::uint32_t)f->first.width, (std::uint32_t)f->first.height};
        std::fwrite(h1, 4, 2, stdout);
        std::fwrite(f->first.data.data(), 1, f->first.data.size(), stdout);
        const std::uint32_t h2[2] = {(std::uint32_t)f->second.width, (std::uint32_t)f->second.height};
        std::fwrite(h2, 4, 2, stdout);
        std::fwrite(f->second.data.data(), 2, f->second.data.size(), stdout);
        frames++;
    }
    std::fprintf(stderr, "%u frames\n", frames);
    return 0;
}
