// replay_reconstruct.cpp -- the fusion half of the reference's `reconstruct` executable
// (src/reconstruct.cpp:69-73) over a keyframe replay dump:
//     replay_reconstruct <dump.sfmr> <out> [--ply] [--voxel-size V] [--t-dist T] [--rewrite <copy.sfmr>] [--timing] [--quiet]
// Host-only C++ over the C ABI (libsfm_fusion.so); --rewrite re-exports the dump through the C++
// writer (used by the tests to check the two implementations of the format against each other).
#include "replay.hpp"

#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::cerr << "usage: replay_reconstruct <dump> <out> [--ply] [--voxel-size V] [--t-dist T] [--rewrite <copy>]\n";
        return 2;
    }
    try {
        using clk = std::chrono::steady_clock;
        const auto ms = [](clk::time_point a, clk::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        const auto t0 = clk::now();
        sfm_amd::Replay replay(argv[1]);
        const auto t1 = clk::now();
        bool ply = false, timing = false, quiet = false;
        std::string rewrite;
        sfm_amd::DenseReconstruction dense{replay.map};
        for (int i = 3; i < argc; i++) {
            const std::string a = argv[i];
            if (a == "--ply") ply = true;
            else if (a == "--voxel-size" && i + 1 < argc) dense.params().voxel_size = std::atof(argv[++i]);
            else if (a == "--t-dist" && i + 1 < argc) dense.params().T_dist = std::atof(argv[++i]);
            else if (a == "--rewrite" && i + 1 < argc) rewrite = argv[++i];
            else if (a == "--timing") timing = true;
            else if (a == "--quiet") quiet = true;  // no progress log (the reference always prints it, dense_reconstruction.cpp:49-58)
            else { std::cerr << "unknown option " << a << "\n"; return 2; }
        }
        if (!rewrite.empty()) sfm_amd::write_replay(rewrite, replay.map);
        std::ofstream out(argv[2], std::ios::binary);
        if (!out) throw std::runtime_error(std::string("cannot create ") + argv[2]);
        const auto t2 = clk::now();
        const sfm_amd::Cloud cloud = dense.fuse();
        const auto t3 = clk::now();
        if (!quiet) dense.write_progress_log(std::cout);
        if (ply) sfm_amd::write_ply_binary(out, cloud.xyz.data(), cloud.bgr.data(), cloud.size());
        else sfm_amd::write_text_fast(out, cloud.xyz.data(), cloud.bgr.data(), cloud.size());
        out.flush();
        const auto t4 = clk::now();
        if (timing)
            std::cout << "timing (ms): open dump " << ms(t0, t1) << ", upload + fuse + read back " << ms(t2, t3)
                      << " (device run " << dense.stats().last_run_ms << "), write " << ms(t3, t4) << "\n";
        std::cout << dense.stats().n_frames << " keyframes selected, " << dense.stats().n_vox << " points\n";
    } catch (const std::exception &e) {
        std::cerr << "replay_reconstruct: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
