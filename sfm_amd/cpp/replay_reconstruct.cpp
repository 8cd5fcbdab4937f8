// replay_reconstruct.cpp -- the fusion half of the reference's `reconstruct` executable
// (src/reconstruct.cpp:69-73) over a keyframe replay dump:
//     replay_reconstruct <dump.sfmr> <out> [--ply] [--voxel-size V] [--t-dist T] [--rewrite <copy.sfmr>]
// Host-only C++ over the C ABI (libsfm_fusion.so); --rewrite re-exports the dump through the C++
// writer (used by the tests to check the two implementations of the format against each other).
#include "replay.hpp"

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
        sfm_amd::Replay replay(argv[1]);
        bool ply = false;
        std::string rewrite;
        sfm_amd::DenseReconstruction dense{replay.map};
        for (int i = 3; i < argc; i++) {
            const std::string a = argv[i];
            if (a == "--ply") ply = true;
            else if (a == "--voxel-size" && i + 1 < argc) dense.params().voxel_size = std::atof(argv[++i]);
            else if (a == "--t-dist" && i + 1 < argc) dense.params().T_dist = std::atof(argv[++i]);
            else if (a == "--rewrite" && i + 1 < argc) rewrite = argv[++i];
            else { std::cerr << "unknown option " << a << "\n"; return 2; }
        }
        if (!rewrite.empty()) sfm_amd::write_replay(rewrite, replay.map);
        std::ofstream out(argv[2], std::ios::binary);
        if (!out) throw std::runtime_error(std::string("cannot create ") + argv[2]);
        if (ply) dense.reconstruct_ply(out);
        else dense.reconstruct(out);
        std::cout << dense.stats().n_frames << " keyframes selected, " << dense.stats().n_vox << " points\n";
    } catch (const std::exception &e) {
        std::cerr << "replay_reconstruct: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
