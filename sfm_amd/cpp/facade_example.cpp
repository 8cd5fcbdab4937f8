// Two-keyframe plane scene through the C++ facade; prints the cloud like the reference's
// dense.txt.  Build: g++ -std=c++17 facade_example.cpp -L../lib -lsfm_fusion -o facade_example
// (tests/test_gpu_cpp_facade.py builds and runs it on the GPU box and compares with the oracle).
#include <cstdio>
#include <iostream>
#include <vector>

#include "dense_reconstruction.hpp"

int main()
{
    const int W = 48, H = 36;
    std::vector<std::uint16_t> depth(W * H, 40);
    std::vector<std::uint8_t> img(W * H * 3);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            img[(y * W + x) * 3 + 0] = (std::uint8_t)((x * 7 + 1) % 256);
            img[(y * W + x) * 3 + 1] = (std::uint8_t)((y * 5 + 2) % 256);
            img[(y * W + x) * 3 + 2] = (std::uint8_t)((x + y + 3) % 256);
        }
    sfm_amd::Map map;
    for (int i = 0; i < 3; i++) {
        sfm_amd::KeyFrame k;
        k.width = W; k.height = H;
        k.depth = depth.data(); k.depth_step = W * 2;
        k.img = img.data(); k.img_step = W * 3;
        k.K = {60.0, 60.0, 23.5, 17.5};
        for (int j = 0; j < 16; j++) k.T_wk[j] = k.T_kw[j] = (j % 5 == 0) ? 1.0 : 0.0;
        k.T_wk[12] = 3.0 * i;   // camera centre (3i, 0, 0)
        k.T_kw[12] = -3.0 * i;
        map.keyframes.push_back(k);
    }
    map.insert_edge(0, 1, 250);   // selected, neighbours
    map.insert_edge(1, 2, 100);   // selects keyframe 2, but 100 is not > 100: not a neighbour
    sfm_amd::DenseReconstruction dense{map};
    dense.params().T_dist = 5.0;
    dense.params().voxel_size = 1e-3;
    dense.set_progress_log(&std::cerr);  // the reference logs to stdout and writes the cloud to ./dense.txt; here the cloud goes to stdout
    dense.reconstruct(std::cout);
    std::fprintf(stderr, "points %llu before %llu\n", (unsigned long long)dense.stats().n_vox,
                 (unsigned long long)dense.stats().n_acc);
    return 0;
}
