// dist_reconstruct.cpp -- replay_reconstruct over several GPUs: one process per GPU, RCCL, C++ host over the C ABI
// (sfm_amd/cpp/dist_fusion.hpp).  Every rank opens the same keyframe replay dump and writes ITS owned voxels:
//     RANK=r WORLD_SIZE=n [LOCAL_RANK=d] [SFM_NCCL_ID_FILE=path] dist_reconstruct <dump.sfmr> <out> [--voxel-size V] [--t-dist T]
// -> <out>.rank<r> in the reference's text format (src/dense_reconstruction.cpp:61-64); the concatenation of the ranks'
// files, sorted, is the one-GPU cloud.  The RCCL id travels through SFM_NCCL_ID_FILE (rank 0 writes it, the others wait
// for it; a launcher with a shared directory is all it needs); one rank needs no file.
#include "dist_fusion.hpp"
#include "replay.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>

int main(int argc, char **argv)
{
    if (argc == 6 && std::string(argv[1]) == "--plan") {
        // dist_reconstruct --plan <keyframes> <window> <world> <closure>: the block plan of a synthetic sequence (keyframe i
        // names i-window .. i+window; closure > 0: keyframe 2 also names keyframe `closure`), one line per rank -- no GPU,
        // no RCCL.  tests/test_sequence_split.py compares it with the Python planner's.
        const std::size_t n = (std::size_t)std::atoll(argv[2]);
        const long window = std::atol(argv[3]), closure = std::atol(argv[5]);
        const int world = std::atoi(argv[4]);
        std::vector<std::vector<std::uint32_t>> nbrs(n);
        for (std::size_t i = 0; i < n; i++)
            for (long j = (long)i - window; j <= (long)i + window; j++)
                if (j >= 0 && j < (long)n && j != (long)i) nbrs[i].push_back((std::uint32_t)j);
        if (closure > 0 && n > 2 && closure < (long)n) {
            nbrs[2].push_back((std::uint32_t)closure);
            std::sort(nbrs[2].begin(), nbrs[2].end());
            nbrs[2].erase(std::unique(nbrs[2].begin(), nbrs[2].end()), nbrs[2].end());
        }
        const auto plans = sfm_amd::plan_sequence(nbrs, std::vector<std::uint64_t>(n, 921600), world);
        for (int r = 0; r < world; r++) {
            const auto &p = plans[r];
            std::cout << r << " block " << p.first << " " << p.end << " frames " << p.frames.size();
            for (const auto &[t, fr] : p.sends) {
                std::cout << " send " << t << ":";
                for (auto g : fr) std::cout << g << ",";
            }
            for (const auto &[q, fr] : p.recvs) {
                std::cout << " recv " << q << ":";
                for (auto g : fr) std::cout << g << ",";
            }
            std::cout << "\n";
        }
        return 0;
    }
    if (argc < 3) {
        std::cerr << "usage: dist_reconstruct <dump> <out> [--voxel-size V] [--t-dist T]\n"
                     "       dist_reconstruct --plan <keyframes> <window> <world> <closure>\n";
        return 2;
    }
    try {
        const auto env_int = [](const char *n, int d) { const char *v = std::getenv(n); return v ? std::atoi(v) : d; };
        const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", rank);
        if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice");
        ncclUniqueId id;
        const char *id_file = std::getenv("SFM_NCCL_ID_FILE");
        if (world > 1 && !id_file) throw std::runtime_error("WORLD_SIZE > 1 needs SFM_NCCL_ID_FILE");
        if (rank == 0) {
            if (ncclGetUniqueId(&id) != ncclSuccess) throw std::runtime_error("ncclGetUniqueId");
            if (id_file) {
                std::ofstream f(std::string(id_file) + ".tmp", std::ios::binary);
                f.write(reinterpret_cast<const char *>(&id), sizeof id);
                f.close();
                std::rename((std::string(id_file) + ".tmp").c_str(), id_file);
            }
        } else {
            for (int tries = 0;; tries++) {
                std::ifstream f(id_file, std::ios::binary);
                if (f && f.read(reinterpret_cast<char *>(&id), sizeof id)) break;
                if (tries > 600) throw std::runtime_error("no RCCL id file");
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        }
        ncclComm_t comm;
        if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) throw std::runtime_error("ncclCommInitRank");
        sfm_amd::Replay replay(argv[1]);
        sfm_amd::DistributedDenseReconstruction dense{replay.map, rank, world, comm, device};
        for (int i = 3; i < argc; i++) {
            const std::string a = argv[i];
            if (a == "--voxel-size" && i + 1 < argc) dense.params().voxel_size = std::atof(argv[++i]);
            else if (a == "--t-dist" && i + 1 < argc) dense.params().T_dist = std::atof(argv[++i]);
            else { std::cerr << "unknown option " << a << "\n"; return 2; }
        }
        const sfm_amd::Cloud cloud = dense.fuse();
        const std::string path = std::string(argv[2]) + ".rank" + std::to_string(rank);
        std::ofstream out(path, std::ios::binary);
        if (!out) throw std::runtime_error("cannot create " + path);
        sfm_amd::write_text_fast(out, cloud.xyz.data(), cloud.bgr.data(), cloud.size());
        out.flush();
        std::cout << "rank " << rank << " of " << world << ": " << cloud.size() << " owned points\n";
        ncclCommDestroy(comm);
    } catch (const std::exception &e) {
        std::cerr << "dist_reconstruct: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
