// writer_probe.cpp -- host-only test helper (no GPU, no library): exposes the C++ stream
// formatting the reference's writer uses, so that tests/test_writer_parity.py can pin the Python
// writer and the fast writer against `std::ostream << double` byte for byte.
//   writer_probe fmt              stdin: one hex u64 per line -> stdout: `os << double` per line
//   writer_probe text|fast|ply|csv FILE   FILE: u64 n, n*3 f64 xyz, n*3 i32 bgr -> stdout: the writer's bytes
#include "../../sfm_amd/cpp/cloud_writer.hpp"

#include <fstream>
#include <iostream>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "fmt") {
        std::string tok;
        while (std::cin >> tok) {
            const uint64_t bits = std::stoull(tok, nullptr, 16);
            double d;
            std::memcpy(&d, &bits, 8);
            std::cout << d << '\n';
        }
        return 0;
    }
    if (argc < 3) return 2;
    std::ifstream in(argv[2], std::ios::binary);
    uint64_t n = 0;
    in.read(reinterpret_cast<char *>(&n), 8);
    std::vector<double> xyz(3 * n);
    std::vector<int32_t> bgr(3 * n);
    in.read(reinterpret_cast<char *>(xyz.data()), (std::streamsize)(24 * n));
    in.read(reinterpret_cast<char *>(bgr.data()), (std::streamsize)(12 * n));
    if (!in) return 3;
    if (mode == "text") sfm_amd::write_text(std::cout, xyz.data(), bgr.data(), n);
    else if (mode == "fast") sfm_amd::write_text_fast(std::cout, xyz.data(), bgr.data(), n);
    else if (mode == "ply") sfm_amd::write_ply_binary(std::cout, xyz.data(), bgr.data(), n);
    else if (mode == "csv") sfm_amd::write_sparse_csv(std::cout, xyz.data(), bgr.data(), n);
    else return 2;
    return 0;
}
