// Host-side C++ mirror smoke test: reads the committed golden commit fixture exported as raw u64
// by the pytest wrapper, runs PolynomialBatch::from_values on the GPU and prints the cap so the
// wrapper can compare it with the fixture. Also exercises MerkleTree::prove and the error path.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "plonky2_hip.hpp"

using namespace plonky2_hip;

static std::vector<uint64_t> read_u64(const char *path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    size_t bytes = f.tellg();
    f.seekg(0);
    std::vector<uint64_t> v(bytes / 8);
    f.read(reinterpret_cast<char *>(v.data()), bytes);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s values.bin n_polys rate_bits cap_height out_prefix\n", argv[0]);
        return 2;
    }
    try {
        Context ctx(0);
        std::vector<uint64_t> values = read_u64(argv[1]);
        uint64_t n_polys = std::strtoull(argv[2], nullptr, 10);
        uint32_t rate_bits = std::atoi(argv[3]), cap_height = std::atoi(argv[4]);
        PolynomialBatch b = PolynomialBatch::from_values(ctx, values, n_polys, rate_bits, false, cap_height);
        std::vector<uint64_t> cap = b.merkle_tree.cap();
        std::printf("CAP");
        for (uint64_t x : cap) std::printf(" %llu", (unsigned long long)x);
        std::printf("\n");
        std::vector<uint64_t> row = b.get_lde_values(3);
        std::printf("ROW3");
        for (uint64_t x : row) std::printf(" %llu", (unsigned long long)x);
        std::printf("\n");
        MerkleProof p = b.merkle_tree.prove(5);
        std::printf("PROOF5");
        for (auto &s : p.siblings)
            for (uint64_t x : s) std::printf(" %llu", (unsigned long long)x);
        std::printf("\n");
        std::vector<uint64_t> c = b.polynomials();
        std::vector<uint64_t> back = fft_with_options(ctx, c, n_polys);
        std::printf("FFT_OF_COEFFS_EQUALS_VALUES %d\n", (int)(back == values));
        try {
            MerkleTree::new_(ctx, std::vector<uint64_t>(256 * 7, 1), 256, 9);  // merkle_tree.rs:470-482 should_panic
            std::printf("CAP_TOO_BIG no-error\n");
        } catch (const Error &e) {
            std::printf("CAP_TOO_BIG error %d\n", e.code);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "FAILED: %s\n", e.what());
        return 1;
    }
    return 0;
}
