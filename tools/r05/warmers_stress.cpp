// The 255-ary decoder's row warmers under the sanitizers: several threads decode blocks of cold rows at once (one owns the helpers, the
// others go without), buffers are freed right after each decode.
//   g++ -O1 -g -fsanitize=thread -std=c++17 -pthread tools/r05/warmers_stress.cpp fastpcc_amd/csrc/host/rans_host.cpp -o /tmp/warm_tsan
#include "../../include/fpcc_host.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

static int one_round(int seed) {
    std::mt19937_64 rng(seed);
    const int64_t n = 2500 + (int64_t)(rng() % 3000), width = 255;
    std::vector<uint16_t> *rows = new std::vector<uint16_t>((size_t)(n * width));
    std::vector<uint16_t> sym((size_t)n), out((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        uint32_t acc = 0;
        for (int64_t j = 0; j < width; ++j) { acc += 1 + (uint32_t)(rng() % 250); (*rows)[(size_t)(i * width + j)] = (uint16_t)acc; }
        (*rows)[(size_t)(i * width + width - 1)] = 65535;
        sym[(size_t)i] = (uint16_t)(rng() % width);
    }
    fpcc_simple_enc *e = fpcc_simple_enc_new(16 << 20);
    if (fpcc_simple_enc_push(e, rows->data(), n, width, sym.data(), n) < 0) return 1;
    std::vector<uint8_t> stream(16 << 20);
    const int64_t len = fpcc_simple_enc_finish(e, stream.data(), (int64_t)stream.size());
    fpcc_simple_enc_free(e);
    if (len < 4) return 1;
    fpcc_simple_dec *d = fpcc_simple_dec_new(stream.data(), len);
    if (!d || fpcc_simple_dec_pop(d, rows->data(), n, width, out.data(), n) < 0) return 1;
    fpcc_simple_dec_free(d);
    delete rows;                                   // freed right after the decode: a helper still reading would be caught by ASan
    return std::memcmp(out.data(), sym.data(), (size_t)n * 2) != 0;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    std::vector<int> rc(4, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t) th.emplace_back([&, t] { for (int r = 0; r < rounds && !rc[t]; ++r) rc[t] = one_round(t * 1000 + r); });
    for (auto &t : th) t.join();
    int bad = rc[0] | rc[1] | rc[2] | rc[3];
    std::printf("warmers_stress: 4 threads x %d decodes of 2500-5500 cold 255-entry rows: %s\n", rounds, bad ? "FAILED" : "clean");
    return bad;
}
