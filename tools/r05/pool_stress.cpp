// Stress of libfpcc_host's background coder pool for the sanitizer builds (tools/r05/sanitize.sh).
//
// Models the two hand-overs of the codec exactly as fastpcc_amd/codecs/geo_lossl_em.py drives them, with a CPU thread in the role of
// the GPU's copy engine:
//   encoder: per frame 6 binary jobs + 1 histogram job are SUBMITTED first, on buffers that still hold the previous frame's
//            content (or poison); a producer thread then fills each job's inputs and stores the job's flag last (release);
//            fpcc_pool_wait; every stream is decoded and compared with what was handed over;
//   decoder: a table-decode job publishes progress while the consumer takes prefixes (fpcc_progress_wait), as `residuals()` does.
// Two pools with their own producer and consumer threads run side by side (two frames in flight, fastpcc_amd/serving.py).  Buffers
// are heap blocks freed after every frame, so AddressSanitizer sees any use after the wait; ThreadSanitizer sees any access that is
// not ordered by the flag / the pool's mutex.
//
//   g++ -O1 -g -fsanitize=thread  -std=c++17 -pthread tools/r05/pool_stress.cpp fastpcc_amd/csrc/host/rans_host.cpp -o /tmp/pool_tsan
//   usage: pool_stress [frames per context = 800] [contexts = 2]      (7 encode jobs + 1 decode job per frame)
#include "../../include/fpcc_host.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

namespace {

struct Level {
    std::vector<uint8_t> bits, want_bits;
    std::vector<uint16_t> prob, want_prob;
    std::vector<uint8_t> out;
    int64_t len = 0;
};

int run_context(int ctx, int frames, std::atomic<long> *jobs_done) {
    std::mt19937_64 rng(1234 + ctx);
    fpcc_pool *pool = fpcc_pool_new(8);
    if (!pool) return 1;
    // one block of flags per context, zeroed by the consumer at the start of every frame (geo_lossl_em.py: st['flags'].zero_())
    std::vector<uint32_t> flags(64, 0);
    int bad = 0;
    for (int f = 0; f < frames && !bad; ++f) {
        std::fill(flags.begin(), flags.end(), 0u);
        const int n_levels = 6;
        std::vector<Level> lv(n_levels);
        for (int l = 0; l < n_levels; ++l) {
            const int64_t n = 8 + (int64_t)(rng() % (l == n_levels - 1 ? 200000 : 6000));
            Level &L = lv[l];
            L.want_bits.resize(n); L.want_prob.resize(n);
            for (int64_t i = 0; i < n; ++i) {
                L.want_prob[i] = (uint16_t)(1 + rng() % 65535);
                L.want_bits[i] = (rng() & 0xffff) < L.want_prob[i];
            }
            // what the job must NOT read: poison (probability 0 makes the coder refuse, a wrong bit makes the round trip fail)
            L.bits.assign(n, 1); L.prob.assign(n, 0);
            L.out.resize(4 * n + 64);
        }
        const int64_t n_sym = 64 + (int64_t)(rng() % 300000);
        std::vector<int32_t> want_sym(n_sym), sym(n_sym, 1 << 20);
        for (auto &s : want_sym) s = (int32_t)(rng() % 41) - 20;
        std::vector<uint8_t> sym_out(4 * n_sym + 64);
        std::vector<uint32_t> cdf(1 << 12);
        int64_t cdf_len = 0, sym_len = 0;
        int32_t offset = 0;
        // submit first ...
        int n_flags = 0;
        if (fpcc_pool_histogram_encode(pool, &flags[n_flags], 1, sym.data(), n_sym, 0, &offset, cdf.data(), (int64_t)cdf.size(), &cdf_len,
                                       sym_out.data(), (int64_t)sym_out.size(), &sym_len) < 0) bad = 1;
        const int sym_flag = n_flags++;
        std::vector<int> level_flag(n_levels);
        for (int l = n_levels - 1; l >= 0 && !bad; --l) {                 // finest first, as `defer_occupancy` does
            Level &L = lv[l];
            level_flag[l] = n_flags;
            if (fpcc_pool_binary_encode(pool, &flags[n_flags++], 1, L.bits.data(), L.prob.data(), (int64_t)L.bits.size(), L.out.data(),
                                        (int64_t)L.out.size(), &L.len) < 0) bad = 1;
        }
        // ... then the "copy engine" delivers the inputs, each followed by its flag
        std::thread producer([&] {
            auto deliver = [&](void *dst, const void *src, size_t bytes, uint32_t *flag) {
                if (rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 200));
                std::memcpy(dst, src, bytes);
                if (flag) __atomic_store_n(flag, 1u, __ATOMIC_RELEASE);
            };
            deliver(sym.data(), want_sym.data(), n_sym * sizeof(int32_t), &flags[sym_flag]);
            for (int l = n_levels - 1; l >= 0; --l) {
                deliver(lv[l].bits.data(), lv[l].want_bits.data(), lv[l].bits.size(), nullptr);
                deliver(lv[l].prob.data(), lv[l].want_prob.data(), lv[l].prob.size() * 2, &flags[level_flag[l]]);
            }
        });
        const int64_t rc = fpcc_pool_wait(pool);
        producer.join();
        if (rc < 0) { std::fprintf(stderr, "ctx %d frame %d: pool wait %lld (%s)\n", ctx, f, (long long)rc, fpcc_host_strerror(rc)); bad = 1; break; }
        // every stream decodes to what was delivered
        for (int l = 0; l < n_levels && !bad; ++l) {
            Level &L = lv[l];
            std::vector<uint8_t> got(L.bits.size());
            if (L.len < 4 || fpcc_rans_binary_decode(L.out.data() + L.out.size() - L.len, L.len, L.want_prob.data(), (int64_t)got.size(), got.data()) < 0 ||
                std::memcmp(got.data(), L.want_bits.data(), got.size()) != 0) {
                std::fprintf(stderr, "ctx %d frame %d level %d: round trip differs\n", ctx, f, l);
                bad = 1;
            }
        }
        // decoder side: background table decode with progress, prefixes consumed while it runs
        if (!bad) {
            std::vector<int32_t> dec(n_sym, -99);
            int64_t progress = 0;
            const int64_t first = 16;
            if (sym_len < 4 || fpcc_pool_table_decode(pool, sym_out.data() + sym_out.size() - sym_len, sym_len, n_sym, cdf.data(), cdf_len, offset, dec.data(),
                                                      first, &progress) < 0) bad = 1;
            int64_t taken = 0;
            while (!bad && taken < n_sym) {
                const int64_t want = std::min<int64_t>(n_sym, taken + 1 + (int64_t)(rng() % 50000));
                if (fpcc_progress_wait(&progress, want) < 0) { bad = 1; break; }
                if (std::memcmp(dec.data() + taken, want_sym.data() + taken, (want - taken) * sizeof(int32_t)) != 0) {
                    std::fprintf(stderr, "ctx %d frame %d: decoded prefix [%lld, %lld) differs\n", ctx, f, (long long)taken, (long long)want);
                    bad = 1;
                }
                taken = want;
            }
            if (fpcc_pool_wait(pool) < 0) bad = 1;
        }
        jobs_done->fetch_add(n_levels + 2);
    }
    fpcc_pool_free(pool);
    return bad;
}

}  // namespace

int main(int argc, char **argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 800;
    const int contexts = argc > 2 ? std::atoi(argv[2]) : 2;
    std::atomic<long> jobs{0};
    std::vector<int> rc(contexts, 0);
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < contexts; ++c) th.emplace_back([&, c] { rc[c] = run_context(c, frames, &jobs); });
    for (auto &t : th) t.join();
    int bad = 0;
    for (int r : rc) bad |= r;
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("pool_stress: %d contexts x %d frames, %ld jobs, %.1f s: %s\n", contexts, frames, jobs.load(), s, bad ? "FAILED" : "clean");
    return bad;
}
