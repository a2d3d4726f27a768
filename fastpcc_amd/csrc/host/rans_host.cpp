// libfpcc_host.so -- host-side entropy coders of the hot path (C ABI in include/fpcc_host.h).
//
// Byte-wise rANS with a 32-bit state, L = 2^23, 16-bit probabilities: the stream format of the reference's
// rans_ext_cpp / simple_rans_ext_cpp (format description: /root/reference/lib/entropy_models/rans_coder/rans_byte.h:66-165),
// so streams are interchangeable with the reference's.  Written from the format, not from the reference sources:
// one templated writer/reader pair serves the indexed, binary and persistent-stream coders.
#include "../../../include/fpcc_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <cstring>
#include <new>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#endif
#include <cstdio>
#include <cstdlib>

namespace {

constexpr uint32_t kLow = 1u << 23;   // lower bound of the normalised state interval
constexpr uint32_t kProbBits = 16;
constexpr uint32_t kProbOne = 1u << kProbBits;
constexpr int kMaxRefill = 3;         // bytes a decoder may shift in per symbol (2 suffice for any valid stream)

// a quantised CDF the coders accept: starts at 0, strictly increasing, ends at 2^16
bool cdf_well_formed(const uint32_t *c, int64_t len) {
    if (len < 2 || c[0] != 0 || c[len - 1] != kProbOne) return false;
    for (int64_t i = 1; i < len; ++i) if (c[i] <= c[i - 1]) return false;
    return true;
}

// Writes a stream backwards into [base, base + cap).
struct RcpTable {
    uint64_t m[65537];
    RcpTable() {
        m[0] = 0;
        for (uint32_t d = 1; d <= 65536u; ++d) m[d] = ((uint64_t(1) << 47) + d - 1) / d;      // ceil(2^47 / d)
    }
};
static const RcpTable kRcp;

class BackWriter {
public:
    BackWriter(uint8_t *base, int64_t cap) : base_(base), cur_(base + cap), end_(base + cap), state_(kLow) {}

    // range [start, start + freq) out of 2^bits; freq >= 1
    template <uint32_t BITS>
    inline void put(uint32_t start, uint32_t freq) {
        uint32_t x = state_;
        const uint32_t ceiling = ((kLow >> BITS) << 8) * freq;
        while (x >= ceiling) {
            if (cur_ == base_) { full_ = true; break; }
            *--cur_ = static_cast<uint8_t>(x);
            x >>= 8;
        }
        // x / freq without a divide on the state's dependency chain: x < 2^31 (it is below the ceiling) and freq <= 2^16, so
        // floor(x * ceil(2^47 / freq) / 2^47) == floor(x / freq) exactly (the error term is < 2^-16 <= 1 / freq); the
        // reciprocal comes from a 65537-entry table (512 KB, fetched off the chain -- it depends on the symbol, not on x)
        const uint32_t q = freq <= 65536u ? static_cast<uint32_t>((static_cast<unsigned __int128>(x) * kRcp.m[freq]) >> 47)
                                          : x / freq;
        state_ = x + start + q * ((1u << BITS) - freq);           // == (q << BITS) + (x - q * freq) + start, one multiply-add on the chain
    }

    // the same step without a data-dependent branch, for callers that have checked the room (2 bytes per symbol) up front: a symbol
    // shifts out 0, 1 or 2 bytes (x < 2^31, the ceiling >= 2^15 * freq), which of them is as good as random to the branch predictor
    // -- the loop above spends more on mispredictions than on arithmetic.  Both bytes are stored below the write position whether or
    // not they are emitted; what is not emitted is overwritten by the next symbol or by finish().
    template <uint32_t BITS>
    inline void put_roomy(uint32_t start, uint32_t freq) {
        static_assert(BITS == 16, "the byte count below is derived for 16-bit frequencies");
        uint32_t x = state_;
        const uint32_t ceiling = ((kLow >> BITS) << 8) * freq;
        const uint32_t k = static_cast<uint32_t>(x >= ceiling) + static_cast<uint32_t>((x >> 8) >= ceiling);
        cur_[-1] = static_cast<uint8_t>(x);
        cur_[-2] = static_cast<uint8_t>(x >> 8);
        cur_ -= k;
        x >>= 8 * k;
        const uint32_t q = static_cast<uint32_t>((static_cast<unsigned __int128>(x) * kRcp.m[freq]) >> 47);
        state_ = x + start + q * ((1u << BITS) - freq);
    }
    bool room(int64_t bytes) const { return cur_ - base_ >= bytes; }

    // emit the state and return the stream length, or a negative code
    int64_t finish() {
        if (full_ || cur_ - base_ < 4) return FPCC_HOST_E_BUFFER;
        cur_ -= 4;
        cur_[0] = static_cast<uint8_t>(state_);
        cur_[1] = static_cast<uint8_t>(state_ >> 8);
        cur_[2] = static_cast<uint8_t>(state_ >> 16);
        cur_[3] = static_cast<uint8_t>(state_ >> 24);
        return end_ - cur_;
    }

    const uint8_t *head() const { return cur_; }
    int64_t buffered() const { return end_ - cur_; }
    bool full() const { return full_; }
    void reset() { cur_ = end_; state_ = kLow; full_ = false; }

private:
    uint8_t *base_, *cur_, *end_;
    uint32_t state_;
    bool full_ = false;
};

class FrontReader {
public:
    FrontReader(const uint8_t *p, int64_t len) : p_(p + 4), end_(p + len) {
        state_ = uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24;
    }
    template <uint32_t BITS>
    inline uint32_t peek() const { return state_ & ((1u << BITS) - 1u); }
    // a valid encoder flushes a state in [2^23, 2^31); anything below marks a corrupt or truncated stream
    bool valid() const { return state_ >= kLow; }
    uint32_t state() const { return state_; }
    const uint8_t *cursor() const { return p_; }
    template <uint32_t BITS>
    inline void take(uint32_t start, uint32_t freq) {
        uint32_t x = freq * (state_ >> BITS) + (state_ & ((1u << BITS) - 1u)) - start;
        // A state of a valid stream is >= 2^7 after the update, so at most two bytes bring it back above 2^23.  The
        // refill is bounded (never `while`): a corrupt stream whose state collapses to 0 would otherwise shift in zeros
        // for ever.  Reading past the end yields zeros: garbage out, never out of bounds, always finite.
        for (int r = 0; r < kMaxRefill && x < kLow; ++r) {
            uint32_t b = p_ < end_ ? *p_ : 0u;
            ++p_;
            x = (x << 8) | b;
        }
        state_ = x;
    }
    // The same step for the long symbol loops (255-ary rows of the integer codec): whether a symbol shifts in 0, 1 or 2 bytes is as
    // good as random to the branch predictor, and the loop above pays for that on the decoder's serial chain.  With two readable
    // bytes ahead and a state that needs at most two (every valid stream: x >= 2^7), the count is arithmetic and both bytes are read
    // unconditionally; anything else (the stream's last bytes, a corrupt state) takes the checked loop -- same result in every case.
    template <uint32_t BITS>
    inline void take_fast(uint32_t start, uint32_t freq) {
        uint32_t x = freq * (state_ >> BITS) + (state_ & ((1u << BITS) - 1u)) - start;
        if (__builtin_expect(end_ - p_ >= 2 && x >= (1u << 7), 1)) {
            const uint32_t k = static_cast<uint32_t>(x < kLow) + static_cast<uint32_t>(x < (kLow >> 8));
            const uint32_t w = uint32_t(p_[0]) << 8 | uint32_t(p_[1]);
            state_ = (x << (8 * k)) | (w >> (16 - 8 * k));
            p_ += k;
            return;
        }
        for (int r = 0; r < kMaxRefill && x < kLow; ++r) {
            uint32_t b = p_ < end_ ? *p_ : 0u;
            ++p_;
            x = (x << 8) | b;
        }
        state_ = x;
    }

private:
    const uint8_t *p_, *end_;
    uint32_t state_;
};

struct Tables {
    const uint32_t *cdf;
    const int64_t *start;
    const int64_t *len;
    const int32_t *offsets;
    int64_t count;
};

template <bool ESCAPE>
int64_t indexed_encode(const int32_t *sym, const int32_t *index, int64_t n, const Tables &t, uint8_t *out, int64_t cap) {
    BackWriter w(out, cap);
    for (int64_t i = n - 1; i >= 0; --i) {
        const int64_t ti = index ? index[i] : i % t.count;
        if (ti < 0 || ti >= t.count) return FPCC_HOST_E_ARG;
        const uint32_t *c = t.cdf + t.start[ti];
        const int32_t bins = static_cast<int32_t>(t.len[ti]) - 1;
        int32_t v = sym[i] - t.offsets[ti];
        if (ESCAPE) {
            // Values outside [0, bins-2] go through the last bin followed by an Elias-gamma-like tail of 1-bit
            // uniform symbols.  Pushed in reverse of the decoder's reading order.
            const int32_t esc = bins - 1;
            const bool neg = v < 0;
            uint32_t mag = 0;
            if (neg) { mag = static_cast<uint32_t>(-static_cast<int64_t>(v)); v = esc; }
            else if (v >= esc) { mag = static_cast<uint32_t>(v - esc + 1); v = esc; }
            if (v == esc) {
                w.put<1>(neg ? 1u : 0u, 1u);
                int width = 0;
                for (; mag != 0; mag >>= 1, ++width) w.put<1>(mag & 1u, 1u);
                for (int z = 1; z < width; ++z) w.put<1>(0u, 1u);
            }
        } else if (v < 0 || v >= bins) {
            return FPCC_HOST_E_ARG;
        }
        const uint32_t lo = c[v], hi = c[v + 1];
        if (hi <= lo) return FPCC_HOST_E_ARG;
        w.put<kProbBits>(lo, hi - lo);
    }
    return w.finish();
}

// One table, no escape symbols, many symbols: resolve the 16-bit slot through a 64 Ki-entry lookup table instead of a
// binary search per symbol (the decoder's critical path: the next state depends on the decoded symbol).
// `progress`, when given, is advanced (release order) as symbols become final, so that a consumer on another thread can
// use out[0 .. *progress) while the rest of the stream is still being decoded.
int64_t single_table_decode(const uint8_t *stream, int64_t stream_len, int64_t n, const uint32_t *c, int32_t bins,
                            int32_t offset, int32_t *out, std::atomic<int64_t> *progress = nullptr,
                            int64_t first_chunk = 0) {
    std::vector<uint16_t> slot_to_bin(kProbOne);
    for (int32_t b = 0; b < bins; ++b)
        for (uint32_t s = c[b]; s < c[b + 1]; ++s) slot_to_bin[s] = static_cast<uint16_t>(b);
    FrontReader r(stream, stream_len);
    if (!r.valid() && n > 0) return FPCC_HOST_E_ARG;
    int64_t next_mark = progress ? std::min<int64_t>(n, first_chunk > 0 ? first_chunk : 4096) : n;
    for (int64_t i = 0; i < n;) {
        for (; i < next_mark; ++i) {
            const uint32_t v = slot_to_bin[r.peek<kProbBits>()];
            r.take<kProbBits>(c[v], c[v + 1] - c[v]);
            out[i] = static_cast<int32_t>(v) + offset;
        }
        if (progress) progress->store(i, std::memory_order_release);
        next_mark = std::min<int64_t>(n, next_mark + 4096);
    }
    return FPCC_HOST_OK;
}

template <bool ESCAPE>
int64_t indexed_decode(const uint8_t *stream, int64_t stream_len, const int32_t *index, int64_t n, const Tables &t,
                       int32_t *out) {
    if (stream_len < 4) return FPCC_HOST_E_ARG;
    for (int64_t ti = 0; ti < t.count; ++ti)
        if (!cdf_well_formed(t.cdf + t.start[ti], t.len[ti])) return FPCC_HOST_E_CDF;
    if (!ESCAPE && t.count == 1 && n >= 8192 && t.len[0] - 1 <= 65535)
        return single_table_decode(stream, stream_len, n, t.cdf + t.start[0], static_cast<int32_t>(t.len[0]) - 1,
                                   t.offsets[0], out);
    FrontReader r(stream, stream_len);
    if (!r.valid() && n > 0) return FPCC_HOST_E_ARG;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t ti = index ? index[i] : i % t.count;
        if (ti < 0 || ti >= t.count) return FPCC_HOST_E_ARG;
        const uint32_t *c = t.cdf + t.start[ti];
        const int32_t bins = static_cast<int32_t>(t.len[ti]) - 1;
        const uint32_t target = r.peek<kProbBits>();
        // bin whose range holds `target`: first edge above it, minus one
        int32_t v = static_cast<int32_t>(std::upper_bound(c + 1, c + bins + 1, target) - c) - 1;
        r.take<kProbBits>(c[v], c[v + 1] - c[v]);
        if (ESCAPE && v == bins - 1) {
            int width = 0;
            while (r.peek<1>() == 0) {
                r.take<1>(0u, 1u);
                if (++width > 31) return FPCC_HOST_E_ARG;      // no int32 magnitude is that wide: corrupt stream
            }
            r.take<1>(1u, 1u);
            int32_t mag = 1;
            for (int b = 0; b < width; ++b) {
                const uint32_t bit = r.peek<1>();
                r.take<1>(bit, 1u);
                mag = (mag << 1) | static_cast<int32_t>(bit);
            }
            const uint32_t neg = r.peek<1>();
            r.take<1>(neg, 1u);
            v = neg ? -mag : mag + (bins - 1) - 1;
        }
        out[i] = v + t.offsets[ti];
    }
    return FPCC_HOST_OK;
}

int64_t binary_encode(const uint8_t *bits, const uint16_t *p1, int64_t n, uint8_t *out, int64_t cap) {
    BackWriter w(out, cap);
    if (w.room(2 * n + 8)) {
        // room for the worst case (two bytes per symbol): the branch-free step (round 6 -- the loop below spends more on mispredicted
        // renormalisation branches than on arithmetic: the largest occupancy level of a 1 M-voxel frame, 709 K symbols, is the serial
        // tail the GPU waits for at the end of every encode).  Same state sequence, same bytes.
        uint32_t bad = 0;
        for (int64_t i = n - 1; i >= 0; --i) {
            const uint32_t p = p1[i];
            bad |= uint32_t(p == 0);
            const uint32_t mask = 0u - uint32_t(bits[i] != 0);
            const uint32_t split = kProbOne - p;
            w.put_roomy<kProbBits>(split & mask, ((p & mask) | (split & ~mask)) | uint32_t(p == 0));     // (a zero probability: any valid frequency; the result is discarded)
        }
        if (bad) return FPCC_HOST_E_ARG;
        return w.finish();
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        const uint32_t p = p1[i];
        if (p == 0) return FPCC_HOST_E_ARG;
        // ones sit at the top of the interval: [65536 - p, 65536); zeros at [0, 65536 - p).  Selects, not a branch: the
        // symbol is a coin flip for the branch predictor wherever the model is unsure
        const uint32_t mask = 0u - uint32_t(bits[i] != 0);
        const uint32_t split = kProbOne - p;
        w.put<kProbBits>(split & mask, (p & mask) | (split & ~mask));
    }
    return w.finish();
}

}  // namespace

extern "C" {

const char *fpcc_host_strerror(int64_t code) {
    switch (code) {
        case FPCC_HOST_OK: return "ok";
        case FPCC_HOST_E_BUFFER: return "output buffer too small";
        case FPCC_HOST_E_ARG: return "invalid argument";
        case FPCC_HOST_E_CDF: return "cdf cannot be made strictly increasing";
        case FPCC_HOST_E_TIMEOUT: return "timed out waiting for the inputs of a background job";
        default: return "unknown error";
    }
}

int64_t fpcc_pmf_to_quantized_cdf(const double *pmf, int64_t n, int overflow, int32_t *offset, uint32_t *cdf) {
    if (!pmf || !cdf || n < 1 || (overflow && !offset)) return FPCC_HOST_E_ARG;
    double mass = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        if (!(pmf[i] >= 0.0) || !std::isfinite(pmf[i])) return FPCC_HOST_E_ARG;
        mass += pmf[i];
    }
    double denom = mass;
    if (overflow) denom += std::max(1.0 - mass, 0.0);   // the escape bin takes what is left of 1
    if (!(denom > 0.0) || !std::isfinite(denom)) return FPCC_HOST_E_ARG;   // all-zero histogram (the reference asserts)

    std::vector<uint32_t> edges(static_cast<size_t>(n) + (overflow ? 2 : 1));
    double running = 0.0;
    edges[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        running += pmf[i];
        edges[i + 1] = static_cast<uint32_t>(std::round(double(kProbOne) * (running / denom)));
    }
    edges.back() = kProbOne;

    if (overflow) {
        const int64_t m = static_cast<int64_t>(edges.size());
        int64_t head = 0, tail = 0;
        for (int64_t i = 0; i + 1 < m; ++i) if (edges[i + 1] != edges[i]) { head = i; break; }
        for (int64_t i = m - 2; i > 0; --i) if (edges[i - 1] != edges[i]) { tail = i; break; }
        *offset += static_cast<int32_t>(head);
        if (head > tail) { head = m - 3; tail = head + 1; }   // all mass in the escape bin
        std::vector<uint32_t> kept(edges.begin() + head, edges.begin() + tail + 1);
        kept.push_back(kProbOne);
        edges.swap(kept);
    }

    const int64_t bins = static_cast<int64_t>(edges.size()) - 1;
    for (int64_t i = 0; i < bins; ++i) {
        if (edges[i + 1] != edges[i]) continue;
        int64_t donor = -1;
        uint32_t narrowest = ~0u;
        for (int64_t j = 0; j < bins; ++j) {
            const uint32_t f = edges[j + 1] - edges[j];
            if (f > 1 && f < narrowest) { narrowest = f; donor = j; }
        }
        if (donor < 0) return FPCC_HOST_E_CDF;
        if (donor < i) for (int64_t j = donor + 1; j <= i; ++j) --edges[j];
        else for (int64_t j = i + 1; j <= donor; ++j) ++edges[j];
    }
    std::memcpy(cdf, edges.data(), edges.size() * sizeof(uint32_t));
    return static_cast<int64_t>(edges.size());
}

int64_t fpcc_rans_indexed_encode(const int32_t *symbols, const int32_t *index, int64_t n, const uint32_t *cdf,
                                 const int64_t *cdf_start, const int64_t *cdf_len, const int32_t *offsets,
                                 int64_t n_tables, int overflow, uint8_t *out, int64_t cap) {
    if (!symbols || !cdf || !cdf_start || !cdf_len || !offsets || !out || n < 0 || n_tables < 1) return FPCC_HOST_E_ARG;
    const Tables t{cdf, cdf_start, cdf_len, offsets, n_tables};
    return overflow ? indexed_encode<true>(symbols, index, n, t, out, cap)
                    : indexed_encode<false>(symbols, index, n, t, out, cap);
}

int64_t fpcc_rans_indexed_decode(const uint8_t *stream, int64_t stream_len, const int32_t *index, int64_t n,
                                 const uint32_t *cdf, const int64_t *cdf_start, const int64_t *cdf_len,
                                 const int32_t *offsets, int64_t n_tables, int overflow, int32_t *symbols_out) {
    if (!stream || !cdf || !cdf_start || !cdf_len || !offsets || !symbols_out || n < 0 || n_tables < 1)
        return FPCC_HOST_E_ARG;
    const Tables t{cdf, cdf_start, cdf_len, offsets, n_tables};
    return overflow ? indexed_decode<true>(stream, stream_len, index, n, t, symbols_out)
                    : indexed_decode<false>(stream, stream_len, index, n, t, symbols_out);
}

int64_t fpcc_rans_binary_encode(const uint8_t *bits, const uint16_t *prob1, int64_t n, uint8_t *out, int64_t cap) {
    if (!bits || !prob1 || !out || n < 0) return FPCC_HOST_E_ARG;
    return binary_encode(bits, prob1, n, out, cap);
}

int64_t fpcc_rans_binary_decode(const uint8_t *stream, int64_t stream_len, const uint16_t *prob1, int64_t n,
                                uint8_t *bits_out) {
    if (!stream || !prob1 || !bits_out || n < 0 || stream_len < 4) return FPCC_HOST_E_ARG;
    // The symbol decision is data dependent (a coin flip for the branch predictor wherever the model is unsure), so it
    // is computed with selects; only the byte refill -- rare and regular -- stays a branch.
    const uint8_t *p = stream + 4, *end = stream + stream_len;
    uint32_t x = uint32_t(stream[0]) | uint32_t(stream[1]) << 8 | uint32_t(stream[2]) << 16 | uint32_t(stream[3]) << 24;
    if (x < kLow && n > 0) return FPCC_HOST_E_ARG;                    // not a state any encoder flushes
    for (int64_t i = 0; i < n; ++i) {
        // both successor states are formed side by side and the comparison picks one: the dependency chain through x is
        // shift -> multiply -> add -> select instead of compare -> select frequency -> multiply -> add
        const uint32_t p1 = prob1[i];
        const uint32_t split = kProbOne - p1;
        const uint32_t slot = x & (kProbOne - 1u), hi = x >> kProbBits;
        const uint32_t x0 = split * hi + slot;                    // a zero: range [0, split)
        const uint32_t x1 = p1 * hi + (slot - split);             // a one:  range [split, 65536)
        const bool one = slot >= split;
        // a conditional move, not a branch: the symbol is a coin flip for the predictor wherever the model is unsure
        // (gcc turns `one ? x1 : x0` into a branch with a multiply on either side)
#if defined(__x86_64__)
        uint32_t nx = x0;
        __asm__("cmp %[split], %[slot]\n\tcmovae %[x1], %[nx]" : [nx] "+r"(nx) : [x1] "r"(x1), [slot] "r"(slot), [split] "r"(split) : "cc");
        x = nx;
#else
        x = x0 ^ ((x0 ^ x1) & (0u - uint32_t(one)));
#endif
        bits_out[i] = static_cast<uint8_t>(one);
        for (int r = 0; r < kMaxRefill && x < kLow; ++r) {        // bounded: see FrontReader::take
            const uint32_t b = p < end ? *p : 0u;                 // past the end: zeros (garbage out, never out of bounds)
            ++p;
            x = (x << 8) | b;
        }
    }
    return FPCC_HOST_OK;
}

int64_t fpcc_rans_binary_encode_multi(const uint8_t *bits, const uint16_t *prob1, const int64_t *start,
                                      int64_t n_streams, uint8_t *out, int64_t cap_each, int64_t *len_out,
                                      int n_threads) {
    if (!bits || !prob1 || !start || !out || !len_out || n_streams < 0) return FPCC_HOST_E_ARG;
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t s = next.fetch_add(1);
            if (s >= n_streams) return;
            len_out[s] = binary_encode(bits + start[s], prob1 + start[s], start[s + 1] - start[s],
                                       out + s * cap_each, cap_each);
        }
    };
    const int64_t want = std::max<int64_t>(1, std::min<int64_t>(n_threads, n_streams));
    std::vector<std::thread> pool;
    for (int64_t k = 1; k < want; ++k) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    for (int64_t s = 0; s < n_streams; ++s) if (len_out[s] < 0) return len_out[s];
    return FPCC_HOST_OK;
}

// ---- persistent single-stream coder ----------------------------------------------------------------------------
struct fpcc_simple_enc {
    std::vector<uint8_t> store;
    BackWriter w;
    explicit fpcc_simple_enc(int64_t bytes) : store(static_cast<size_t>(bytes)), w(store.data(), bytes) {}
};

struct fpcc_simple_dec {
    FrontReader r;
    const uint8_t *base;
    fpcc_simple_dec(const uint8_t *p, int64_t n) : r(p, n), base(p) {}
};

static inline void edge_range(const uint16_t *row, int64_t width, uint32_t s, uint32_t &lo, uint32_t &hi) {
    lo = s ? row[s - 1] : 0u;
    hi = (int64_t(s) == width - 1) ? kProbOne : row[s];
}

// number of entries of a non-decreasing uint16 row that are <= target (what std::upper_bound returns as an index).  The
// decoder meets every row cold (freshly DMA-written pinned memory, 510 bytes each): a binary search walks 3-4 cache
// lines one after the other, a vector scan asks for all of them at once and has no data-dependent branch.
static inline int64_t count_le(const uint16_t *row, int64_t width, uint16_t target) {
    int64_t j = 0, count = 0;
#if defined(__AVX2__)
    const __m256i t = _mm256_set1_epi16(static_cast<short>(target));
    for (; j + 16 <= width; j += 16) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(row + j));
        const __m256i le = _mm256_cmpeq_epi16(_mm256_min_epu16(v, t), v);
        count += __builtin_popcount(static_cast<unsigned>(_mm256_movemask_epi8(le))) >> 1;
    }
#endif
    for (; j < width; ++j) count += row[j] <= target;
    return count;
}

#if defined(__x86_64__)
// the same count with 32 entries per compare where the CPU has AVX-512BW (the GPU boxes' EPYC 9575F does; selected at run time, the
// library itself is built for x86-64-v3)
__attribute__((target("avx512f,avx512bw"))) static int64_t count_le_avx512(const uint16_t *row, int64_t width, uint16_t target) {
    int64_t j = 0, count = 0;
    const __m512i t = _mm512_set1_epi16(static_cast<short>(target));
    for (; j + 32 <= width; j += 32)
        count += __builtin_popcount(_mm512_cmple_epu16_mask(_mm512_loadu_si512(row + j), t));
    if (j < width) {
        const __mmask32 tail = (__mmask32)((1u << (width - j)) - 1u);
        count += __builtin_popcount(_mm512_mask_cmple_epu16_mask(tail, _mm512_maskz_loadu_epi16(tail, row + j), t));
    }
    return count;
}
static const bool kHaveAvx512 = __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f");
#else
static const bool kHaveAvx512 = false;
static int64_t count_le_avx512(const uint16_t *row, int64_t width, uint16_t target) { return count_le(row, width, target); }
#endif

// ---- row warmers ----------------------------------------------------------------------------------------------------------------
// The 255-ary decoder meets every CDF row cold: 510 bytes that a DMA engine wrote to pinned memory a moment ago, behind a serial
// dependency (the row to search is known, the slot inside it only after the previous symbol) -- one core streaming 510 B per symbol
// from DRAM, 15.5 ns per symbol on the GPU box whatever the search costs.  A few helper threads of this library read the rows of the
// block AHEAD of the decoder (whole 4-KB pieces, round robin), which pulls them into the cache level the cores share; the decoder's
// loads then hit there.  Nothing depends on the helpers: a row they have not reached yet comes from DRAM as before.  They are pinned to
// CPUs that share a last-level cache with the calling thread (sysfs), FPCC_HOST_WARMERS = 0 .. 8 sets their number.
// Measured (tools/r05/dec_bench.py, profiles/r05/int8_stage.md): on the build container's Xeon one helper halves the time per symbol
// (62.8 -> 31.2 ns); on the GPU boxes' EPYC 9575F a single core already streams the rows at 42 GB/s and helpers change nothing
// (12.1 ns without, 11.1-13.3 ns with 1-6) in that bench, whose rows the CPU itself had just written.  In the codec the rows arrive
// by DMA from the GPU and are cold: there ONE helper takes cfg#3's decode from 19.1 to 16.9 ms (2 or 4: 18.5; tools/r05/g27.sh) in a
// process the scheduler moves between the sockets; in a process bound to the GPU's NUMA node (replicas.bind_to_device_numa_node, what
// bench.py does) it is neutral (15.3-16.2 ms without, 15.4-15.9 with; tools/r05/g47.sh) -- hence one helper by default: it costs
// nothing where the placement is right and recovers 2 ms where it is not.
namespace {
class RowWarmers {
public:
    static RowWarmers &get() { static RowWarmers w; return w; }
    int count() const { return n_; }
    // start warming [base, base + bytes); returns at once.  One decoder at a time owns the helpers (another thread's decoder that
    // arrives meanwhile simply runs without them): true = the caller owns them and must call end()
    bool begin(const void *base, size_t bytes) {
        if (n_ == 0 || bytes < (size_t)1 << 20) return false;
        if (getpid() != owner_) return false;        // a fork()ed child inherits the object but none of its threads: decode without helpers
        if (busy_.exchange(true, std::memory_order_acquire)) return false;
        pin_near_caller();
        std::lock_guard<std::mutex> g(mu_);
        base_ = static_cast<const char *>(base);
        bytes_ = bytes;
        next_.store(0, std::memory_order_relaxed);
        stop_.store(false, std::memory_order_relaxed);
        ++generation_;
        active_ = n_;
        cv_.notify_all();
        return true;
    }
    // (owner only) stop the helpers and wait until none of them touches the buffer any more: it may be freed afterwards
    void end() {
        stop_.store(true, std::memory_order_relaxed);
        {
            std::unique_lock<std::mutex> g(mu_);
            idle_.wait(g, [this] { return active_ == 0; });
            base_ = nullptr;
        }
        busy_.store(false, std::memory_order_release);
    }

private:
    RowWarmers() {
        const char *e = getenv("FPCC_HOST_WARMERS");
        n_ = e ? atoi(e) : 1;
        n_ = n_ < 0 ? 0 : (n_ > 8 ? 8 : n_);
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)n_ + 1 > hw) n_ = hw > 1 ? (int)hw - 1 : 0;
        for (int i = 0; i < n_; ++i) threads_.emplace_back([this] { run(); });
        owner_ = getpid();
    }
    ~RowWarmers() {
        { std::lock_guard<std::mutex> g(mu_); quit_ = true; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void run() {
        uint64_t seen = 0;
        for (;;) {
            const char *base;
            size_t bytes;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
                base = base_;
                bytes = bytes_;
            }
            constexpr size_t kPiece = 4096;
            unsigned sink = 0;
            while (base && !stop_.load(std::memory_order_relaxed)) {
                const size_t at = next_.fetch_add(kPiece, std::memory_order_relaxed);
                if (at >= bytes) break;
                const size_t end = at + kPiece < bytes ? at + kPiece : bytes;
                for (size_t b = at; b < end; b += 64) __builtin_prefetch(base + b, 0, 2);
                sink += static_cast<unsigned char>(*reinterpret_cast<const volatile char *>(base + end - 1));   // paces the prefetches
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                sink_ += sink;                     // (keeps the pacing loads alive)
                --active_;
            }
            idle_.notify_all();
        }
    }
    // the helpers onto CPUs that share a last-level cache with the caller (once per caller CPU)
    void pin_near_caller() {
#if defined(__linux__)
        const int cpu = sched_getcpu();
        if (cpu < 0 || cpu == pinned_for_) return;
        pinned_for_ = cpu;
        char path[128];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        FILE *f = fopen(path, "r");
        if (!f) return;
        char list[512] = {0};
        const bool ok = fgets(list, sizeof list, f) != nullptr;
        fclose(f);
        if (!ok) return;
        cpu_set_t allowed, set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
        for (char *p = list; *p;) {                                   // "a-b,c,d-e"
            char *q;
            const long lo = strtol(p, &q, 10);
            if (q == p) break;
            long hi = lo;
            if (*q == '-') { p = q + 1; hi = strtol(p, &q, 10); }
            for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c)
                if (c != cpu && CPU_ISSET(c, &allowed)) CPU_SET(c, &set);
            p = (*q == ',') ? q + 1 : q;
            if (*q != ',') break;
        }
        if (CPU_COUNT(&set) == 0) return;
        for (auto &t : threads_) pthread_setaffinity_np(t.native_handle(), sizeof set, &set);
#endif
    }
    int n_ = 0;
    pid_t owner_ = 0;
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, idle_;
    const char *base_ = nullptr;
    size_t bytes_ = 0;
    std::atomic<size_t> next_{0};
    std::atomic<bool> stop_{false}, busy_{false};
    uint64_t generation_ = 0;
    int active_ = 0;
    bool quit_ = false;
    int pinned_for_ = -1;
    unsigned sink_ = 0;
};
}  // namespace

fpcc_simple_enc *fpcc_simple_enc_new(int64_t buf_bytes) {
    if (buf_bytes < 16) return nullptr;
    return new (std::nothrow) fpcc_simple_enc(buf_bytes);
}
void fpcc_simple_enc_free(fpcc_simple_enc *e) { delete e; }

int64_t fpcc_simple_enc_push(fpcc_simple_enc *e, const uint16_t *rows, int64_t n_rows, int64_t width,
                             const uint16_t *symbols, int64_t n) {
    if (!e || !rows || !symbols || width < 1 || (n_rows != 1 && n_rows != n)) return FPCC_HOST_E_ARG;
    for (int64_t i = n - 1; i >= 0; --i) {
        const uint16_t *row = rows + (n_rows == 1 ? 0 : i * width);
        const uint32_t s = symbols[i];
        if (int64_t(s) >= width) return FPCC_HOST_E_ARG;
        uint32_t lo, hi;
        edge_range(row, width, s, lo, hi);
        if (hi <= lo) return FPCC_HOST_E_ARG;
        e->w.put<kProbBits>(lo, hi - lo);
    }
    return e->w.full() ? int64_t(FPCC_HOST_E_BUFFER) : e->w.buffered();
}

int64_t fpcc_simple_enc_push_bin(fpcc_simple_enc *e, const uint16_t *edge, int64_t n_rows, const uint8_t *bits, int64_t n) {
    if (!e || !edge || !bits || (n_rows != 1 && n_rows != n)) return FPCC_HOST_E_ARG;
    for (int64_t i = n - 1; i >= 0; --i) {
        const uint32_t c = edge[n_rows == 1 ? 0 : i];
        if (bits[i]) e->w.put<kProbBits>(c, kProbOne - c);
        else e->w.put<kProbBits>(0u, c);
    }
    return e->w.full() ? int64_t(FPCC_HOST_E_BUFFER) : e->w.buffered();
}

int64_t fpcc_simple_enc_push_ranges(fpcc_simple_enc *e, const uint16_t *start, const uint16_t *freq_m1, int64_t n) {
    if (!e || !start || !freq_m1) return FPCC_HOST_E_ARG;
    if (e->w.room(2 * n + 8)) {
        for (int64_t i = n - 1; i >= 0; --i) e->w.put_roomy<kProbBits>(start[i], uint32_t(freq_m1[i]) + 1u);
    } else {
        for (int64_t i = n - 1; i >= 0; --i) e->w.put<kProbBits>(start[i], uint32_t(freq_m1[i]) + 1u);
    }
    return e->w.full() ? int64_t(FPCC_HOST_E_BUFFER) : e->w.buffered();
}

int64_t fpcc_simple_enc_finish(fpcc_simple_enc *e, uint8_t *out, int64_t cap) {
    if (!e || !out) return FPCC_HOST_E_ARG;
    int64_t n = e->w.finish();
    if (n >= 0) {
        if (n > cap) n = FPCC_HOST_E_BUFFER;
        else std::memcpy(out, e->w.head(), static_cast<size_t>(n));
    }
    e->w.reset();
    return n;
}

fpcc_simple_dec *fpcc_simple_dec_new(const uint8_t *stream, int64_t stream_len) {
    if (!stream || stream_len < 4) return nullptr;
    auto *d = new (std::nothrow) fpcc_simple_dec(stream, stream_len);
    if (d && !d->r.valid()) { delete d; return nullptr; }              // initial state below 2^23: corrupt stream
    return d;
}
void fpcc_simple_dec_free(fpcc_simple_dec *d) { delete d; }

int64_t fpcc_simple_dec_pop(fpcc_simple_dec *d, const uint16_t *rows, int64_t n_rows, int64_t width,
                            uint16_t *symbols_out, int64_t n) {
    if (!d || !rows || !symbols_out || width < 1 || (n_rows != 1 && n_rows != n)) return FPCC_HOST_E_ARG;
    constexpr int64_t kAhead = 6;                       // rows requested ahead of the one being searched
    const int64_t row_bytes = width * 2;
    const bool wide = kHaveAvx512 && width >= 64;
    const bool warm = n_rows != 1 && n >= 2048;         // a block of cold rows: helpers read ahead of this thread (RowWarmers)
    struct WarmEnd { bool on; ~WarmEnd() { if (on) RowWarmers::get().end(); } }
        warm_end{warm && RowWarmers::get().begin(rows, static_cast<size_t>(n) * static_cast<size_t>(row_bytes))};
    for (int64_t i = 0; i < n; ++i) {
        const uint16_t *row = rows + (n_rows == 1 ? 0 : i * width);
        if (n_rows != 1 && i + kAhead < n) {
            const char *ahead = reinterpret_cast<const char *>(row + kAhead * width);
            for (int64_t b = 0; b < row_bytes; b += 64) __builtin_prefetch(ahead + b, 0, 0);
        }
        const uint32_t target = d->r.peek<kProbBits>();
        int64_t s = wide ? count_le_avx512(row, width, static_cast<uint16_t>(target))
                         : width >= 32 ? count_le(row, width, static_cast<uint16_t>(target))
                                       : std::upper_bound(row, row + width, static_cast<uint16_t>(target)) - row;
        s = std::min<int64_t>(s, width - 1);
        uint32_t lo, hi;
        edge_range(row, width, static_cast<uint32_t>(s), lo, hi);
        d->r.take_fast<kProbBits>(lo, hi - lo);
        symbols_out[i] = static_cast<uint16_t>(s);
    }
    return FPCC_HOST_OK;
}

int64_t fpcc_simple_dec_tell(const fpcc_simple_dec *d, uint32_t *state_out, int64_t *position_out) {
    if (!d || !state_out || !position_out) return FPCC_HOST_E_ARG;
    *state_out = d->r.state();
    *position_out = d->r.cursor() - d->base;
    return FPCC_HOST_OK;
}

int64_t fpcc_simple_dec_pop_bin(fpcc_simple_dec *d, const uint16_t *edge, int64_t n_rows, uint8_t *bits_out, int64_t n) {
    if (!d || !edge || !bits_out || (n_rows != 1 && n_rows != n)) return FPCC_HOST_E_ARG;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t c = edge[n_rows == 1 ? 0 : i];
        const bool one = d->r.peek<kProbBits>() >= c;
        if (one) d->r.take<kProbBits>(c, kProbOne - c);
        else d->r.take<kProbBits>(0u, c);
        bits_out[i] = one;
    }
    return FPCC_HOST_OK;
}

// ---- background coder pool ----------------------------------------------------------------------------------------
// Jobs are ordinary calls of the coders above, started once a host-visible flag has the expected value.  The caller makes
// the flag the LAST of a stream-ordered group of device->host copies into pinned memory, so a job begins the moment its
// inputs have landed while the GPU keeps working on later levels; no HIP call is made from this library.
struct fpcc_pool {
    std::vector<std::thread> threads;
    std::deque<std::function<int64_t()>> jobs;
    std::mutex mu;
    std::condition_variable cv_job, cv_idle;
    int64_t pending = 0;
    int64_t first_error = 0;
    bool closing = false;

    explicit fpcc_pool(int n) {
        for (int i = 0; i < n; ++i) threads.emplace_back([this] { run(); });
    }
    ~fpcc_pool() {
        { std::lock_guard<std::mutex> g(mu); closing = true; }
        cv_job.notify_all();
        for (auto &t : threads) t.join();
    }
    void run() {
        for (;;) {
            std::function<int64_t()> job;
            {
                std::unique_lock<std::mutex> g(mu);
                cv_job.wait(g, [this] { return closing || !jobs.empty(); });
                if (jobs.empty()) return;
                job = std::move(jobs.front());
                jobs.pop_front();
            }
            const int64_t rc = job();
            {
                std::lock_guard<std::mutex> g(mu);
                if (rc < 0 && first_error == 0) first_error = rc;
                --pending;
            }
            cv_idle.notify_all();
        }
    }
    void submit(std::function<int64_t()> job) {
        { std::lock_guard<std::mutex> g(mu); jobs.push_back(std::move(job)); ++pending; }
        cv_job.notify_one();
    }
};

namespace {
constexpr int64_t kFlagTimeoutUs = 60ll * 1000 * 1000;

bool await_flag(const volatile uint32_t *flag, uint32_t ready) {
    if (!flag) return true;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
        // an acquire load of the word the producer stores last (a DMA engine in production, a thread in the tests): the job's
        // reads of its inputs cannot be moved ahead of it
        if (__atomic_load_n(const_cast<const uint32_t *>(flag), __ATOMIC_ACQUIRE) == ready) return true;
        if ((spin & 63u) == 63u) {
            std::this_thread::yield();
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >
                kFlagTimeoutUs) return false;
        }
    }
}
}  // namespace

fpcc_pool *fpcc_pool_new(int n_threads) {
    if (n_threads < 1 || n_threads > 256) return nullptr;
    return new (std::nothrow) fpcc_pool(n_threads);
}
void fpcc_pool_free(fpcc_pool *p) { delete p; }

int64_t fpcc_pool_binary_encode(fpcc_pool *p, const volatile uint32_t *flag, uint32_t ready, const uint8_t *bits,
                                const uint16_t *prob1, int64_t n, uint8_t *out, int64_t cap, int64_t *len_out) {
    if (!p || !bits || !prob1 || !out || !len_out || n < 0) return FPCC_HOST_E_ARG;
    p->submit([=]() -> int64_t {
        if (!await_flag(flag, ready)) return *len_out = FPCC_HOST_E_TIMEOUT;
        return *len_out = binary_encode(bits, prob1, n, out, cap);
    });
    return FPCC_HOST_OK;
}

int64_t fpcc_pool_histogram_encode(fpcc_pool *p, const volatile uint32_t *flag, uint32_t ready, const int32_t *symbols,
                                   int64_t n, int fixed_offset, int32_t *offset_io, uint32_t *cdf_out, int64_t cdf_cap,
                                   int64_t *cdf_len_out, uint8_t *out, int64_t cap, int64_t *len_out) {
    if (!p || !symbols || !offset_io || !cdf_out || !cdf_len_out || !out || !len_out || n < 1 || cdf_cap < 2)
        return FPCC_HOST_E_ARG;
    p->submit([=]() -> int64_t {
        if (!await_flag(flag, ready)) return *len_out = FPCC_HOST_E_TIMEOUT;
        int32_t lo = symbols[0], hi = symbols[0];
        for (int64_t i = 1; i < n; ++i) { lo = std::min(lo, symbols[i]); hi = std::max(hi, symbols[i]); }
        if (fixed_offset) {
            if (lo < *offset_io) return *len_out = FPCC_HOST_E_ARG;
            lo = *offset_io;
        }
        const int64_t bins = int64_t(hi) - lo + 1;
        if (bins + 1 > cdf_cap) return *len_out = FPCC_HOST_E_BUFFER;
        std::vector<double> hist(static_cast<size_t>(bins), 0.0);
        for (int64_t i = 0; i < n; ++i) hist[static_cast<size_t>(symbols[i] - lo)] += 1.0;
        const int64_t m = fpcc_pmf_to_quantized_cdf(hist.data(), bins, 0, nullptr, cdf_out);
        if (m < 0) return *len_out = m;
        *cdf_len_out = m;
        *offset_io = lo;
        const int64_t start = 0, len = m;
        const Tables t{cdf_out, &start, &len, offset_io, 1};
        return *len_out = indexed_encode<false>(symbols, nullptr, n, t, out, cap);
    });
    return FPCC_HOST_OK;
}

int64_t fpcc_pool_table_decode(fpcc_pool *p, const uint8_t *stream, int64_t stream_len, int64_t n, const uint32_t *cdf,
                               int64_t cdf_len, int32_t offset, int32_t *symbols_out, int64_t first_chunk,
                               int64_t *progress) {
    if (!p || !stream || !cdf || !symbols_out || !progress || n < 0 || cdf_len < 2 || cdf_len - 1 > 65535 || stream_len < 4)
        return FPCC_HOST_E_ARG;
    if (!cdf_well_formed(cdf, cdf_len)) return FPCC_HOST_E_CDF;       // the table comes straight from the bitstream
    static_assert(sizeof(std::atomic<int64_t>) == sizeof(int64_t), "progress counter must be a plain 64-bit word");
    auto *prog = reinterpret_cast<std::atomic<int64_t> *>(progress);
    prog->store(0, std::memory_order_relaxed);
    p->submit([=]() -> int64_t {
        const int64_t rc = single_table_decode(stream, stream_len, n, cdf, static_cast<int32_t>(cdf_len) - 1, offset,
                                               symbols_out, prog, first_chunk);
        prog->store(rc < 0 ? rc : n, std::memory_order_release);
        return rc;
    });
    return FPCC_HOST_OK;
}

int64_t fpcc_pool_binary_decode(fpcc_pool *p, const uint8_t *stream, int64_t stream_len, const uint16_t *prob1, int64_t n,
                                uint8_t *bits_out, int64_t *done) {
    if (!p || !stream || !prob1 || !bits_out || !done || n < 0 || stream_len < 4) return FPCC_HOST_E_ARG;
    auto *flag = reinterpret_cast<std::atomic<int64_t> *>(done);
    flag->store(0, std::memory_order_relaxed);
    p->submit([=]() -> int64_t {
        const int64_t rc = fpcc_rans_binary_decode(stream, stream_len, prob1, n, bits_out);
        flag->store(rc < 0 ? rc : 1, std::memory_order_release);
        return rc;
    });
    return FPCC_HOST_OK;
}

int64_t fpcc_progress_wait(const int64_t *progress, int64_t needed) {
    if (!progress) return FPCC_HOST_E_ARG;
    const auto *prog = reinterpret_cast<const std::atomic<int64_t> *>(progress);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
        const int64_t v = prog->load(std::memory_order_acquire);
        if (v < 0) return v;
        if (v >= needed) return v;
        if ((spin & 63u) == 63u) {
            std::this_thread::yield();
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >
                kFlagTimeoutUs) return FPCC_HOST_E_TIMEOUT;
        }
    }
}

int64_t fpcc_pool_wait(fpcc_pool *p) {
    if (!p) return FPCC_HOST_E_ARG;
    std::unique_lock<std::mutex> g(p->mu);
    p->cv_idle.wait(g, [p] { return p->pending == 0; });
    const int64_t rc = p->first_error;
    p->first_error = 0;
    return rc;
}

}  // extern "C"
