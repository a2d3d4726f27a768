// Device-side rANS DECODERS of the two stream formats the codecs decode level by level:
//   binary coder      (lossy_coord_v2 occupancy levels; /root/reference/lib/entropy_models/rans_coder/rans_wrapper.cpp:385-428)
//   255-ary row coder (lossl_coord_int; /root/reference/models/convolutional/lossy_coord_v3/rans_coder/simple_rans_wrapper.cpp:206-239)
// Same streams, same arithmetic as libfpcc_host (32-bit state, L = 2^23, byte renormalisation, 16-bit probabilities).
//
// A stream has ONE state, so its symbols decode one after the other whatever the hardware: a launch is ONE wave.  What the
// 64 lanes add is everything around the serial chain: probabilities / CDF rows are fetched 64 symbols (or one row) at a time
// with coalesced loads, the byte stream sits in a 256-byte register window, the symbol search of the 255-ary coder is one
// compare per lane plus a ballot, results are stored 64 at a time.  The chain itself runs on the scalar unit (the state,
// the stream position and the current probability are wave-uniform).  Measured against the host path in
// profiles/r02/device_rans.md -- the point of these kernels is that a level's symbols never leave the GPU (no D2H of
// probabilities or of 510-byte CDF rows, no H2D of the decoded symbols); the serial chain is slower than a host core's.
#include "common.h"

namespace fpcc {
namespace {

constexpr uint32_t kLow = 1u << 23;
constexpr uint32_t kOne = 1u << 16;

// 256-byte window of the byte stream held by the wave: lane l keeps bytes [4l, 4l + 4) from `base`
struct Window {
    const uint8_t *stream;
    int64_t len;
    int64_t base;      // stream offset of the window's first byte (wave-uniform)
    uint32_t word;     // this lane's four bytes

    __device__ __forceinline__ void load(int64_t at, int lane) {
        base = at;
        const int64_t p = at + 4 * lane;
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (p + b < len) w |= (uint32_t)stream[p + b] << (8 * b);      // past the end: zeros, as on the host
        word = w;
    }
    // byte at stream offset `pos` (wave-uniform); reloads when pos leaves the window
    __device__ __forceinline__ uint32_t byte(int64_t pos, int lane) {
        if (pos - base >= 256 || pos < base) load(pos, lane);
        const int idx = (int)(pos - base);
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)word, idx >> 2);
        return (w >> (8 * (idx & 3))) & 0xffu;
    }
};

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// state[0] = x, state[1..2] = stream position (low, high); one wave
__global__ __launch_bounds__(64) void k_rans_binary_decode(const uint8_t *__restrict__ stream, int64_t stream_len,
                                                           const uint16_t *__restrict__ prob1, int64_t n,
                                                           uint8_t *__restrict__ bits, int32_t *__restrict__ ones_out,
                                                           int32_t *__restrict__ status) {
    const int lane = threadIdx.x;
    Window win{stream, stream_len, 0, 0};
    win.load(0, lane);
    uint32_t x = uniform((uint32_t)__builtin_amdgcn_readlane((int)win.word, 0));
    if (x < kLow && n > 0) {                       // not a state any encoder flushes (same rule as the host decoder)
        if (lane == 0) *status = -2;
        return;
    }
    int64_t pos = 4;
    int ones = 0;
    for (int64_t i0 = 0; i0 < n; i0 += 64) {
        const int64_t i = i0 + lane;
        const uint32_t pv = i < n ? prob1[i] : 1u;
        const int m = n - i0 < 64 ? (int)(n - i0) : 64;
        unsigned long long mask = 0;
        for (int j = 0; j < m; ++j) {
            const uint32_t p1 = (uint32_t)__builtin_amdgcn_readlane((int)pv, j);
            const uint32_t split = kOne - p1;
            const uint32_t slot = x & (kOne - 1u);
            const bool one = slot >= split;
            const uint32_t freq = one ? p1 : split;
            const uint32_t start = one ? split : 0u;
            x = freq * (x >> 16) + slot - start;
            mask |= (unsigned long long)one << j;
#pragma unroll 1
            for (int r = 0; r < 3 && x < kLow; ++r) {       // bounded refill, as on the host
                x = (x << 8) | win.byte(pos, lane);
                ++pos;
            }
        }
        if (i < n) bits[i] = (uint8_t)((mask >> lane) & 1ull);
        ones += __popcll(mask);
    }
    if (lane == 0) {
        if (ones_out) *ones_out = ones;
        *status = 0;
    }
}

// 255-ary coder with one uint16 CDF row per symbol (row[j] = upper edge of symbol j, the last edge is implied 65536).
// state: int32[4] = {x, pos_lo, pos_hi, status}; persists between launches (one stream is decoded level by level).
template <int UNUSED>
__global__ __launch_bounds__(64) void k_simple_dec_pop(int32_t *__restrict__ state, const uint8_t *__restrict__ stream,
                                                       int64_t stream_len, const uint16_t *__restrict__ rows, int64_t n_rows,
                                                       int width, uint16_t *__restrict__ symbols, int64_t n,
                                                       int32_t *__restrict__ children_out) {
    const int lane = threadIdx.x;
    uint32_t x = (uint32_t)state[0];
    int64_t pos = ((int64_t)(uint32_t)state[2] << 32) | (uint32_t)state[1];
    Window win{stream, stream_len, 0, 0};
    win.load(pos, lane);
    // lane l holds edges [4l, 4l + 4) of the current row (65536 past the row: never <= slot), one row prefetched
    auto load_row = [&](int64_t i, uint32_t (&e)[4]) {
        const uint16_t *row = rows + (n_rows == 1 ? 0 : i * (int64_t)width);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = 4 * lane + c;
            e[c] = (i < n && j < width - 1) ? (uint32_t)row[j] : kOne;      // the last symbol's upper edge is 65536
        }
    };
    uint32_t cur[4], nxt[4];
    load_row(0, cur);
    int children = 0;
    // sticky status word state[3]: 1 = state below the renormalisation bound on entry (not a rANS state), 2 = a CDF row that is
    // not increasing at the decoded symbol, 4 = read past the end of the stream (beyond the zero slack a valid stream may touch)
    int bad = x < kLow ? 1 : 0;
    uint32_t sym_keep = 0;                               // lane (i % 64) keeps symbol i until the 64-wide store
    for (int64_t i = 0; i < n; ++i) {
        load_row(i + 1, nxt);
        const uint32_t slot = x & (kOne - 1u);
        // number of edges <= slot = the symbol (edges are non-decreasing); clamp as the reference does
        int s = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) s += __popcll(__ballot(cur[c] <= slot));
        if (s > width - 1) s = width - 1;
        // lo = edge[s - 1] (0 for s == 0), hi = edge[s] (65536 for the last symbol): two lane reads
        uint32_t lo = 0, hi = kOne;
        {
            const int a = s - 1, b = s;
            uint32_t va = 0, vb = kOne;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t ra = (uint32_t)__builtin_amdgcn_readlane((int)cur[c], a >= 0 ? a >> 2 : 0);
                const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane((int)cur[c], b >> 2);
                if ((a & 3) == c && a >= 0) va = ra;
                if ((b & 3) == c) vb = rb;
            }
            lo = va;
            hi = vb;
        }
        bad |= hi <= lo ? 2 : 0;
        x = (hi - lo) * (x >> 16) + slot - lo;
#pragma unroll 1
        for (int r = 0; r < 3 && x < kLow; ++r) {
            x = (x << 8) | win.byte(pos, lane);
            ++pos;
        }
        children += __popc((unsigned)(s + 1) & 0xffu);
        if ((int)(i & 63) == lane) sym_keep = (uint32_t)s;
        if ((i & 63) == 63 || i == n - 1) {
            const int64_t at = (i & ~63ll) + lane;
            if (at <= i) symbols[at] = (uint16_t)sym_keep;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = nxt[c];
    }
    if (lane == 0) {
        state[0] = (int32_t)x;
        state[1] = (int32_t)(uint32_t)pos;
        state[2] = (int32_t)(uint32_t)(pos >> 32);
        state[3] |= bad | (pos > stream_len + 4 ? 4 : 0);
        if (children_out) { children_out[0] = children; children_out[1] = state[3]; }
    }
}

}  // namespace
}  // namespace fpcc

using namespace fpcc;

extern "C" int fpcc_rans_binary_decode_dev(const uint8_t *stream, int64_t stream_len, const uint16_t *prob1, int64_t n,
                                           uint8_t *bits_out, int32_t *ones_out, int32_t *status, void *hip_stream) {
    if (n < 0 || stream_len < 4) return fail_arg("rans_binary_decode_dev: bad sizes");
    if (!stream || !status || (n > 0 && (!prob1 || !bits_out))) return fail_arg("rans_binary_decode_dev: null pointer");
    hipLaunchKernelGGL(k_rans_binary_decode, dim3(1), dim3(64), 0, as_stream(hip_stream), stream, stream_len, prob1, n, bits_out,
                       ones_out, status);
    return check_hip(hipGetLastError(), "k_rans_binary_decode");
}

extern "C" int fpcc_simple_dec_pop_dev(int32_t *state, const uint8_t *stream, int64_t stream_len, const uint16_t *rows,
                                       int64_t n_rows, int64_t width, uint16_t *symbols_out, int64_t n, int32_t *children_out,
                                       void *hip_stream) {
    if (n < 0 || stream_len < 4 || width < 1 || width > 256 || (n_rows != 1 && n_rows != n))
        return fail_arg("simple_dec_pop_dev: bad sizes (rows of at most 256 symbols, one row or one per symbol)");
    if (n == 0) return FPCC_OK;
    if (!state || !stream || !rows || !symbols_out) return fail_arg("simple_dec_pop_dev: null pointer");
    hipLaunchKernelGGL(k_simple_dec_pop<0>, dim3(1), dim3(64), 0, as_stream(hip_stream), state, stream, stream_len, rows, n_rows,
                       (int)width, symbols_out, n, children_out);
    return check_hip(hipGetLastError(), "k_simple_dec_pop");
}
