// Entropy-model glue kernels: everything between the feature tensors and the (host) rANS coders that the reference
// does with small PyTorch ops and device->host copies.  All are streaming, HBM-bound passes.
#include <cstring>

#include <algorithm>

#include "common.h"

namespace fpcc {
namespace {

constexpr int kThreads = 256;

// The logistic function, SPECIFIED (numerics version 3): exp(-x) by Cody-Waite reduction and a degree-5 polynomial; every operation is
// an IEEE-754 binary32 fma / multiplication / addition / round-to-nearest-even / correctly rounded division, written so that no
// compiler may contract or reorder it.  Any conforming device or host gives the same bits (oracle/sparse_conv.c:sigmoid_spec is the
// same text), so a stream written here decodes anywhere -- the device's expf and a host's libm differ in the last bit now and then, and
// one differing 16-bit probability desynchronises a binary rANS stream.  Within 2 ulp of torch.sigmoid (the reference's call).
__device__ __forceinline__ float sigmoid_spec(float x) {
    float t = -x;
    t = fminf(fmaxf(t, -87.0f), 87.0f);
    const float n = rintf(__fmul_rn(t, 1.44269504088896341f));
    float r = __fmaf_rn(n, -0.693145751953125f, t);
    r = __fmaf_rn(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __fmaf_rn(p, r, 1.3981999507e-3f);
    p = __fmaf_rn(p, r, 8.3334519073e-3f);
    p = __fmaf_rn(p, r, 4.1665795894e-2f);
    p = __fmaf_rn(p, r, 1.6666665459e-1f);
    p = __fmaf_rn(p, r, 5.0000001201e-1f);
    const float pr = __fmul_rn(p, r);
    float e = __fmaf_rn(pr, r, r);
    e = __fadd_rn(e, 1.0f);
    e = __fmul_rn(e, __uint_as_float((uint32_t)((int32_t)n + 127) << 23));
    return __fdiv_rn(1.0f, __fadd_rn(1.0f, e));
}

__global__ void k_logit_to_prob16(const float *__restrict__ x, int64_t n, uint16_t *__restrict__ p) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = sigmoid_spec(x[i]);
    double q = rint((double)s * 65536.0);                     // float64 product, round half to even (np.round)
    q = fmin(fmax(q, 1.0), 65535.0);
    p[i] = (uint16_t)q;
}

__global__ void k_quantize(float *__restrict__ x, int64_t n, float scale, int32_t *__restrict__ sym) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = rintf(x[i] * scale);
    if (sym) sym[i] = (int32_t)v;
    x[i] = v / scale;
}

__global__ void k_child_mask(const int32_t *__restrict__ child_row, int64_t n, uint8_t *__restrict__ mask) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) mask[i] = child_row[i] >= 0 ? 1 : 0;
}

__device__ __forceinline__ float max8(const float4 a, const float4 b) {
    return fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
}

// --- variant whose "cell" is any run of candidate groups (e.g. all descendants of a stride-4 voxel) -------------------
__device__ __forceinline__ unsigned f2ord(float f) {     // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ void k_seg_max(const float4 *__restrict__ logit, int64_t m, const int32_t *__restrict__ seg, unsigned *__restrict__ seg_max) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= m) return;
    atomicMax(&seg_max[seg[p]], f2ord(max8(logit[2 * p], logit[2 * p + 1])));   // exact and order independent
}

// ---- k-th smallest candidate by radix SELECT (three histogram passes over 11 + 11 + 10 key bits) instead of a full sort --------------
// The threshold of the decoder's adaptive pruning is ONE order statistic of the 8 m candidates (lossy_coord_v2/layers.py:164-180: kthvalue);
// rounds 1-5 sorted all of them (rocprim radix sort: ~40 bytes moved per candidate and a ranked copy).  A pass reads the logits once
// (32 bytes per group of 8), recomputes "is this candidate its cell's maximum" on the fly (maxima rank as +inf) and histograms the keys
// that still match the prefix found so far in LDS; a one-workgroup kernel then walks the 2048 bins.  Exact: an order statistic does not
// depend on how it is found.
struct SelectState {
    unsigned long long rank;      // 1-based rank still to find among the keys matching `prefix`; 0 = no threshold (keep everything)
    unsigned prefix;              // the key bits fixed so far (right-aligned)
    float thr;                    // the result, after the last pass
};
constexpr int kSelBins = 2048;

// one pass: histogram of the key bits [kShift, kShift + 11) (10 in the last pass) of the candidates whose higher bits equal the prefix
// found so far.  PASS 0 takes the rank as an argument.  One workgroup per CU at most: every workgroup merges its bins into the global
// histogram with atomics, and on real logits most candidates share a few dozen bins (sign + exponent) -- ~12 ns per atomic on one address.
template <int PASS, bool SEG>
__global__ __launch_bounds__(256) void k_select_hist(const float4 *__restrict__ logit, int64_t m, const int32_t *__restrict__ seg,
                                                     const unsigned *__restrict__ seg_max, const SelectState *__restrict__ st,
                                                     unsigned *__restrict__ hist, unsigned long long rank0) {
    __shared__ unsigned s_h[kSelBins];
    for (int i = threadIdx.x; i < kSelBins; i += 256) s_h[i] = 0;
    __syncthreads();
    const unsigned long long rank = PASS ? st->rank : rank0;
    const unsigned prefix = PASS ? st->prefix : 0u;
    // PASS 0: bits 31..21, PASS 1: bits 20..10 under an 11-bit prefix, PASS 2: bits 9..0 under a 22-bit prefix
    constexpr int kShift = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
    constexpr unsigned kMask = PASS == 2 ? 1023u : 2047u;
    if (rank != 0ull)
        for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < m; p += (int64_t)gridDim.x * 256) {
            const float4 a = logit[2 * p], b = logit[2 * p + 1];
            const float mx = SEG ? ord2f(seg_max[seg[p]]) : max8(a, b);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned key = f2ord(v[k] == mx ? __builtin_huge_valf() : v[k]);
                if (PASS == 0 || (key >> (kShift + (PASS == 2 ? 10 : 11))) == prefix) atomicAdd(&s_h[(key >> kShift) & kMask], 1u);
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < kSelBins; i += 256)
        if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

// one workgroup after each pass (its own launch: the kernel boundary is what orders it behind every workgroup's atomics -- a
// last-arriver scheme inside the pass kernel raced on this hardware and a device-scope fence per workgroup cost ~50 us): finds the bin
// that holds the wanted rank by a scan over 256 chunk sums, extends the prefix, clears the bins for the next pass
template <int PASS>
__global__ __launch_bounds__(256) void k_select_pick(unsigned *__restrict__ hist, SelectState *__restrict__ st, unsigned long long rank0) {
    __shared__ unsigned long long s_part[256];
    const unsigned long long rank = PASS ? st->rank : rank0;
    const unsigned prefix = PASS ? st->prefix : 0u;
    if (rank == 0ull) {                                                         // no threshold: everything above -inf is kept
        if (threadIdx.x == 0) { st->rank = 0ull; if (PASS == 2) st->thr = -__builtin_huge_valf(); }
        return;
    }
    unsigned h8[8];                                                              // 8 bins per thread, ascending
#pragma unroll
    for (int j = 0; j < 8; ++j) h8[j] = hist[threadIdx.x * 8 + j];
    unsigned long long mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) mine += h8[j];
    // inclusive scan of the 256 chunk sums (Hillis-Steele in LDS), then the one thread whose chunk holds the rank walks its 8 bins
    s_part[threadIdx.x] = mine;
    __syncthreads();
    unsigned long long incl = mine;
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned long long other = (int)threadIdx.x >= d ? s_part[threadIdx.x - d] : 0ull;
        __syncthreads();
        incl += other;
        s_part[threadIdx.x] = incl;
        __syncthreads();
    }
    const unsigned long long excl = incl - mine;
    // (a rank beyond the total -- cannot happen for 1 <= rank <= 8 m -- would fall to the last chunk's last bin)
    const bool holder = (excl < rank && rank <= incl) || (threadIdx.x == 255 && rank > incl);
    if (holder) {
        unsigned long long before = excl;
        int bin = (int)threadIdx.x * 8 + 7;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (before + h8[j] >= rank) { bin = (int)threadIdx.x * 8 + j; break; }
            before += h8[j];
        }
        st->rank = rank - before;
        const unsigned np = PASS == 0 ? (unsigned)bin : ((prefix << (PASS == 2 ? 10 : 11)) | (unsigned)bin);
        st->prefix = np;
        if (PASS == 2) st->thr = ord2f(np);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) hist[threadIdx.x * 8 + j] = 0;                  // clean for the next pass
}

template <bool SEG>
__global__ void k_keep_thr(const float4 *__restrict__ logit, int64_t m, const int32_t *__restrict__ seg, const unsigned *__restrict__ seg_max,
                           const SelectState *__restrict__ st, uint8_t *__restrict__ keep) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= m) return;
    const float thr = st->thr;
    const float4 a = logit[2 * p], b = logit[2 * p + 1];
    const float mx = SEG ? ord2f(seg_max[seg[p]]) : max8(a, b);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint64_t bits = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) bits |= (uint64_t)((v[k] > thr || v[k] == mx) ? 1 : 0) << (8 * k);
    reinterpret_cast<uint64_t *>(keep)[p] = bits;
}

// the three passes + the final mask; ws = SelectState (padded to 256 bytes) + 2048 bins
template <bool SEG>
int select_and_keep(const float *logit, int64_t m, const int32_t *seg, const unsigned *seg_max, int64_t kth, uint8_t *keep_out, void *ws,
                    hipStream_t s) {
    SelectState *st = static_cast<SelectState *>(ws);
    unsigned *hist = reinterpret_cast<unsigned *>(static_cast<char *>(ws) + 256);
    FPCC_HIP(hipMemsetAsync(ws, 0, 256 + 4 * kSelBins, s));
    const float4 *lg = reinterpret_cast<const float4 *>(logit);
    const unsigned blocks = (unsigned)std::min<int64_t>(blocks_for(m, 256), 256);
    const unsigned long long rank = kth >= 1 ? (unsigned long long)kth : 0ull;
    hipLaunchKernelGGL((k_select_hist<0, SEG>), dim3(blocks), dim3(256), 0, s, lg, m, seg, seg_max, (const SelectState *)st, hist, rank);
    hipLaunchKernelGGL(k_select_pick<0>, dim3(1), dim3(256), 0, s, hist, st, rank);
    hipLaunchKernelGGL((k_select_hist<1, SEG>), dim3(blocks), dim3(256), 0, s, lg, m, seg, seg_max, (const SelectState *)st, hist, 0ull);
    hipLaunchKernelGGL(k_select_pick<1>, dim3(1), dim3(256), 0, s, hist, st, 0ull);
    hipLaunchKernelGGL((k_select_hist<2, SEG>), dim3(blocks), dim3(256), 0, s, lg, m, seg, seg_max, (const SelectState *)st, hist, 0ull);
    hipLaunchKernelGGL(k_select_pick<2>, dim3(1), dim3(256), 0, s, hist, st, 0ull);
    hipLaunchKernelGGL(k_keep_thr<SEG>, dim3(blocks_for(m, kThreads)), dim3(kThreads), 0, s, lg, m, seg, seg_max, (const SelectState *)st, keep_out);
    return check_hip(hipGetLastError(), "topk select");
}

}  // namespace
}  // namespace fpcc

using namespace fpcc;

extern "C" int fpcc_logit_to_prob16(const float *logit, int64_t n, uint16_t *prob_out, void *stream) {
    if (n < 0 || (n > 0 && (!logit || !prob_out))) return fail_arg("logit_to_prob16: null pointer");
    if (n == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_logit_to_prob16, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), logit, n,
                       prob_out);
    FPCC_LAUNCHED(k_logit_to_prob16);
    return FPCC_OK;
}

extern "C" int fpcc_quantize_symbols(float *x, int64_t n, float scale, int32_t *symbols_out, void *stream) {
    if (n < 0 || (n > 0 && !x) || !(scale > 0.0f)) return fail_arg("quantize_symbols: null pointer or scale <= 0");
    if (n == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_quantize, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), x, n, scale,
                       symbols_out);
    FPCC_LAUNCHED(k_quantize);
    return FPCC_OK;
}

extern "C" int fpcc_child_mask(const int32_t *child_row, int64_t m, uint8_t *mask_out, void *stream) {
    if (m < 0 || (m > 0 && (!child_row || !mask_out))) return fail_arg("child_mask: null pointer");
    if (m == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_child_mask, dim3(blocks_for(8 * m, kThreads)), dim3(kThreads), 0, as_stream(stream), child_row,
                       8 * m, mask_out);
    FPCC_LAUNCHED(k_child_mask);
    return FPCC_OK;
}

static int64_t select_ws_bytes() { return 256 + 4 * kSelBins; }

extern "C" int64_t fpcc_topk_keep(const float *logit, int64_t m, int64_t target, uint8_t *keep_out, void *ws,
                                  int64_t ws_bytes, void *stream) {
    if (m < 0 || target < 0) return fail_arg("topk_keep: negative size");
    const int64_t need = select_ws_bytes();
    if (!ws) return need;
    if (ws_bytes < need) { set_error("topk_keep: workspace %lld < %lld", (long long)ws_bytes, (long long)need); return FPCC_E_WORKSPACE; }
    if (m == 0) return FPCC_OK;
    if (!logit || !keep_out) return fail_arg("topk_keep: null pointer");
    if ((reinterpret_cast<uintptr_t>(logit) & 15) || (reinterpret_cast<uintptr_t>(keep_out) & 7) || (reinterpret_cast<uintptr_t>(ws) & 15))
        return fail_arg("topk_keep: logit / workspace must be 16-byte and keep_out 8-byte aligned");
    // the k-th smallest (1-based) of the candidates ranked with their cell maxima at +inf is the threshold
    return select_and_keep<false>(logit, m, nullptr, nullptr, 8 * m - target, keep_out, ws, as_stream(stream));
}

extern "C" int64_t fpcc_topk_keep_cells(const float *logit, int64_t m, const int32_t *cell_of_group, int64_t n_cells,
                                        int64_t target, uint8_t *keep_out, void *ws, int64_t ws_bytes, void *stream) {
    if (m < 0 || target < 0 || n_cells < 0) return fail_arg("topk_keep_cells: negative size");
    const int64_t cells = align_up(4 * (n_cells > 0 ? n_cells : 1), 256);
    const int64_t need = select_ws_bytes() + cells;
    if (!ws) return need;
    if (ws_bytes < need) { set_error("topk_keep_cells: workspace %lld < %lld", (long long)ws_bytes, (long long)need); return FPCC_E_WORKSPACE; }
    if (m == 0) return FPCC_OK;
    if (!logit || !keep_out || !cell_of_group) return fail_arg("topk_keep_cells: null pointer");
    if ((reinterpret_cast<uintptr_t>(logit) & 15) || (reinterpret_cast<uintptr_t>(keep_out) & 7) || (reinterpret_cast<uintptr_t>(ws) & 15))
        return fail_arg("topk_keep_cells: logit / workspace must be 16-byte and keep_out 8-byte aligned");
    unsigned *seg_max = reinterpret_cast<unsigned *>(static_cast<char *>(ws) + select_ws_bytes());
    hipStream_t s = as_stream(stream);
    FPCC_HIP(hipMemsetAsync(seg_max, 0, (size_t)(4 * n_cells), s));       // 0 orders below every float
    hipLaunchKernelGGL(k_seg_max, dim3(blocks_for(m, kThreads)), dim3(kThreads), 0, s, reinterpret_cast<const float4 *>(logit), m, cell_of_group,
                       seg_max);
    FPCC_LAUNCHED(k_seg_max);
    return select_and_keep<true>(logit, m, cell_of_group, (const unsigned *)seg_max, 8 * m - target, keep_out, ws, s);
}
