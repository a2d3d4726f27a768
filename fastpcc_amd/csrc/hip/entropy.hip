// Entropy-model glue kernels: everything between the feature tensors and the (host) rANS coders that the reference
// does with small PyTorch ops and device->host copies.  All are streaming, HBM-bound passes.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

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

// one thread per parent cell: candidates that equal the cell maximum are "local maxima" and leave the ranking
__global__ void k_mask_local_max(const float4 *__restrict__ logit, int64_t m, float4 *__restrict__ ranked) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= m) return;
    const float4 a = logit[2 * p], b = logit[2 * p + 1];
    const float mx = max8(a, b);
    const float inf = __builtin_huge_valf();
    ranked[2 * p] = make_float4(a.x == mx ? inf : a.x, a.y == mx ? inf : a.y, a.z == mx ? inf : a.z, a.w == mx ? inf : a.w);
    ranked[2 * p + 1] = make_float4(b.x == mx ? inf : b.x, b.y == mx ? inf : b.y, b.z == mx ? inf : b.z, b.w == mx ? inf : b.w);
}

__global__ void k_keep(const float4 *__restrict__ logit, int64_t m, const float *__restrict__ sorted, int64_t kth,
                       uint8_t *__restrict__ keep) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= m) return;
    const float thr = kth >= 1 ? sorted[kth - 1] : -__builtin_huge_valf();
    const float4 a = logit[2 * p], b = logit[2 * p + 1];
    const float mx = max8(a, b);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint64_t bits = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) bits |= (uint64_t)((v[k] > thr || v[k] == mx) ? 1 : 0) << (8 * k);
    reinterpret_cast<uint64_t *>(keep)[p] = bits;
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

__global__ void k_mask_seg_max(const float4 *__restrict__ logit, int64_t m, const int32_t *__restrict__ seg,
                               const unsigned *__restrict__ seg_max, float4 *__restrict__ ranked) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= m) return;
    const float4 a = logit[2 * p], b = logit[2 * p + 1];
    const float mx = ord2f(seg_max[seg[p]]);
    const float inf = __builtin_huge_valf();
    ranked[2 * p] = make_float4(a.x == mx ? inf : a.x, a.y == mx ? inf : a.y, a.z == mx ? inf : a.z, a.w == mx ? inf : a.w);
    ranked[2 * p + 1] = make_float4(b.x == mx ? inf : b.x, b.y == mx ? inf : b.y, b.z == mx ? inf : b.z, b.w == mx ? inf : b.w);
}

__global__ void k_keep_seg(const float4 *__restrict__ logit, int64_t m, const int32_t *__restrict__ seg,
                           const unsigned *__restrict__ seg_max, const float *__restrict__ sorted, int64_t kth,
                           uint8_t *__restrict__ keep) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= m) return;
    const float thr = kth >= 1 ? sorted[kth - 1] : -__builtin_huge_valf();
    const float4 a = logit[2 * p], b = logit[2 * p + 1];
    const float mx = ord2f(seg_max[seg[p]]);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint64_t bits = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) bits |= (uint64_t)((v[k] > thr || v[k] == mx) ? 1 : 0) << (8 * k);
    reinterpret_cast<uint64_t *>(keep)[p] = bits;
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

extern "C" int64_t fpcc_topk_keep(const float *logit, int64_t m, int64_t target, uint8_t *keep_out, void *ws,
                                  int64_t ws_bytes, void *stream) {
    if (m < 0 || target < 0) return fail_arg("topk_keep: negative size");
    const int64_t n = 8 * (m > 0 ? m : 1);
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, sort_bytes, (const float *)nullptr, (float *)nullptr, (size_t)n);
    if (e != hipSuccess) return check_hip(e, "radix_sort_keys(size query)");
    const int64_t arr = align_up(4 * n, 256);
    const int64_t need = 2 * arr + align_up((int64_t)sort_bytes, 256);
    if (!ws) return need;
    if (ws_bytes < need) { set_error("topk_keep: workspace %lld < %lld", (long long)ws_bytes, (long long)need); return FPCC_E_WORKSPACE; }
    if (m == 0) return FPCC_OK;
    if (!logit || !keep_out) return fail_arg("topk_keep: null pointer");
    if ((reinterpret_cast<uintptr_t>(logit) & 15) || (reinterpret_cast<uintptr_t>(keep_out) & 7))
        return fail_arg("topk_keep: logit must be 16-byte and keep_out 8-byte aligned");
    float *ranked = static_cast<float *>(ws);
    float *sorted = reinterpret_cast<float *>(static_cast<char *>(ws) + arr);
    void *tmp = static_cast<char *>(ws) + 2 * arr;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_mask_local_max, dim3(blocks_for(m, kThreads)), dim3(kThreads), 0, s,
                       reinterpret_cast<const float4 *>(logit), m, reinterpret_cast<float4 *>(ranked));
    FPCC_LAUNCHED(k_mask_local_max);
    FPCC_HIP(rocprim::radix_sort_keys(tmp, sort_bytes, (const float *)ranked, sorted, (size_t)(8 * m), 0u, 32u, s));
    const int64_t kth = 8 * m - target;   // k-th smallest (1-based) of the non-maximum candidates is the threshold
    hipLaunchKernelGGL(k_keep, dim3(blocks_for(m, kThreads)), dim3(kThreads), 0, s,
                       reinterpret_cast<const float4 *>(logit), m, (const float *)sorted, kth, keep_out);
    FPCC_LAUNCHED(k_keep);
    return FPCC_OK;
}

extern "C" int64_t fpcc_topk_keep_cells(const float *logit, int64_t m, const int32_t *cell_of_group, int64_t n_cells,
                                        int64_t target, uint8_t *keep_out, void *ws, int64_t ws_bytes, void *stream) {
    if (m < 0 || target < 0 || n_cells < 0) return fail_arg("topk_keep_cells: negative size");
    const int64_t n = 8 * (m > 0 ? m : 1);
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, sort_bytes, (const float *)nullptr, (float *)nullptr, (size_t)n);
    if (e != hipSuccess) return check_hip(e, "radix_sort_keys(size query)");
    const int64_t arr = align_up(4 * n, 256);
    const int64_t cells = align_up(4 * (n_cells > 0 ? n_cells : 1), 256);
    const int64_t need = 2 * arr + cells + align_up((int64_t)sort_bytes, 256);
    if (!ws) return need;
    if (ws_bytes < need) { set_error("topk_keep_cells: workspace %lld < %lld", (long long)ws_bytes, (long long)need); return FPCC_E_WORKSPACE; }
    if (m == 0) return FPCC_OK;
    if (!logit || !keep_out || !cell_of_group) return fail_arg("topk_keep_cells: null pointer");
    if ((reinterpret_cast<uintptr_t>(logit) & 15) || (reinterpret_cast<uintptr_t>(keep_out) & 7))
        return fail_arg("topk_keep_cells: logit must be 16-byte and keep_out 8-byte aligned");
    float *ranked = static_cast<float *>(ws);
    float *sorted = reinterpret_cast<float *>(static_cast<char *>(ws) + arr);
    unsigned *seg_max = reinterpret_cast<unsigned *>(static_cast<char *>(ws) + 2 * arr);
    void *tmp = static_cast<char *>(ws) + 2 * arr + cells;
    hipStream_t s = as_stream(stream);
    FPCC_HIP(hipMemsetAsync(seg_max, 0, (size_t)(4 * n_cells), s));       // 0 orders below every float
    const dim3 grid(blocks_for(m, kThreads)), block(kThreads);
    const float4 *lg = reinterpret_cast<const float4 *>(logit);
    hipLaunchKernelGGL(k_seg_max, grid, block, 0, s, lg, m, cell_of_group, seg_max);
    FPCC_LAUNCHED(k_seg_max);
    hipLaunchKernelGGL(k_mask_seg_max, grid, block, 0, s, lg, m, cell_of_group, (const unsigned *)seg_max,
                       reinterpret_cast<float4 *>(ranked));
    FPCC_LAUNCHED(k_mask_seg_max);
    FPCC_HIP(rocprim::radix_sort_keys(tmp, sort_bytes, (const float *)ranked, sorted, (size_t)(8 * m), 0u, 32u, s));
    hipLaunchKernelGGL(k_keep_seg, grid, block, 0, s, lg, m, cell_of_group, (const unsigned *)seg_max, (const float *)sorted,
                       8 * m - target, keep_out);
    FPCC_LAUNCHED(k_keep_seg);
    return FPCC_OK;
}
