// libfpcc_hip.so -- geometry distortion (point-to-point, "D1") on the device.
//
// Replaces the reference's evaluation step, which writes the reconstruction to a PLY file and runs the external MPEG
// `pc_error` binary on it (/root/reference/lib/evaluators.py:94-112, lib/metrics/pc_error_wrapper.py:40-107; the
// brute-force KNN kernel of lib/knn3d/src/knn3d.cu:74-130 serves the same purpose for training losses).
// Exact nearest neighbour in a voxel set held as SORTED Morton keys: around the query, the 27 blocks of edge 2^l that
// touch its own block are contiguous key ranges; every point outside them is farther than 2^l + 1 along some axis, so
// the search ends at the first level whose best squared distance is <= (2^l + 1)^2.  Integer arithmetic throughout.
#include "common.h"

#include <algorithm>

namespace fpcc {
namespace {

__device__ __forceinline__ uint64_t m_spread21(uint32_t v) {
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__device__ __forceinline__ uint32_t m_gather21(uint64_t x) {
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
    x = (x ^ (x >> 32)) & 0x1fffffull;
    return static_cast<uint32_t>(x);
}

__device__ __forceinline__ int64_t lower_bound(const int64_t *__restrict__ keys, int64_t n, int64_t want) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_nn_dist2(const int64_t *__restrict__ keys, int64_t m, int bits,
                                                  const int32_t *__restrict__ query, int64_t n,
                                                  int64_t *__restrict__ dist2, int32_t *__restrict__ nn_row) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 q = reinterpret_cast<const int4 *>(query)[i];      // (batch, x, y, z)
    const int64_t prefix = (int64_t)q.x << (3 * bits);
    const int64_t morton_mask = ((int64_t)1 << (3 * bits)) - 1;
    const int32_t side = 1 << bits;
    int64_t best = -1;
    int32_t best_row = -1;
    // coordinates outside the cube are legal queries: clamp the block walk, not the distance
    for (int l = 0; l <= bits; ++l) {
        const int32_t nblk = side >> l;                              // blocks per axis at this level
        const int32_t bx = min(max(q.y, 0), side - 1) >> l, by = min(max(q.z, 0), side - 1) >> l,
                      bz = min(max(q.w, 0), side - 1) >> l;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int32_t cx = bx + dx, cy = by + dy, cz = bz + dz;
                    if (cx < 0 || cy < 0 || cz < 0 || cx >= nblk || cy >= nblk || cz >= nblk) continue;
                    const int64_t first = prefix | (int64_t)((m_spread21(cx) | m_spread21(cy) << 1 | m_spread21(cz) << 2) << (3 * l));
                    const int64_t r0 = lower_bound(keys, m, first);
                    const int64_t last = first + ((int64_t)1 << (3 * l));
                    for (int64_t r = r0; r < m; ++r) {
                        int64_t k = keys[r];
                        if (k >= last) break;
                        k &= morton_mask;
                        const int64_t ex = (int64_t)m_gather21((uint64_t)k) - q.y, ey = (int64_t)m_gather21((uint64_t)k >> 1) - q.z,
                                      ez = (int64_t)m_gather21((uint64_t)k >> 2) - q.w;
                        const int64_t d = ex * ex + ey * ey + ez * ez;
                        if (best < 0 || d < best) { best = d; best_row = (int32_t)r; }
                    }
                }
        // a query outside the cube is farther from everything by its overshoot; the bound below stays valid because
        // the overshoot only adds to distances of points outside the examined blocks as well
        const int64_t reach = ((int64_t)1 << l) + 1;
        if (best >= 0 && best <= reach * reach) break;
    }
    dist2[i] = best;
    if (nn_row) nn_row[i] = best_row;
}

__global__ __launch_bounds__(256) void k_sum_i64(const int64_t *__restrict__ v, int64_t n, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long part[4];
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        acc += v[i] > 0 ? (unsigned long long)v[i] : 0ull;
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);   // integer: order independent
}


// K nearest neighbours of float points, brute force: a workgroup of 256 queries walks the candidate set in tiles of 1024
// points staged through LDS (12 KB; every lane reads the same LDS address per step: a broadcast, no bank conflicts) and
// keeps its K best (distance, index) pairs in registers, sorted by insertion.  O(P1 * P2) like the reference's kernel
// (lib/knn3d/src/knn3d.cu:74-130); the codec itself never needs it (nearest voxels come from k_nn_dist2 on sorted keys).
template <int K>
__global__ __launch_bounds__(256) void k_knn3d(const float *__restrict__ p1, int64_t n1, const float *__restrict__ p2,
                                               int64_t n2, int64_t *__restrict__ idx, float *__restrict__ dist2) {
    constexpr int kTile = 1024;
    __shared__ float s_pts[kTile * 3];
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = q < n1;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (live) { qx = p1[3 * q]; qy = p1[3 * q + 1]; qz = p1[3 * q + 2]; }
    float best[K];
    int64_t who[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { best[k] = __builtin_inff(); who[k] = -1; }
    for (int64_t base = 0; base < n2; base += kTile) {
        const int count = (int)(n2 - base < kTile ? n2 - base : kTile);
        for (int i = threadIdx.x; i < count * 3; i += 256) s_pts[i] = p2[3 * base + i];
        __syncthreads();
        if (live) {
            for (int j = 0; j < count; ++j) {
                const float dx = qx - s_pts[3 * j], dy = qy - s_pts[3 * j + 1], dz = qz - s_pts[3 * j + 2];
                float d = dx * dx + dy * dy + dz * dz;
                if (d < best[K - 1]) {                       // insert, keeping ascending order (ties: the earlier index stays first)
                    int64_t w = base + j;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if (d < best[k]) {
                            const float td = best[k]; best[k] = d; d = td;
                            const int64_t tw = who[k]; who[k] = w; w = tw;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < K; ++k) { idx[q * K + k] = who[k]; dist2[q * K + k] = best[k]; }
    }
}
}  // namespace
}  // namespace fpcc

using namespace fpcc;

extern "C" int fpcc_nn_dist2(const int64_t *keys, int64_t m, int bits, const int32_t *query, int64_t n, int64_t *dist2_out,
                             int32_t *nn_row_out, void *stream) {
    if (m < 0 || n < 0 || bits < 1 || bits > 21) return fail_arg("nn_dist2: sizes out of range (bits 1..21)");
    if (n == 0) return FPCC_OK;
    if (!query || !dist2_out || (m > 0 && !keys)) return fail_arg("nn_dist2: null pointer");
    if (reinterpret_cast<uintptr_t>(query) & 15) return fail_arg("nn_dist2: query rows must be 16-byte aligned int32[4]");
    hipLaunchKernelGGL(k_nn_dist2, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), keys, m, bits, query, n,
                       dist2_out, nn_row_out);
    return check_hip(hipGetLastError(), "k_nn_dist2");
}

extern "C" int fpcc_sum_i64(const int64_t *values, int64_t n, uint64_t *sum_out, void *stream) {
    if (n < 0 || !sum_out || (n > 0 && !values)) return fail_arg("sum_i64: null pointer");
    hipStream_t s = as_stream(stream);
    if (int rc = check_hip(hipMemsetAsync(sum_out, 0, sizeof(uint64_t), s), "hipMemsetAsync")) return rc;
    if (n == 0) return FPCC_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>(blocks_for(n, 256), 1024);
    hipLaunchKernelGGL(k_sum_i64, dim3(blocks), dim3(256), 0, s, values, n, reinterpret_cast<unsigned long long *>(sum_out));
    return check_hip(hipGetLastError(), "k_sum_i64");
}

extern "C" int fpcc_knn3d(const float *p1, int64_t n1, const float *p2, int64_t n2, int k, int64_t *idx_out, float *dist2_out,
                          void *stream) {
    if (n1 < 0 || n2 < 0 || k < 1 || k > 16) return fail_arg("knn3d: sizes out of range (K 1..16)");
    if (n1 == 0) return FPCC_OK;
    if (!p1 || !idx_out || !dist2_out || (n2 > 0 && !p2)) return fail_arg("knn3d: null pointer");
    hipStream_t s = as_stream(stream);
    const dim3 grid(blocks_for(n1, 256)), block(256);
    switch (k) {
#define FPCC_KNN_CASE(KK) case KK: hipLaunchKernelGGL(k_knn3d<KK>, grid, block, 0, s, p1, n1, p2, n2, idx_out, dist2_out); break;
        FPCC_KNN_CASE(1) FPCC_KNN_CASE(2) FPCC_KNN_CASE(3) FPCC_KNN_CASE(4) FPCC_KNN_CASE(5) FPCC_KNN_CASE(6) FPCC_KNN_CASE(7) FPCC_KNN_CASE(8)
        FPCC_KNN_CASE(9) FPCC_KNN_CASE(10) FPCC_KNN_CASE(11) FPCC_KNN_CASE(12) FPCC_KNN_CASE(13) FPCC_KNN_CASE(14) FPCC_KNN_CASE(15) FPCC_KNN_CASE(16)
#undef FPCC_KNN_CASE
    }
    return check_hip(hipGetLastError(), "k_knn3d");
}
