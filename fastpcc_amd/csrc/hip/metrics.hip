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

// ---------------------------------------------------------------------------------------------------------------------------------
// Point-to-plane (D2) and Hausdorff distortion: what the reference obtains from `pc_error -n <normals> --hausdorff=1`
// (/root/reference/lib/metrics/pc_error_wrapper.py:40-76; normals from the PLY file or Open3D's estimate_normals, :57-72).
// Everything works on SORTED Morton key sets like k_nn_dist2; rows are indexes into those sorted sets.

struct Blocks27 {                    // the 27 blocks of edge 2^l around a query: contiguous key ranges
    int64_t prefix;
    int32_t side;
    int4 q;
    __device__ Blocks27(int4 q_, int bits) : prefix((int64_t)q_.x << (3 * bits)), side(1 << bits), q(q_) {}
    template <class F>
    __device__ __forceinline__ void scan(const int64_t *__restrict__ keys, int64_t m, int bits, int l, F &&f) const {
        const int64_t morton_mask = ((int64_t)1 << (3 * bits)) - 1;
        const int32_t nblk = side >> l;
        const int32_t bx = min(max(q.y, 0), side - 1) >> l, by = min(max(q.z, 0), side - 1) >> l, bz = min(max(q.w, 0), side - 1) >> l;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int32_t cx = bx + dx, cy = by + dy, cz = bz + dz;
                    if (cx < 0 || cy < 0 || cz < 0 || cx >= nblk || cy >= nblk || cz >= nblk) continue;
                    const int64_t first = prefix | (int64_t)((m_spread21(cx) | m_spread21(cy) << 1 | m_spread21(cz) << 2) << (3 * l));
                    const int64_t last = first + ((int64_t)1 << (3 * l));
                    for (int64_t r = lower_bound(keys, m, first); r < m; ++r) {
                        int64_t k = keys[r];
                        if (k >= last) break;
                        k &= morton_mask;
                        const int64_t ex = (int64_t)m_gather21((uint64_t)k) - q.y, ey = (int64_t)m_gather21((uint64_t)k >> 1) - q.z,
                                      ez = (int64_t)m_gather21((uint64_t)k >> 2) - q.w;
                        f(r, ex, ey, ez, ex * ex + ey * ey + ez * ez);
                    }
                }
    }
};

// The K nearest voxels of every query, ordered by (squared distance, row): the order is total, so the result is unique.  A level is
// final when K candidates are STRICTLY nearer than 2^l + 1 -- every voxel outside the 27 blocks is at least that far along one axis.
template <int K>
__global__ __launch_bounds__(128) void k_knn_voxels(const int64_t *__restrict__ keys, int64_t m, int bits, const int32_t *__restrict__ query,
                                                    int64_t n, int k, int l0, int32_t *__restrict__ idx, int64_t *__restrict__ dist2) {
    const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    const Blocks27 blocks(reinterpret_cast<const int4 *>(query)[i], bits);
    int64_t best[K];
    int32_t who[K];
    for (int l = min(l0, bits); l <= bits; ++l) {
#pragma unroll
        for (int j = 0; j < K; ++j) { best[j] = INT64_MAX; who[j] = -1; }
        blocks.scan(keys, m, bits, l, [&](int64_t r, int64_t, int64_t, int64_t, int64_t d) {
            int32_t w = (int32_t)r;
            if (d < best[K - 1] || (d == best[K - 1] && w < who[K - 1])) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    if (d < best[j] || (d == best[j] && (uint32_t)w < (uint32_t)who[j])) {      // an empty slot has row -1 = the largest
                        const int64_t td = best[j]; best[j] = d; d = td;
                        const int32_t tw = who[j]; who[j] = w; w = tw;
                    }
                }
            }
        });
        int64_t kth = INT64_MAX;
#pragma unroll
        for (int j = 0; j < K; ++j) if (j == k - 1) kth = best[j];
        const int64_t reach = ((int64_t)1 << l) + 1;
        if (kth < reach * reach) break;
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < k) { idx[i * k + j] = who[j]; dist2[i * k + j] = who[j] < 0 ? -1 : best[j]; }
}

// Smallest-eigenvalue eigenvector of a symmetric 3x3 matrix without iteration (D. Eberly, "A Robust Eigensolver for 3x3 Symmetric
// Matrices", the method behind Open3D's fast normal computation): trigonometric eigenvalues of the scaled matrix, the eigenvector of
// the best separated eigenvalue from the largest cross product of two rows, the others from the 2x2 problem in its complement.
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 v_cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double v_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 v_scale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 v_sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }

__device__ V3 eigvec_first(double a00, double a01, double a02, double a11, double a12, double a22, double ev) {
    const V3 r0{a00 - ev, a01, a02}, r1{a01, a11 - ev, a12}, r2{a02, a12, a22 - ev};
    const V3 c01 = v_cross(r0, r1), c02 = v_cross(r0, r2), c12 = v_cross(r1, r2);
    const double d0 = v_dot(c01, c01), d1 = v_dot(c02, c02), d2 = v_dot(c12, c12);
    double dmax = d0;
    V3 best = c01;
    if (d1 > dmax) { dmax = d1; best = c02; }
    if (d2 > dmax) { dmax = d2; best = c12; }
    return dmax > 0.0 ? v_scale(best, 1.0 / sqrt(dmax)) : V3{0.0, 0.0, 0.0};
}

__device__ V3 eigvec_second(double a00, double a01, double a02, double a11, double a12, double a22, V3 e0, double ev) {
    V3 u;                                                   // orthonormal complement (u, v) of e0
    if (fabs(e0.x) > fabs(e0.y)) {
        const double inv = 1.0 / sqrt(e0.x * e0.x + e0.z * e0.z);
        u = {-e0.z * inv, 0.0, e0.x * inv};
    } else {
        const double inv = 1.0 / sqrt(e0.y * e0.y + e0.z * e0.z);
        u = {0.0, e0.z * inv, -e0.y * inv};
    }
    const V3 v = v_cross(e0, u);
    const V3 au{a00 * u.x + a01 * u.y + a02 * u.z, a01 * u.x + a11 * u.y + a12 * u.z, a02 * u.x + a12 * u.y + a22 * u.z};
    const V3 av{a00 * v.x + a01 * v.y + a02 * v.z, a01 * v.x + a11 * v.y + a12 * v.z, a02 * v.x + a12 * v.y + a22 * v.z};
    double m00 = v_dot(u, au) - ev, m01 = v_dot(u, av), m11 = v_dot(v, av) - ev;
    const double abs00 = fabs(m00), abs01 = fabs(m01), abs11 = fabs(m11);
    if (abs00 >= abs11) {
        if (fmax(abs00, abs01) > 0.0) {
            if (abs00 >= abs01) { m01 /= m00; m00 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1.0 / sqrt(1.0 + m00 * m00); m00 *= m01; }
            return v_sub(v_scale(u, m01), v_scale(v, m00));
        }
        return u;
    }
    if (fmax(abs11, abs01) > 0.0) {
        if (abs11 >= abs01) { m01 /= m11; m11 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m11; }
        else { m11 /= m01; m01 = 1.0 / sqrt(1.0 + m11 * m11); m11 *= m01; }
        return v_sub(v_scale(u, m11), v_scale(v, m01));
    }
    return u;
}

// One normal per point from the covariance of its k neighbours (rows of the sorted key set, the point itself among them -- Open3D's
// estimate_normals with a KNN search, lib/metrics/pc_error_wrapper.py:66-68).  A PCA normal has no sign; it is given the one that
// makes n . (1, sqrt 2, sqrt 5) positive (no lattice direction is perpendicular to that), so that the averages pc_error takes over
// neighbouring normals mean something.  Fewer than 3 neighbours or a vanishing covariance: (0, 0, 1), as Open3D.
__global__ __launch_bounds__(128) void k_pca_normals(const int64_t *__restrict__ keys, int bits, const int32_t *__restrict__ nbr, int64_t n,
                                                     int k, double *__restrict__ normals) {
    const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    const int64_t morton_mask = ((int64_t)1 << (3 * bits)) - 1;
    double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    int cnt = 0;
    // integer sums are exact in double up to 2^53: coordinates are taken relative to the first neighbour
    int64_t ox = 0, oy = 0, oz = 0;
    for (int j = 0; j < k; ++j) {
        const int32_t r = nbr[i * k + j];
        if (r < 0) continue;
        const uint64_t key = (uint64_t)(keys[r] & morton_mask);
        const int64_t x = m_gather21(key), y = m_gather21(key >> 1), z = m_gather21(key >> 2);
        if (cnt == 0) { ox = x; oy = y; oz = z; }
        const double dx = (double)(x - ox), dy = (double)(y - oy), dz = (double)(z - oz);
        sx += dx; sy += dy; sz += dz;
        sxx += dx * dx; sxy += dx * dy; sxz += dx * dz; syy += dy * dy; syz += dy * dz; szz += dz * dz;
        ++cnt;
    }
    V3 nrm{0.0, 0.0, 1.0};
    if (cnt >= 3) {
        const double inv = 1.0 / cnt;
        const double mx = sx * inv, my = sy * inv, mz = sz * inv;
        double a00 = sxx * inv - mx * mx, a01 = sxy * inv - mx * my, a02 = sxz * inv - mx * mz, a11 = syy * inv - my * my,
               a12 = syz * inv - my * mz, a22 = szz * inv - mz * mz;
        const double amax = fmax(fmax(fmax(fabs(a00), fabs(a01)), fmax(fabs(a02), fabs(a11))), fmax(fabs(a12), fabs(a22)));
        if (amax > 0.0) {
            const double s = 1.0 / amax;
            a00 *= s; a01 *= s; a02 *= s; a11 *= s; a12 *= s; a22 *= s;
            const double off = a01 * a01 + a02 * a02 + a12 * a12;
            V3 e{0.0, 0.0, 0.0};
            if (off > 0.0) {
                const double q = (a00 + a11 + a22) / 3.0;
                const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
                const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * off) / 6.0);
                const double c00 = b11 * b22 - a12 * a12, c01 = a01 * b22 - a12 * a02, c02 = a01 * a12 - b11 * a02;
                const double det = (b00 * c00 - a01 * c01 + a02 * c02) / (p * p * p);
                const double half = fmin(fmax(0.5 * det, -1.0), 1.0);
                const double angle = acos(half) / 3.0;
                const double two_thirds_pi = 2.09439510239319549;
                const double beta2 = 2.0 * cos(angle), beta0 = 2.0 * cos(angle + two_thirds_pi), beta1 = -(beta0 + beta2);
                const double ev0 = q + p * beta0, ev1 = q + p * beta1, ev2 = q + p * beta2;
                if (half >= 0.0) {
                    const V3 e2 = eigvec_first(a00, a01, a02, a11, a12, a22, ev2);
                    const V3 e1 = eigvec_second(a00, a01, a02, a11, a12, a22, e2, ev1);
                    e = v_cross(e1, e2);
                } else {
                    e = eigvec_first(a00, a01, a02, a11, a12, a22, ev0);
                }
            } else {                                        // diagonal: the axis of the smallest entry (first on ties)
                e = (a00 <= a11 && a00 <= a22) ? V3{1.0, 0.0, 0.0} : (a11 <= a22 ? V3{0.0, 1.0, 0.0} : V3{0.0, 0.0, 1.0});
            }
            const double nn = v_dot(e, e);
            if (nn > 0.0) {
                nrm = v_scale(e, 1.0 / sqrt(nn));
                if (nrm.x + 1.4142135623730951 * nrm.y + 2.23606797749979 * nrm.z < 0.0) nrm = v_scale(nrm, -1.0);
            }
        }
    }
    normals[3 * i] = nrm.x; normals[3 * i + 1] = nrm.y; normals[3 * i + 2] = nrm.z;
}

// Nearest voxel(s) of every query with ALL ties at the minimum distance (pc_error collects the neighbours at the same distance as the
// nearest one): first the nearest distance as in k_nn_dist2 -- with a STRICT bound, so that no voxel outside the examined blocks can
// tie --, then one more pass over the final level's blocks.
//   MODE 0 (distortion): plane[i] = mean over the ties j of ((q - p_j) . normal_j)^2, dist2[i], nn_row[i] = the first tie by row
//   MODE 1 (normal transfer): out3[i] = mean over the ties j of normal_j
template <int MODE>
__global__ __launch_bounds__(128) void k_nn_ties(const int64_t *__restrict__ keys, int64_t m, int bits, const double *__restrict__ normals,
                                                 const int32_t *__restrict__ query, int64_t n, int64_t *__restrict__ dist2,
                                                 int32_t *__restrict__ nn_row, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    const Blocks27 blocks(reinterpret_cast<const int4 *>(query)[i], bits);
    int64_t best = -1;
    int l = 0;
    for (; l <= bits; ++l) {
        best = -1;
        blocks.scan(keys, m, bits, l, [&](int64_t, int64_t, int64_t, int64_t, int64_t d) { if (best < 0 || d < best) best = d; });
        const int64_t reach = ((int64_t)1 << l) + 1;
        if ((best >= 0 && best < reach * reach) || l == bits) break;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    int32_t first = -1, ties = 0;
    if (best >= 0)
        blocks.scan(keys, m, bits, l, [&](int64_t r, int64_t ex, int64_t ey, int64_t ez, int64_t d) {
            if (d != best) return;
            double nx = 0.0, ny = 0.0, nz = 0.0;                                // normals == NULL: only distance and row are wanted
            if (normals) { nx = normals[3 * r]; ny = normals[3 * r + 1]; nz = normals[3 * r + 2]; }
            if (nx != nx || ny != ny || nz != nz) return;                       // pc_error skips normals that are not numbers
            if (first < 0 || (int32_t)r < first) first = (int32_t)r;
            ++ties;
            if (MODE == 0) {
                const double proj = (double)(-ex) * nx + (double)(-ey) * ny + (double)(-ez) * nz;     // (query - voxel) . normal
                a0 += proj * proj;
            } else {
                a0 += nx; a1 += ny; a2 += nz;
            }
        });
    // the ties arrive in block order, not row order: sums of up to ~30 terms whose order is nevertheless a function of the geometry alone
    if (dist2) dist2[i] = best;
    if (nn_row) nn_row[i] = first;
    if (MODE == 0) {
        out[i] = ties ? a0 / ties : 0.0;
    } else {
        out[3 * i] = ties ? a0 / ties : 0.0; out[3 * i + 1] = ties ? a1 / ties : 0.0; out[3 * i + 2] = ties ? a2 / ties : 0.0;
    }
}

// pc_error's normals for the second cloud, step 1: every point of the first cloud adds its normal to its nearest voxel of the second
// (nn_row from k_nn_ties).  Fixed point (2^-40) in 64-bit integers: the sum does not depend on the order of the atomics.
constexpr double kNormalFix = 1099511627776.0;          // 2^40
__global__ __launch_bounds__(256) void k_scatter_normals(const double *__restrict__ normals, const int32_t *__restrict__ nn_row, int64_t n,
                                                         long long *__restrict__ acc, int32_t *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t r = nn_row[i];
    if (r < 0) return;
    const double x = normals[3 * i], y = normals[3 * i + 1], z = normals[3 * i + 2];
    if (x != x || y != y || z != z) return;
    atomicAdd(reinterpret_cast<unsigned long long *>(acc + 3 * (int64_t)r), (unsigned long long)__double2ll_rn(x * kNormalFix));
    atomicAdd(reinterpret_cast<unsigned long long *>(acc + 3 * (int64_t)r + 1), (unsigned long long)__double2ll_rn(y * kNormalFix));
    atomicAdd(reinterpret_cast<unsigned long long *>(acc + 3 * (int64_t)r + 2), (unsigned long long)__double2ll_rn(z * kNormalFix));
    atomicAdd(count + r, 1);
}
// step 2: a voxel that received normals takes their mean; one that received none keeps the mean normal of its own nearest
// neighbours in the first cloud (already in `normals`, from k_nn_ties<1>)
__global__ __launch_bounds__(256) void k_finish_normals(const long long *__restrict__ acc, const int32_t *__restrict__ count, int64_t m,
                                                        double *__restrict__ normals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int32_t c = count[i];
    if (c <= 0) return;
    const double s = 1.0 / (kNormalFix * c);
    normals[3 * i] = (double)acc[3 * i] * s; normals[3 * i + 1] = (double)acc[3 * i + 1] * s; normals[3 * i + 2] = (double)acc[3 * i + 2] * s;
}

// Sum and maximum of doubles in a FIXED order: a workgroup reduces a fixed slice of 4096 entries (lane-strided partial sums, then a
// fixed tree), one workgroup then adds the slice results in ascending order.  The result is a function of the data alone.
__global__ __launch_bounds__(256) void k_sum_max_f64(const double *__restrict__ v, int64_t n, double *__restrict__ part) {
    __shared__ double s_sum[256], s_max[256];
    const int64_t base = (int64_t)blockIdx.x * 4096;
    double acc = 0.0, mx = -__builtin_inf();
    for (int j = 0; j < 16; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < n) { acc += v[i]; mx = fmax(mx, v[i]); }
    }
    s_sum[threadIdx.x] = acc; s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { s_sum[threadIdx.x] += s_sum[threadIdx.x + w]; s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + w]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s_sum[0]; part[2 * blockIdx.x + 1] = s_max[0]; }
}
__global__ void k_sum_max_f64_final(const double *__restrict__ part, int64_t blocks, double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double acc = 0.0, mx = -__builtin_inf();
    for (int64_t b = 0; b < blocks; ++b) { acc += part[2 * b]; mx = fmax(mx, part[2 * b + 1]); }
    out[0] = acc; out[1] = mx;
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

extern "C" int fpcc_knn_voxels(const int64_t *keys, int64_t m, int bits, const int32_t *query, int64_t n, int k, int start_level,
                               int32_t *idx_out, int64_t *dist2_out, void *stream) {
    if (m < 0 || n < 0 || bits < 1 || bits > 21 || k < 1 || k > 32 || start_level < 0) return fail_arg("knn_voxels: sizes out of range (bits 1..21, K 1..32)");
    if (n == 0) return FPCC_OK;
    if (!query || !idx_out || !dist2_out || (m > 0 && !keys)) return fail_arg("knn_voxels: null pointer");
    if (reinterpret_cast<uintptr_t>(query) & 15) return fail_arg("knn_voxels: query rows must be 16-byte aligned int32[4]");
    const dim3 grid(blocks_for(n, 128)), block(128);
    hipStream_t s = as_stream(stream);
    if (k <= 8) hipLaunchKernelGGL(k_knn_voxels<8>, grid, block, 0, s, keys, m, bits, query, n, k, start_level, idx_out, dist2_out);
    else if (k <= 16) hipLaunchKernelGGL(k_knn_voxels<16>, grid, block, 0, s, keys, m, bits, query, n, k, start_level, idx_out, dist2_out);
    else hipLaunchKernelGGL(k_knn_voxels<32>, grid, block, 0, s, keys, m, bits, query, n, k, start_level, idx_out, dist2_out);
    return check_hip(hipGetLastError(), "k_knn_voxels");
}

extern "C" int fpcc_pca_normals(const int64_t *keys, int64_t m, int bits, const int32_t *nbr, int64_t n, int k, double *normals_out,
                                void *stream) {
    if (m < 0 || n < 0 || bits < 1 || bits > 21 || k < 1) return fail_arg("pca_normals: sizes out of range");
    if (n == 0) return FPCC_OK;
    if (!keys || !nbr || !normals_out) return fail_arg("pca_normals: null pointer");
    hipLaunchKernelGGL(k_pca_normals, dim3(blocks_for(n, 128)), dim3(128), 0, as_stream(stream), keys, bits, nbr, n, k, normals_out);
    return check_hip(hipGetLastError(), "k_pca_normals");
}

extern "C" int fpcc_nn_plane_dist2(const int64_t *keys, int64_t m, int bits, const double *normals, const int32_t *query, int64_t n,
                                   int64_t *dist2_out, int32_t *nn_row_out, double *plane_out, void *stream) {
    if (m < 0 || n < 0 || bits < 1 || bits > 21) return fail_arg("nn_plane_dist2: sizes out of range (bits 1..21)");
    if (n == 0) return FPCC_OK;
    if (!query || !plane_out || (m > 0 && (!keys || !normals))) return fail_arg("nn_plane_dist2: null pointer");
    if (reinterpret_cast<uintptr_t>(query) & 15) return fail_arg("nn_plane_dist2: query rows must be 16-byte aligned int32[4]");
    hipLaunchKernelGGL(k_nn_ties<0>, dim3(blocks_for(n, 128)), dim3(128), 0, as_stream(stream), keys, m, bits, normals, query, n, dist2_out,
                       nn_row_out, plane_out);
    return check_hip(hipGetLastError(), "k_nn_ties<0>");
}

extern "C" int fpcc_transfer_normals(const int64_t *keys_a, int64_t n_a, const int32_t *coords_a, const double *normals_a,
                                     const int64_t *keys_b, int64_t n_b, const int32_t *coords_b, int bits, double *normals_b_out,
                                     void *ws, int64_t ws_bytes, void *stream) {
    if (n_a < 0 || n_b < 0 || bits < 1 || bits > 21) return fail_arg("transfer_normals: sizes out of range (bits 1..21)");
    if (n_b == 0) return FPCC_OK;
    if (!keys_b || !coords_b || !normals_b_out || (n_a > 0 && (!keys_a || !coords_a || !normals_a))) return fail_arg("transfer_normals: null pointer");
    if ((reinterpret_cast<uintptr_t>(coords_a) | reinterpret_cast<uintptr_t>(coords_b)) & 15) return fail_arg("transfer_normals: coordinate rows must be 16-byte aligned int32[4]");
    const int64_t need = fpcc_transfer_normals_ws_bytes(n_a, n_b);
    if (ws_bytes < need || !ws || (reinterpret_cast<uintptr_t>(ws) & 15)) return fail_arg("transfer_normals: workspace too small or misaligned");
    hipStream_t s = as_stream(stream);
    char *w = static_cast<char *>(ws);
    const int64_t o_count = align_up(24 * n_b, 16), o_row = o_count + align_up(4 * n_b, 16), o_plane = o_row + align_up(4 * n_a, 16);
    long long *acc = reinterpret_cast<long long *>(w);
    int32_t *count = reinterpret_cast<int32_t *>(w + o_count);
    int32_t *row = reinterpret_cast<int32_t *>(w + o_row);
    double *plane = reinterpret_cast<double *>(w + o_plane);
    FPCC_HIP(hipMemsetAsync(w, 0, o_row, s));
    // every voxel of B: the mean normal of its nearest voxels of A (kept where nothing is scattered onto the voxel)
    hipLaunchKernelGGL(k_nn_ties<1>, dim3(blocks_for(n_b, 128)), dim3(128), 0, s, keys_a, n_a, bits, normals_a, coords_b, n_b,
                       (int64_t *)nullptr, (int32_t *)nullptr, normals_b_out);
    if (n_a > 0) {
        // every voxel of A: its nearest voxel of B (the first by row among ties), which receives A's normal
        hipLaunchKernelGGL(k_nn_ties<0>, dim3(blocks_for(n_a, 128)), dim3(128), 0, s, keys_b, n_b, bits, (const double *)nullptr, coords_a, n_a,
                           (int64_t *)nullptr, row, plane);
        hipLaunchKernelGGL(k_scatter_normals, dim3(blocks_for(n_a, 256)), dim3(256), 0, s, normals_a, row, n_a, acc, count);
        hipLaunchKernelGGL(k_finish_normals, dim3(blocks_for(n_b, 256)), dim3(256), 0, s, acc, count, n_b, normals_b_out);
    }
    return check_hip(hipGetLastError(), "transfer_normals");
}

extern "C" int64_t fpcc_transfer_normals_ws_bytes(int64_t n_a, int64_t n_b) {
    if (n_a < 0 || n_b < 0) return FPCC_E_ARG;
    return align_up(24 * n_b, 16) + align_up(4 * n_b, 16) + align_up(4 * n_a, 16) + align_up(8 * n_a, 16) + 16;
}

extern "C" int fpcc_sum_max_f64(const double *values, int64_t n, double *sum_max_out, void *ws, int64_t ws_bytes, void *stream) {
    if (n < 0 || !sum_max_out || (n > 0 && !values)) return fail_arg("sum_max_f64: null pointer");
    const int64_t blocks = (n + 4095) / 4096;
    if (blocks > 0 && (!ws || ws_bytes < 16 * blocks)) return fail_arg("sum_max_f64: workspace of 16 bytes per 4096 values needed");
    hipStream_t s = as_stream(stream);
    if (blocks > 0) hipLaunchKernelGGL(k_sum_max_f64, dim3((unsigned)blocks), dim3(256), 0, s, values, n, static_cast<double *>(ws));
    hipLaunchKernelGGL(k_sum_max_f64_final, dim3(1), dim3(64), 0, s, static_cast<const double *>(ws), blocks, sum_max_out);
    return check_hip(hipGetLastError(), "k_sum_max_f64");
}
