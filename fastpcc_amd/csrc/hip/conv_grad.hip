// libfpcc_hip.so -- weight gradient of the sparse convolution (training path).
//
//   dW[g][k][ci][co] = sum over output rows o of  X[in(k,o)][ci] * dY[dst(o,g)][co]
//
// with in(k,o) / dst(o,g) exactly the row maps of fpcc_conv_f32 (nbr table / output map), so one entry point serves
// stride-1, strided, transposed and generative convolutions and per-point linear layers.  What MinkowskiEngine computes
// with one gather + GEMM + accumulate per kernel offset in its backward pass
// (called through autograd from lib/minkowski_sparse_conv_layers.py:85-91 during train.py:262-270).
// The input gradient needs no kernel of its own: it is fpcc_conv_f32 on the mirrored row maps with transposed weights
// (fastpcc_amd/autograd.py).
//
// MFMA kernel (C_in, C_out multiples of 32): a workgroup owns (row split s, offset k, 32 input channels) and all output
// columns, one wave per 32-column block; the reduction dimension of this GEMM is the ROW index, so both operands are
// read straight from global memory as 128-byte row segments (lane = channel / column), two rows per MFMA k-step.
// Row splits write partial sums; k_wgrad_reduce adds them in ascending split order (fixed association, reproducible).
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace fpcc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
    const float *x; int c_in; int ldx;
    const float *dy; int c_out; int ldy;
    const int32_t *nbr; int n_off; int64_t nbr_ks; int64_t nbr_os;
    const int32_t *out_map; int64_t om_os; int64_t om_gs; int groups;
    int64_t n; int64_t rows_per_split; int splits;
    float *partial;            // [splits][groups*n_off][c_in][c_out]
};

__device__ float g_wgrad_zero[128];

__device__ __forceinline__ void row_pair(const WgradArgs &a, int k, int g, int64_t r, int64_t end, int64_t &in_row,
                                         int64_t &out_row) {
    in_row = out_row = -1;
    if (r < end) {
        in_row = a.nbr ? (int64_t)a.nbr[(int64_t)k * a.nbr_ks + r * a.nbr_os] : r;
        out_row = a.out_map ? (int64_t)a.out_map[r * a.om_os + g * a.om_gs] : r * a.groups + g;
        if (in_row < 0 || out_row < 0) in_row = out_row = -1;
    }
}

template <int NBT>
__global__ __launch_bounds__(64 * NBT) void k_wgrad_mfma(WgradArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int s = blockIdx.x, kg = blockIdx.y, cb = blockIdx.z;
    const int g = kg / a.n_off, k = kg % a.n_off;
    const int64_t begin = (int64_t)s * a.rows_per_split;
    const int64_t end = min(begin + a.rows_per_split, a.n);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    for (int64_t base = begin; base < end; base += 32) {
        float av[16], bv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int64_t in_row, out_row;
            row_pair(a, k, g, base + 2 * j + lh, end, in_row, out_row);
            const float *px = in_row >= 0 ? a.x + in_row * a.ldx + 32 * cb + li : g_wgrad_zero + li;
            const float *pd = out_row >= 0 ? a.dy + out_row * a.ldy + 32 * wave + li : g_wgrad_zero + li;
            av[j] = *px;
            bv[j] = *pd;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
    }
    float *dst = a.partial + (((int64_t)s * a.groups * a.n_off + kg) * a.c_in + 32 * cb) * a.c_out + 32 * wave + li;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        dst[(int64_t)ci * a.c_out] = acc[reg];
    }
}

// C_in a multiple of 32*CB (CB = 4 or 2): one workgroup covers 32*CB input channels x all output columns.  Wave w owns
// column block w and CB accumulator blocks; lane i of the A operand carries channels CB*i .. CB*i+CB-1 (one 16- or 8-byte
// load per row), block q holding channel CB*i + q -- a permutation of the channel <-> M-index assignment that costs
// nothing at the store.  Every X row is read once per wave (L1-shared by the workgroup's waves) and every dY row once per
// workgroup: 1/(2*CB) of the L2 traffic of the per-block kernel above.
template <int NBT, int CB>
__global__ __launch_bounds__(64 * NBT) void k_wgrad_mfma_wide(WgradArgs a) {
    typedef float fvec __attribute__((ext_vector_type(CB)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int s = blockIdx.x, kg = blockIdx.y, cg = blockIdx.z;
    const int g = kg / a.n_off, k = kg % a.n_off;
    const int64_t begin = (int64_t)s * a.rows_per_split;
    const int64_t end = min(begin + a.rows_per_split, a.n);
    f32x16 acc[CB];
#pragma unroll
    for (int q = 0; q < CB; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

    for (int64_t base = begin; base < end; base += 32) {
        fvec av[16];
        float bv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int64_t in_row, out_row;
            row_pair(a, k, g, base + 2 * j + lh, end, in_row, out_row);
            const float *px = in_row >= 0 ? a.x + in_row * a.ldx + 32 * CB * cg + CB * li : g_wgrad_zero + CB * li;
            const float *pd = out_row >= 0 ? a.dy + out_row * a.ldy + 32 * wave + li : g_wgrad_zero + li;
            av[j] = *reinterpret_cast<const fvec *>(px);
            bv[j] = *pd;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int q = 0; q < CB; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][q], bv[j], acc[q], 0, 0, 0);
    }
    float *dst = a.partial + (((int64_t)s * a.groups * a.n_off + kg) * a.c_in + 32 * CB * cg) * a.c_out + 32 * wave + li;
#pragma unroll
    for (int q = 0; q < CB; ++q)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = (reg & 3) + 8 * (reg >> 2) + 4 * lh;            // M index of the accumulator row
            dst[(int64_t)(CB * m + q) * a.c_out] = acc[q][reg];
        }
}

// 3x3x3-style maps given in neighbour-pattern row order (fpcc_conv_row_keys): the rows that have kernel offset k are
// clustered, so the reduction over rows is walked in blocks of 32 positions and a block none of whose rows has the offset
// is skipped -- a 27-bit mask per block, computed by k_wgrad_blockmask.  On voxelised surfaces each row has ~13 of the 27
// offsets: in natural order nearly every block has every offset and half of the MFMAs multiply the zero row, in pattern
// order ~1.1x the useful work is executed.  Operands are fetched one half block (16 rows) ahead of the MFMAs that use
// them and the row indices (row_order -> nbr) one active block ahead, so the two dependent loads and the gather hide
// behind 32 MFMAs each.  Geometry as k_wgrad_mfma_wide: 32*CB input channels x all output columns per workgroup.
__global__ __launch_bounds__(256) void k_wgrad_blockmask(const int32_t *__restrict__ nbr, int n_off, int64_t nbr_ks, int64_t nbr_os,
                                                         const int32_t *__restrict__ row_order, int64_t n, int64_t n_blocks,
                                                         uint32_t *__restrict__ masks) {
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= n_blocks) return;
    const int64_t pos = b * 32 + lane;
    const int64_t row = (lane < 32 && pos < n) ? (int64_t)row_order[pos] : -1;
    uint32_t m = 0;
    for (int k = 0; k < n_off; ++k) {
        const bool has = row >= 0 && nbr[(int64_t)k * nbr_ks + row * nbr_os] >= 0;
        m |= (__ballot(has) != 0ull ? 1u : 0u) << k;
    }
    if (lane == 0) masks[b] = m;
}

template <int NBT, int CB>
__global__ __launch_bounds__(64 * NBT, 2) void k_wgrad_rows(WgradArgs a, const uint32_t *__restrict__ masks,
                                                            const int32_t *__restrict__ row_order) {
    typedef float fvec __attribute__((ext_vector_type(CB)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int s = blockIdx.x, k = blockIdx.y, cg = blockIdx.z;
    // row split s takes blocks s, s + splits, ...: in pattern order the rows that have an offset are clustered, a
    // contiguous range per split would give some workgroups all of an offset's blocks and others none
    const int64_t b_begin = s, b_end = (a.n + 31) / 32, b_step = a.splits;
    f32x16 acc[CB];
#pragma unroll
    for (int q = 0; q < CB; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

    auto next_active = [&](int64_t b) {
        while (b < b_end && !((masks[b] >> k) & 1u)) b += b_step;
        return b;
    };
    // lane l (and l + 32) holds the rows of position 32 b + l: (input row of offset k, output row), -1 when absent
    auto load_rows = [&](int64_t b, int32_t &in_row, int32_t &out_row) {
        const int64_t pos = b * 32 + li;
        out_row = pos < a.n ? row_order[pos] : -1;
        in_row = out_row >= 0 ? a.nbr[(int64_t)k * a.nbr_ks + (int64_t)out_row * a.nbr_os] : -1;
        if (in_row < 0) out_row = -1;
    };
    auto load_half = [&](int half, int32_t in_row, int32_t out_row, fvec (&av)[8], float (&bv)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int src = 16 * half + 2 * j + lh;
            const int32_t ir = __shfl(in_row, src), orow = __shfl(out_row, src);
            const float *px = ir >= 0 ? a.x + (int64_t)ir * a.ldx + 32 * CB * cg + CB * li : g_wgrad_zero + CB * li;
            const float *pd = orow >= 0 ? a.dy + (int64_t)orow * a.ldy + 32 * wave + li : g_wgrad_zero + li;
            av[j] = *reinterpret_cast<const fvec *>(px);
            bv[j] = *pd;
        }
    };
    auto mma_half = [&](const fvec (&av)[8], const float (&bv)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int q = 0; q < CB; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][q], bv[j], acc[q], 0, 0, 0);
    };

    int64_t b = next_active(b_begin);
    if (b < b_end) {
        int32_t in_row, out_row, in_next, out_next;
        fvec av0[8], av1[8];
        float bv0[8], bv1[8];
        load_rows(b, in_row, out_row);
        load_half(0, in_row, out_row, av0, bv0);
        for (;;) {
            const int64_t bn = next_active(b + b_step);
            const bool more = bn < b_end;
            load_rows(more ? bn : b, in_next, out_next);           // unconditional: the last round re-reads its own block
            load_half(1, in_row, out_row, av1, bv1);
            mma_half(av0, bv0);
            load_half(0, in_next, out_next, av0, bv0);
            mma_half(av1, bv1);
            if (!more) break;
            b = bn;
            in_row = in_next;
            out_row = out_next;
        }
    }
    float *dst = a.partial + (((int64_t)s * a.n_off + k) * a.c_in + 32 * CB * cg) * a.c_out + 32 * wave + li;
#pragma unroll
    for (int q = 0; q < CB; ++q)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            dst[(int64_t)(CB * m + q) * a.c_out] = acc[q][reg];
        }
}

// any channel counts: thread = a few (ci, co) pairs, rows walked serially (row maps are wave-uniform scalar loads)
__global__ __launch_bounds__(256) void k_wgrad_valu(WgradArgs a) {
    const int s = blockIdx.x, kg = blockIdx.y;
    const int g = kg / a.n_off, k = kg % a.n_off;
    const int64_t begin = (int64_t)s * a.rows_per_split;
    const int64_t end = min(begin + a.rows_per_split, a.n);
    const int pairs = a.c_in * a.c_out;
    float *dst = a.partial + ((int64_t)s * a.groups * a.n_off + kg) * pairs;
    for (int p0 = 0; p0 < pairs; p0 += 256 * 4) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int ci[4], co[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * 256 + (int)threadIdx.x;
            ci[u] = p < pairs ? p / a.c_out : 0;
            co[u] = p < pairs ? p % a.c_out : 0;
        }
        for (int64_t r = begin; r < end; ++r) {
            int64_t in_row, out_row;
            row_pair(a, k, g, r, end, in_row, out_row);
            if (in_row < 0) continue;
            const float *px = a.x + in_row * a.ldx;
            const float *pd = a.dy + out_row * a.ldy;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = fmaf(px[ci[u]], pd[co[u]], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * 256 + (int)threadIdx.x;
            if (p < pairs) dst[p] = acc[u];
        }
    }
}

// few (ci, co) pairs (<= 128: one-channel heads, the first layers, the decoder's classifier): every thread walks its own
// rows with ALL pairs in registers, then the block adds the 256 private sums (wave shuffles, then LDS)
template <int CI, int CO>
__global__ __launch_bounds__(256) void k_wgrad_small(WgradArgs a) {
    constexpr int P = CI * CO;
    const int s = blockIdx.x, kg = blockIdx.y;
    const int g = kg / a.n_off, k = kg % a.n_off;
    const int64_t begin = (int64_t)s * a.rows_per_split;
    const int64_t end = min(begin + a.rows_per_split, a.n);
    float acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = 0.0f;
    for (int64_t r = begin + threadIdx.x; r < end; r += 256) {
        int64_t in_row, out_row;
        row_pair(a, k, g, r, end, in_row, out_row);
        if (in_row < 0) continue;
        const float *px = a.x + in_row * a.ldx;
        const float *pd = a.dy + out_row * a.ldy;
        float d[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) d[co] = pd[co];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            const float xv = px[ci];
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[ci * CO + co] = fmaf(xv, d[co], acc[ci * CO + co]);
        }
    }
    __shared__ float part[4][P];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float v = acc[p];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) part[wave][p] = v;
    }
    __syncthreads();
    float *dst = a.partial + ((int64_t)s * a.groups * a.n_off + kg) * P;
    for (int p = threadIdx.x; p < P; p += 256) dst[p] = (part[0][p] + part[1][p]) + (part[2][p] + part[3][p]);
}

template <int CI, int CO>
bool try_small(const WgradArgs &a, int splits, int kg, hipStream_t s) {
    if (a.c_in != CI || a.c_out != CO) return false;
    hipLaunchKernelGGL((k_wgrad_small<CI, CO>), dim3(splits, kg), dim3(256), 0, s, a);
    return true;
}

bool launch_small(const WgradArgs &a, int splits, int kg, hipStream_t s) {
    return try_small<1, 1>(a, splits, kg, s) || try_small<1, 16>(a, splits, kg, s) || try_small<1, 64>(a, splits, kg, s) ||
           try_small<8, 1>(a, splits, kg, s) || try_small<16, 8>(a, splits, kg, s) || try_small<16, 1>(a, splits, kg, s) ||
           try_small<32, 1>(a, splits, kg, s) || try_small<64, 1>(a, splits, kg, s) || try_small<128, 1>(a, splits, kg, s) ||
           try_small<4, 16>(a, splits, kg, s) || try_small<1, 32>(a, splits, kg, s) || try_small<1, 8>(a, splits, kg, s);
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, int splits, int64_t count,
                                                      float *__restrict__ dw, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    float acc = accumulate ? dw[e] : 0.0f;
    for (int s = 0; s < splits; ++s) acc = acc + partial[(int64_t)s * count + e];
    dw[e] = acc;
}

// row splits: enough workgroups to fill the chip ~3x, at least 256 rows each, at most 512 splits
int pick_splits(int c_in, int c_out, int kg, int64_t n, bool mfma) {
    const int64_t per_split_items = mfma ? (int64_t)kg * (c_in % 128 == 0 ? c_in / 128 : c_in % 64 == 0 ? c_in / 64 : c_in / 32) : kg;
    int64_t want = (3 * 256 + per_split_items - 1) / per_split_items;
    want = std::min<int64_t>(want, (n + 255) / 256);
    want = std::max<int64_t>(1, std::min<int64_t>(want, 512));
    (void)c_out;
    return (int)want;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool wgrad_mfma_ok(int c_in, int c_out) { return c_in % 32 == 0 && (c_out == 32 || c_out == 64 || c_out == 128); }

}  // namespace
}  // namespace fpcc

using namespace fpcc;

extern "C" int64_t fpcc_conv_wgrad_ws_bytes(int c_in, int c_out, int n_offsets, int groups, int64_t n) {
    if (c_in < 1 || c_out < 1 || n_offsets < 1 || groups < 1 || n < 0) return FPCC_E_ARG;
    const int kg = n_offsets * groups;
    const int splits = pick_splits(c_in, c_out, kg, n, wgrad_mfma_ok(c_in, c_out));
    return (int64_t)splits * kg * c_in * c_out * 4 + ((n + 31) / 32 + 4) * 4;      // partial sums + one mask per 32 rows
}

extern "C" int fpcc_conv_wgrad_f32(const float *x, int c_in, int ldx, const float *dy, int c_out, int ldy,
                                   const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os,
                                   const int32_t *out_map, int64_t om_os, int64_t om_gs, int groups, int64_t n,
                                   const int32_t *row_order, float *dw, int accumulate, void *ws, int64_t ws_bytes,
                                   void *stream) {
    if (c_in < 1 || c_out < 1 || n_offsets < 1 || groups < 1 || n < 0) return fail_arg("conv_wgrad: sizes out of range");
    if (!dw) return fail_arg("conv_wgrad: null pointer");
    if (!nbr && n_offsets != 1) return fail_arg("conv_wgrad: identity map needs n_offsets == 1");
    if (ldx < c_in || ldy < c_out) return fail_arg("conv_wgrad: row stride smaller than the row");
    const int kg = n_offsets * groups;
    const int64_t count = (int64_t)kg * c_in * c_out;
    hipStream_t s = as_stream(stream);
    if (n == 0) {
        if (!accumulate) return check_hip(hipMemsetAsync(dw, 0, count * 4, s), "hipMemsetAsync");
        return FPCC_OK;
    }
    if (!x || !dy) return fail_arg("conv_wgrad: null pointer");
    const bool mfma = wgrad_mfma_ok(c_in, c_out);
    const int splits = pick_splits(c_in, c_out, kg, n, mfma);
    const int64_t need = (int64_t)splits * count * 4 + ((n + 31) / 32 + 4) * 4;
    if (!ws || ws_bytes < need) return fail_arg("conv_wgrad: workspace of fpcc_conv_wgrad_ws_bytes() bytes required");
    const int64_t rows_per_split = ((n + splits - 1) / splits + 31) / 32 * 32;
    WgradArgs a{x, c_in, ldx, dy, c_out, ldy, nbr, n_offsets, nbr_ks, nbr_os, out_map, om_os, om_gs, groups, n,
                rows_per_split, splits, static_cast<float *>(ws)};
    static const int skip_rows = [] { const char *e = getenv("FPCC_WGRAD_ROWS"); return e ? atoi(e) : 1; }();
    if (skip_rows && mfma && row_order && nbr && !out_map && groups == 1 && n_offsets <= 32 && c_in % 64 == 0 && aligned16(x) &&
        ldx % 4 == 0 && (c_out == 128 || c_out == 64)) {
        uint32_t *masks = reinterpret_cast<uint32_t *>(static_cast<float *>(ws) + (int64_t)splits * count);
        const int64_t n_blocks = (n + 31) / 32;
        hipLaunchKernelGGL(k_wgrad_blockmask, dim3(blocks_for(n_blocks, 4)), dim3(256), 0, s, nbr, n_offsets, nbr_ks, nbr_os,
                           row_order, n, n_blocks, masks);
        if (int rc = check_hip(hipGetLastError(), "k_wgrad_blockmask")) return rc;
        if (c_in % 128 == 0) {
            const dim3 grid(splits, kg, c_in / 128);
            if (c_out == 128) hipLaunchKernelGGL((k_wgrad_rows<4, 4>), grid, dim3(256), 0, s, a, masks, row_order);
            else hipLaunchKernelGGL((k_wgrad_rows<2, 4>), grid, dim3(128), 0, s, a, masks, row_order);
        } else {
            const dim3 grid(splits, kg, c_in / 64);
            if (c_out == 128) hipLaunchKernelGGL((k_wgrad_rows<4, 2>), grid, dim3(256), 0, s, a, masks, row_order);
            else hipLaunchKernelGGL((k_wgrad_rows<2, 2>), grid, dim3(128), 0, s, a, masks, row_order);
        }
    } else if (mfma && c_in % 128 == 0 && aligned16(x) && ldx % 4 == 0) {
        const dim3 grid(splits, kg, c_in / 128);
        if (c_out == 128) hipLaunchKernelGGL((k_wgrad_mfma_wide<4, 4>), grid, dim3(256), 0, s, a);
        else if (c_out == 64) hipLaunchKernelGGL((k_wgrad_mfma_wide<2, 4>), grid, dim3(128), 0, s, a);
        else hipLaunchKernelGGL((k_wgrad_mfma_wide<1, 4>), grid, dim3(64), 0, s, a);
    } else if (mfma && c_in % 64 == 0 && aligned16(x) && ldx % 4 == 0) {
        const dim3 grid(splits, kg, c_in / 64);
        if (c_out == 128) hipLaunchKernelGGL((k_wgrad_mfma_wide<4, 2>), grid, dim3(256), 0, s, a);
        else if (c_out == 64) hipLaunchKernelGGL((k_wgrad_mfma_wide<2, 2>), grid, dim3(128), 0, s, a);
        else hipLaunchKernelGGL((k_wgrad_mfma_wide<1, 2>), grid, dim3(64), 0, s, a);
    } else if (mfma) {
        const dim3 grid(splits, kg, c_in / 32);
        if (c_out == 128) hipLaunchKernelGGL((k_wgrad_mfma<4>), grid, dim3(256), 0, s, a);
        else if (c_out == 64) hipLaunchKernelGGL((k_wgrad_mfma<2>), grid, dim3(128), 0, s, a);
        else hipLaunchKernelGGL((k_wgrad_mfma<1>), grid, dim3(64), 0, s, a);
    } else if (!launch_small(a, splits, kg, s)) {
        hipLaunchKernelGGL(k_wgrad_valu, dim3(splits, kg), dim3(256), 0, s, a);
    }
    if (int rc = check_hip(hipGetLastError(), "k_wgrad")) return rc;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(blocks_for(count, 256)), dim3(256), 0, s, static_cast<const float *>(ws), splits,
                       count, dw, accumulate);
    return check_hip(hipGetLastError(), "k_wgrad_reduce");
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of fpcc_conv_f32's fused epilogue  y = act(pre + bias)  (act: none | ReLU | PReLU with ONE slope > 0), computed
// from the layer's OUTPUT: for slope > 0, pre < 0 <=> y < 0 and pre = y / slope there, so nothing but y has to be kept
// from the forward pass.       g = dy * act'(pre)      dbias[c] = sum_rows g[., c]      dslope = sum dy * pre [pre < 0]
// In the reference these are three autograd nodes (bias add, MinkowskiPReLU, their reductions) per layer
// (lib/minkowski_sparse_conv_layers.py:85-91).  Row blocks write partial sums, reduced in ascending block order.
namespace fpcc {
namespace {
constexpr int kEpiRows = 256;        // rows per workgroup

__global__ __launch_bounds__(256) void k_epilogue_bwd(const float *__restrict__ y, int ldy, const float *__restrict__ dy, int lddy,
                                                      int64_t n, int c, int cpad, int act, const float *__restrict__ slope,
                                                      float *__restrict__ g, int ldg, float *__restrict__ partial) {
    __shared__ float s_col[256];
    __shared__ float s_slope[256];
    const int col = threadIdx.x % cpad, grp = threadIdx.x / cpad, groups = 256 / cpad;
    const int64_t r0 = (int64_t)blockIdx.x * kEpiRows;
    const int64_t r1 = min(r0 + kEpiRows, n);
    const float sl = (act == FPCC_ACT_PRELU) ? slope[0] : 0.0f;
    float *dst = partial + (int64_t)blockIdx.x * (c + 1);
    for (int c0 = 0; c0 < c; c0 += cpad) {            // cpad < c only when c > 256
        const int cc = c0 + col;
        float sum_g = 0.0f, sum_s = 0.0f;
        if (cc < c) {
            for (int64_t r = r0 + grp; r < r1; r += groups) {
                const float yv = y[r * ldy + cc], d = dy[r * lddy + cc];
                float gv = d;
                if (act != FPCC_ACT_NONE && !(yv > 0.0f)) {       // pre <= 0 (torch's convention at 0: the negative branch)
                    gv = d * sl;                                 // sl == 0 for ReLU
                    if (act == FPCC_ACT_PRELU) sum_s = fmaf(d, yv / sl, sum_s);
                }
                g[r * ldg + cc] = gv;
                sum_g += gv;
            }
        }
        s_col[threadIdx.x] = sum_g;
        s_slope[threadIdx.x] = sum_s;
        __syncthreads();
        if (grp == 0 && cc < c) {
            float t = 0.0f;
            for (int q = 0; q < groups; ++q) t += s_col[q * cpad + col];
            dst[cc] = t;
        }
        if (threadIdx.x == 0) {
            float t = c0 == 0 ? 0.0f : dst[c];
            for (int q = 0; q < 256; ++q) t += s_slope[q];
            dst[c] = t;
        }
        __syncthreads();
    }
}

// Same, four columns per lane (16-byte loads and stores; c, the row strides and the pointers multiples of 4 floats): a row of 128
// channels is 32 lanes, a workgroup covers 8 rows per pass and keeps four passes in flight.  The scalar kernel above moved 1.4 TB/s
// over the training step's layers (4-byte accesses, one row per thread and pass); the sums per row block are formed in a fixed
// order here too (per lane over its rows ascending, then over the row groups ascending).
__global__ __launch_bounds__(256) void k_epilogue_bwd_v4(const float *__restrict__ y, int ldy, const float *__restrict__ dy, int lddy,
                                                         int64_t n, int c, int cpad4, int act, const float *__restrict__ slope,
                                                         float *__restrict__ g, int ldg, float *__restrict__ partial) {
    __shared__ float s_col[256 * 4];
    __shared__ float s_slope[256];
    const int col4 = threadIdx.x % cpad4, grp = threadIdx.x / cpad4, groups = 256 / cpad4;
    const int64_t r0 = (int64_t)blockIdx.x * kEpiRows;
    const int64_t r1 = min(r0 + kEpiRows, n);
    const float sl = (act == FPCC_ACT_PRELU) ? slope[0] : 0.0f;
    float *dst = partial + (int64_t)blockIdx.x * (c + 1);
    for (int c0 = 0; c0 < c; c0 += 4 * cpad4) {       // more than one pass only when c > 1024
        const int cc = c0 + 4 * col4;
        float sg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float sum_s = 0.0f;
        if (cc < c) {
#pragma unroll 4
            for (int64_t r = r0 + grp; r < r1; r += groups) {
                const float4 yv = *reinterpret_cast<const float4 *>(y + r * ldy + cc);
                const float4 d = *reinterpret_cast<const float4 *>(dy + r * lddy + cc);
                const float ys[4] = {yv.x, yv.y, yv.z, yv.w}, ds[4] = {d.x, d.y, d.z, d.w};
                float gv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gv[j] = ds[j];
                    if (act != FPCC_ACT_NONE && !(ys[j] > 0.0f)) {
                        gv[j] = ds[j] * sl;
                        if (act == FPCC_ACT_PRELU) sum_s = fmaf(ds[j], ys[j] / sl, sum_s);
                    }
                    sg[j] += gv[j];
                }
                *reinterpret_cast<float4 *>(g + r * ldg + cc) = make_float4(gv[0], gv[1], gv[2], gv[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s_col[threadIdx.x * 4 + j] = sg[j];
        s_slope[threadIdx.x] = sum_s;
        __syncthreads();
        if (grp == 0 && cc < c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = 0.0f;
                for (int q = 0; q < groups; ++q) t += s_col[(q * cpad4 + col4) * 4 + j];
                dst[cc + j] = t;
            }
        }
        if (threadIdx.x == 0) {
            float t = c0 == 0 ? 0.0f : dst[c];
            for (int q = 0; q < 256; ++q) t += s_slope[q];
            dst[c] = t;
        }
        __syncthreads();
    }
}

// one workgroup per column (column c = the slope term): threads stride over the row blocks, then a fixed LDS tree
__global__ __launch_bounds__(256) void k_epilogue_bwd_reduce(const float *__restrict__ partial, int64_t blocks, int c,
                                                             float *__restrict__ dbias, float *__restrict__ dslope) {
    __shared__ float s[256];
    const int cc = blockIdx.x;
    float t = 0.0f;
    for (int64_t b = threadIdx.x; b < blocks; b += 256) t += partial[b * (c + 1) + cc];
    s[threadIdx.x] = t;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (cc < c) { if (dbias) dbias[cc] = s[0]; }
        else if (dslope) dslope[0] = s[0];
    }
}
}  // namespace
}  // namespace fpcc

extern "C" int64_t fpcc_epilogue_bwd_ws_bytes(int64_t n, int c) {
    if (n < 0 || c < 1) return FPCC_E_ARG;
    return ((n + kEpiRows - 1) / kEpiRows) * (int64_t)(c + 1) * 4;
}

extern "C" int fpcc_epilogue_bwd_f32(const float *y, int ldy, const float *dy, int lddy, int64_t n, int c, int act,
                                     const float *slope, float *g, int ldg, float *dbias, float *dslope, void *ws,
                                     int64_t ws_bytes, void *stream) {
    if (n < 0 || c < 1 || ldy < c || lddy < c || ldg < c) return fail_arg("epilogue_bwd: sizes out of range");
    if (act != FPCC_ACT_NONE && act != FPCC_ACT_PRELU && act != FPCC_ACT_RELU) return fail_arg("epilogue_bwd: unknown activation");
    if (act == FPCC_ACT_PRELU && !slope) return fail_arg("epilogue_bwd: PReLU needs its slope");
    hipStream_t s = as_stream(stream);
    if (n == 0) {
        if (dbias) if (int rc = check_hip(hipMemsetAsync(dbias, 0, (size_t)c * 4, s), "hipMemsetAsync")) return rc;
        if (dslope) if (int rc = check_hip(hipMemsetAsync(dslope, 0, 4, s), "hipMemsetAsync")) return rc;
        return FPCC_OK;
    }
    if (!y || !dy || !g) return fail_arg("epilogue_bwd: null pointer");
    const int64_t blocks = (n + kEpiRows - 1) / kEpiRows;
    if (!ws || ws_bytes < blocks * (int64_t)(c + 1) * 4) return fail_arg("epilogue_bwd: workspace of fpcc_epilogue_bwd_ws_bytes() bytes required");
    const bool vec4 = c % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && ldg % 4 == 0 &&
                      ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    if (vec4) {
        int cpad4 = 1;
        while (cpad4 * 4 < c && cpad4 < 256) cpad4 <<= 1;
        hipLaunchKernelGGL(k_epilogue_bwd_v4, dim3((unsigned)blocks), dim3(256), 0, s, y, ldy, dy, lddy, n, c, cpad4, act, slope, g, ldg,
                           static_cast<float *>(ws));
    } else {
        int cpad = 1;
        while (cpad < c && cpad < 256) cpad <<= 1;
        hipLaunchKernelGGL(k_epilogue_bwd, dim3((unsigned)blocks), dim3(256), 0, s, y, ldy, dy, lddy, n, c, cpad, act, slope, g, ldg,
                           static_cast<float *>(ws));
    }
    if (int rc = check_hip(hipGetLastError(), "k_epilogue_bwd")) return rc;
    hipLaunchKernelGGL(k_epilogue_bwd_reduce, dim3(c + 1), dim3(256), 0, s, static_cast<const float *>(ws), blocks, c,
                       dbias, dslope);
    return check_hip(hipGetLastError(), "k_epilogue_bwd_reduce");
}

// ---------------------------------------------------------------------------------------------------------------
// Weights of the input-gradient convolution: wt[k][co][ci] = w[flip ? K-1-k : k][ci][co] (the mirrored offset's kernel,
// transposed) in one pass -- two tensor ops (flip + transposed copy) per layer and step otherwise.
namespace fpcc {
namespace {
__global__ void k_transpose_weights(const float *__restrict__ w, int K, int c_in, int c_out, int flip, float *__restrict__ wt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)c_in * c_out;
    if (e >= K * per) return;
    const int k = (int)(e / per);
    const int r = (int)(e - k * per);
    const int co = r / c_in, ci = r - co * c_in;
    wt[e] = w[(int64_t)(flip ? K - 1 - k : k) * per + (int64_t)ci * c_out + co];
}
}  // namespace
}  // namespace fpcc

extern "C" int fpcc_transpose_weights_f32(const float *w, int n_offsets, int c_in, int c_out, int flip, float *wt, void *stream) {
    if (n_offsets < 1 || c_in < 1 || c_out < 1) return fail_arg("transpose_weights: sizes out of range");
    if (!w || !wt) return fail_arg("transpose_weights: null pointer");
    const int64_t total = (int64_t)n_offsets * c_in * c_out;
    hipLaunchKernelGGL(k_transpose_weights, dim3(blocks_for(total, 256)), dim3(256), 0, as_stream(stream), w, n_offsets, c_in,
                       c_out, flip, wt);
    return check_hip(hipGetLastError(), "k_transpose_weights");
}
