// A CHAIN of per-point layers (MinkowskiLinear / 1x1x1 convolutions + bias + PReLU/ReLU/clamp, with one optional channel
// concatenation in the middle) as ONE launch for gfx950: the decoder blocks of the lossless coder
//     SubDecoderGeoLossl :  MLP(1 -> 64), MLP(64 -> 128), cat(., prediction[128]), MLP(256 -> 128), MLP(128 -> 128)
//     SubDecoderGeoLossl2:  MLP(C -> C'), MLP(C' -> C')
// (/root/reference/models/convolutional/lossy_coord_v2/layers.py:294-331) are four / two dependent launches per pyramid level and
// coding direction otherwise, each writing an activation matrix to HBM that the next one reads back.
//
// Unit of work = one WAVE and 32 rows, walked through every layer of the chain; the activations of the 32 rows never leave the
// CU: a layer's accumulators (MFMA layout: register r = row (r & 3) + 8 (r >> 2) + 4 h, lane = column) go through the wave's
// PRIVATE LDS tile (bias / activation / clamp applied on the way) and come back as the next layer's A operands in MFMA operand
// layout (lane (i, h) reads channels [8 g + 4 h, +4) of row i: one ds_read_b128 per group of 8 channels, row pitch 132 floats =
// conflict-free).  No workgroup barrier anywhere -- the tile is private, a wave's LDS operations execute in order.  B operands
// stream from the PACKED weights of fpcc_conv_pack_weights_f32 (L2-resident, 16-byte coalesced loads) with the in-place refill
// of k_conv_wave; the rows of a concatenated global operand (the prediction) are gathered like a per-point layer's A rows.
// Waves are persistent: wave w walks row blocks w, w + W, ...
//
// Every output element is the same fp32 FMA chain as in the separate launches (summation order 1: chunks of 32 channels
// ascending -- tile part first, then the concatenated part --, groups of 8 ascending, 0,4,1,5,2,6,3,7 inside a group; a
// one-channel first layer is fmaf(x, w[j], 0) + bias as in k_conv_c1_pointwise), so the fused and the unfused evaluation are
// bit-identical (tests/test_gpu_mlp_chain.py) and which of the two ran is not part of the stream format.
//
// Roofline: 2 * rows * sum(C_in * C_out) flop against the fp32 MFMA peak; HBM traffic is the chain's input + concatenated
// operand + output only (the unfused form moves every intermediate twice).
#include "common.h"

namespace fpcc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPitch = 132;            // floats per tile row: 128 channels + 4 (rows start on different banks; 16-byte aligned)
constexpr int kMaxLayers = FPCC_MLP_CHAIN_MAX_LAYERS;

struct Layer {
    const float *wp;        // packed weights [chunks][4][nbt][2][32][4] (c_in multiple of 32), or NULL for a one-channel first layer
    const float *w1;        // one-channel first layer: w[c_out]
    const float *bias;      // [c_out] or NULL
    const float *slope;     // device float[1] for PReLU
    int c_tile;             // input channels taken from the previous layer's output (the LDS tile); 0 for layer 0
    int c_glob;             // input channels taken from global memory AFTER the tile part: x (layer 0) or y (the concatenation)
    int c_out;              // 32 | 64 | 128
    int act;
    float clip;
};

struct ChainArgs {
    const float *x; int ldx;
    const float *y; int ldy;
    float *out; int ldo;
    int64_t n;
    int n_layers;
    unsigned n_row_blocks, n_waves;
    Layer L[kMaxLayers];
};

__device__ __forceinline__ float finish(float v, float b, int act, float slope, float clip) {
    v = v + b;
    if (act == FPCC_ACT_PRELU) v = v < 0.0f ? v * slope : v;
    else if (act == FPCC_ACT_RELU) v = v < 0.0f ? 0.0f : v;
    if (clip > 0.0f) v = fminf(fmaxf(v, -clip), clip);
    return v;
}

// acc += A[32 rows x 32 n_chunks] @ B over the chunks of one operand part.  `arow` points at this lane's A row + 4 h floats (LDS
// tile or global memory: the address space is resolved after inlining), `bp` at the part's first packed chunk + lane * 4.
// Operand registers are refilled in place group by group right after the MFMAs that consumed them (see k_conv_wave).
template <int NBW, typename AP>
__device__ __forceinline__ void accumulate(f32x16 (&acc)[NBW], AP arow, int n_chunks, const float *bp) {
    constexpr int kChunkFloats = 4 * NBW * 256;
    f32x4 ra[4], rb[4][NBW];
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8) {
        __builtin_amdgcn_sched_barrier(0);
        ra[g8] = *reinterpret_cast<const f32x4 *>(arow + 8 * g8);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            __builtin_amdgcn_sched_barrier(0);
            rb[g8][nb] = *reinterpret_cast<const f32x4 *>(bp + (g8 * NBW + nb) * 256);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int cc = 0; cc < n_chunks; ++cc) {
        const int cn = cc + 1 < n_chunks ? cc + 1 : cc;            // past the last chunk: re-read it, never used
        const auto an = arow + 32 * cn;
        const float *bn = bp + cn * kChunkFloats;
        __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
        for (int g8 = 0; g8 < 4; ++g8) {
            const f32x4 av = ra[g8];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, rb[g8][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, rb[g8][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, rb[g8][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, rb[g8][nb].w, acc[nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x6);
            ra[g8] = *reinterpret_cast<const f32x4 *>(an + 8 * g8);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                __builtin_amdgcn_sched_barrier(0x6);
                rb[g8][nb] = *reinterpret_cast<const f32x4 *>(bn + (g8 * NBW + nb) * 256);
            }
            __builtin_amdgcn_sched_barrier(0x6);
        }
    }
}

// one MFMA layer of the chain on this wave's 32 rows: tile (+ global part) -> tile
template <int NBW>
__device__ __forceinline__ void mfma_layer(const Layer &l, float *tile, const float *grow, int li, int lh, int lane) {
    f32x16 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    const float *bp = l.wp + lane * 4;
    const int nt = l.c_tile / 32, ng = l.c_glob / 32;
    if (nt > 0) accumulate<NBW>(acc, tile + li * kPitch + 4 * lh, nt, bp);
    if (ng > 0) accumulate<NBW>(acc, grow, ng, bp + (int64_t)nt * (4 * NBW * 256));
    // every read of the tile has been consumed by an MFMA above: the tile may be overwritten with this layer's output
    const float slope = (l.act == FPCC_ACT_PRELU && l.slope) ? l.slope[0] : 0.0f;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const float b = l.bias ? l.bias[32 * nb + li] : 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            tile[r * kPitch + 32 * nb + li] = finish(acc[nb][reg], b, l.act, slope, l.clip);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256, 2) void k_mlp_chain(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float s_tile[4][32 * kPitch];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned w = blockIdx.x * 4u + (unsigned)wv;
    if (w >= a.n_waves) return;                      // no workgroup barrier below: a wave may leave on its own
    float *tile = s_tile[wv];

    for (unsigned rbk = w; rbk < a.n_row_blocks; rbk += a.n_waves) {
        const int64_t row0 = (int64_t)rbk * 32;
        int64_t row = row0 + li;
        if (row >= a.n) row = a.n - 1;               // tail block: re-read the last row, never stored
#pragma unroll 1
        for (int l = 0; l < a.n_layers; ++l) {
            const Layer &L = a.L[l];
            if (!L.wp) {
                // one-channel first layer: an outer product, straight into the tile (lane = column, rows split over the halves)
                const float slope = (L.act == FPCC_ACT_PRELU && L.slope) ? L.slope[0] : 0.0f;
                const float xv = a.x[row * a.ldx];                              // lane li (both halves) holds row li's input
                const int nbw = L.c_out / 32;
                for (int nb = 0; nb < nbw; ++nb) {
                    const float wj = L.w1[32 * nb + li], b = L.bias ? L.bias[32 * nb + li] : 0.0f;
#pragma unroll
                    for (int r0 = 0; r0 < 32; r0 += 2) {
                        const int r = r0 + lh;
                        const float xr = __shfl(xv, r);
                        tile[r * kPitch + 32 * nb + li] = finish(fmaf(xr, wj, 0.0f), b, L.act, slope, L.clip);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            const float *grow = l == 0 ? a.x + row * a.ldx + 4 * lh : (L.c_glob ? a.y + row * a.ldy + 4 * lh : nullptr);
            if (L.c_out == 128) mfma_layer<4>(L, tile, grow, li, lh, lane);
            else if (L.c_out == 64) mfma_layer<2>(L, tile, grow, li, lh, lane);
            else mfma_layer<1>(L, tile, grow, li, lh, lane);
        }
        // the last layer's output: tile -> global, 16 bytes per lane, whole 128-byte lines per instruction
        const int c_last = a.L[a.n_layers - 1].c_out;
        const int lpr = c_last / 4;                                  // lanes per row: 8 | 16 | 32
        const int rpi = 64 / lpr;                                    // rows per store instruction
        const int qr = lane / lpr, qc = lane - qr * lpr;
        for (int r0 = 0; r0 < 32; r0 += rpi) {
            const int r = r0 + qr;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + r * kPitch + 4 * qc);
            const int64_t o = row0 + r;
            if (o < a.n) *reinterpret_cast<f32x4 *>(a.out + o * a.ldo + 4 * qc) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool width_ok(int c) { return c == 32 || c == 64 || c == 128; }

}  // namespace
}  // namespace fpcc

using namespace fpcc;

extern "C" int fpcc_mlp_chain_f32(const fpcc_mlp_chain *d, void *stream) {
    if (!d) return fail_arg("mlp_chain: null descriptor");
    if (d->n < 0 || d->n_layers < 1 || d->n_layers > kMaxLayers) return fail_arg("mlp_chain: 1..4 layers");
    if (d->n == 0) return FPCC_OK;
    if (!d->x || !d->out || !aligned16(d->out) || d->ldo % 4) return fail_arg("mlp_chain: null or unaligned input / output");
    if (d->cat_layer >= d->n_layers || (d->cat_layer >= 1 && (!d->y || !aligned16(d->y) || d->ldy % 4 || d->ldy < d->cy)))
        return fail_arg("mlp_chain: bad concatenation operand");
    if (d->cat_layer == 0) return fail_arg("mlp_chain: the first layer reads x only (pass the concatenation as a later layer)");
    ChainArgs a{};
    a.x = d->x; a.ldx = d->ldx; a.y = d->y; a.ldy = d->ldy; a.out = d->out; a.ldo = d->ldo; a.n = d->n; a.n_layers = d->n_layers;
    int prev = 0;
    for (int l = 0; l < d->n_layers; ++l) {
        const fpcc_mlp_layer &s = d->layers[l];
        Layer &t = a.L[l];
        if (!width_ok(s.c_out)) return fail_arg("mlp_chain: layer widths must be 32, 64 or 128");
        if (s.act != FPCC_ACT_NONE && s.act != FPCC_ACT_PRELU && s.act != FPCC_ACT_RELU) return fail_arg("mlp_chain: unknown activation");
        if (s.act == FPCC_ACT_PRELU && !s.slope) return fail_arg("mlp_chain: PReLU needs a slope pointer");
        t.bias = s.bias; t.slope = s.slope; t.c_out = s.c_out; t.act = s.act; t.clip = s.clip;
        if (l == 0) {
            t.c_tile = 0;
            if (d->cx == 1) {
                if (!s.w) return fail_arg("mlp_chain: a one-channel first layer needs its plain weights");
                if (d->ldx < 1) return fail_arg("mlp_chain: bad row stride");
                t.wp = nullptr; t.w1 = s.w; t.c_glob = 0;
            } else {
                if (d->cx % 32 || d->cx < 32 || d->cx > 256 || d->ldx < d->cx || d->ldx % 4 || !aligned16(d->x) || !s.w_packed ||
                    !aligned16(s.w_packed))
                    return fail_arg("mlp_chain: the first layer needs 1 or a multiple of 32 (<= 256) aligned input channels and packed weights");
                t.wp = s.w_packed; t.c_glob = d->cx;
            }
        } else {
            if (!s.w_packed || !aligned16(s.w_packed)) return fail_arg("mlp_chain: packed weights required");
            t.wp = s.w_packed; t.c_tile = prev; t.c_glob = 0;
            if (l == d->cat_layer) {
                if (d->cy % 32 || d->cy < 32 || d->cy > 128) return fail_arg("mlp_chain: the concatenated operand needs 32, 64, 96 or 128 channels");
                t.c_glob = d->cy;
            }
        }
        prev = s.c_out;
    }
    if (d->ldo < prev) return fail_arg("mlp_chain: output row stride smaller than the row");
    a.n_row_blocks = (unsigned)((d->n + 31) / 32);
    unsigned waves = 256u * 4u * 2u;                         // two waves per SIMD on the whole chip
    if (waves > a.n_row_blocks) waves = a.n_row_blocks;
    a.n_waves = waves;
    hipLaunchKernelGGL(k_mlp_chain, dim3((waves + 3) / 4), dim3(256), 0, as_stream(stream), a);
    return check_hip(hipGetLastError(), "k_mlp_chain");
}
