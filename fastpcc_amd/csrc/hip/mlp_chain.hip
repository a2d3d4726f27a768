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
#include <cstdlib>
#include <type_traits>

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

// finish() with the activation and the clamp resolved once per layer (both are wave-uniform run-time values: left inside the
// per-element code they become a chain of scalar branches around every one of the 16 accumulator registers)
template <int ACT, bool CLIP>
__device__ __forceinline__ float finish_t(float v, float b, float slope, float clip) {
    v = v + b;
    if (ACT == FPCC_ACT_PRELU) v = v < 0.0f ? v * slope : v;
    else if (ACT == FPCC_ACT_RELU) v = v < 0.0f ? 0.0f : v;
    if (CLIP) v = fminf(fmaxf(v, -clip), clip);
    return v;
}

template <typename F>
__device__ __forceinline__ void with_epilogue(int act, float clip, F &&f) {
    if (clip > 0.0f) {
        if (act == FPCC_ACT_PRELU) f(std::integral_constant<int, FPCC_ACT_PRELU>(), std::true_type());
        else if (act == FPCC_ACT_RELU) f(std::integral_constant<int, FPCC_ACT_RELU>(), std::true_type());
        else f(std::integral_constant<int, FPCC_ACT_NONE>(), std::true_type());
    } else {
        if (act == FPCC_ACT_PRELU) f(std::integral_constant<int, FPCC_ACT_PRELU>(), std::false_type());
        else if (act == FPCC_ACT_RELU) f(std::integral_constant<int, FPCC_ACT_RELU>(), std::false_type());
        else f(std::integral_constant<int, FPCC_ACT_NONE>(), std::false_type());
    }
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
    const float clip = l.clip;
    with_epilogue(l.act, clip, [&](auto act_c, auto clip_c) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const float b = l.bias ? l.bias[32 * nb + li] : 0.0f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                tile[r * kPitch + 32 * nb + li] = finish_t<decltype(act_c)::value, decltype(clip_c)::value>(acc[nb][reg], b, slope, clip);
            }
        }
    });
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


// ---------------------------------------------------------------------------------------------------------------
// Workgroup form for the chain shapes of the codecs (template instances below): a workgroup of four waves owns a tile of 64
// rows; wave w owns column block w of every 128-wide layer (two waves per column block and one row block each for 64-wide
// layers) and keeps ITS B operands of EVERY layer in registers for the whole launch (224 VGPRs for the four-layer decoder
// block: one wave per SIMD) -- no weight traffic at all in steady state, where the wave form above re-streams 224 KB of packed
// weights from L2 for every 32 rows.  Activations ping-pong between two LDS tiles (one barrier per layer), the concatenated
// operand and the next tile's inputs are staged through registers while the current tile computes, the output leaves through
// the tile as whole 128-byte lines.  A unit of work is 64 rows x one layer-set = 448 MFMAs per wave (wave form: 896 on 32 rows
// x all columns), so maps of a few hundred to a few ten thousand rows spread over four times as many waves and the partial last
// round of a launch is a quarter as long.  Same FMA chains: bit-identical to the wave form and to the separate launches.
constexpr int kTileRows = 64;
constexpr int kTileFloats = kTileRows * kPitch;

struct WgArgs {
    const float *x; int ldx;
    const float *y; int ldy;
    float *out; int ldo;
    int64_t n;
    unsigned n_tiles;
    const float *wp[kMaxLayers];      // packed weights (NULL for a one-channel first layer)
    const float *w1;                  // one-channel first layer: w[c_out]
    const float *bias[kMaxLayers];
    const float *slope[kMaxLayers];
    int act[kMaxLayers];
    float clip[kMaxLayers];
};

template <int G>
__device__ __forceinline__ void load_b(f32x4 (&B)[G], const float *wp, int nbt, int cb, int lane) {
#pragma unroll
    for (int g = 0; g < G; ++g) B[g] = *reinterpret_cast<const f32x4 *>(wp + ((int64_t)g * nbt + cb) * 256 + lane * 4);
}

// one MFMA layer on the workgroup's 64-row tile: A from `tin` (GT groups of 8 channels) and then from `ty` (GY groups), this
// wave's column block `cb` of the output into `tout`.  NCB = column blocks of the layer (4: every wave does both row blocks;
// 2: waves {w, w + 2} share a column block and take one row block each).
template <int GT, int GY, int NCB>
__device__ __forceinline__ void wg_layer(const float *tin, const float *ty, float *tout, const f32x4 (&B)[GT + GY], float bias, int act,
                                         float slope, float clip, int wv, int li, int lh) {
    constexpr int G = GT + GY, NRG = 4 / NCB;
    const int cb = wv % NCB, rg = wv / NCB;
    for (int rb = rg; rb < 2; rb += NRG) {
        const float *arow = tin + (32 * rb + li) * kPitch + 4 * lh;
        const float *yrow = GY ? ty + (32 * rb + li) * kPitch + 4 * lh : arow;
        auto frag = [&](int g) -> f32x4 {
            return g < GT ? *reinterpret_cast<const f32x4 *>(arow + 8 * g) : *reinterpret_cast<const f32x4 *>(yrow + 8 * (g - GT));
        };
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // A fragments three groups (12 MFMAs) ahead of their use; sched_barrier keeps them there (hipcc otherwise sinks every LDS
        // read to just before its first use and the wave -- alone on its SIMD -- waits out the LDS latency once per group)
        f32x4 a0 = frag(0), a1 = frag(G > 1 ? 1 : 0), a2 = frag(G > 2 ? 2 : 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 a3 = g + 3 < G ? frag(g + 3) : a0;
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, B[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, B[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, B[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, B[g].w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = a1; a1 = a2; a2 = a3;
        }
        with_epilogue(act, clip, [&](auto act_c, auto clip_c) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = 32 * rb + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                tout[r * kPitch + 32 * cb + li] = finish_t<decltype(act_c)::value, decltype(clip_c)::value>(acc[reg], bias, slope, clip);
            }
        });
    }
}

// rows [row0, row0 + 64) x C channels of a global matrix -> registers (C / 4 float4 per row over 256 threads), zeros past n
template <int C>
__device__ __forceinline__ void stage_load(f32x4 (&v)[C / 16], const float *src, int ld, int64_t row0, int64_t n, int t) {
    constexpr int Q = C / 4;                       // float4 per row
#pragma unroll
    for (int k = 0; k < C / 16; ++k) {
        const int idx = t + 256 * k, r = idx / Q, c4 = idx - r * Q;
        v[k] = row0 + r < n ? *reinterpret_cast<const f32x4 *>(src + (row0 + r) * ld + 4 * c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
}

template <int C>
__device__ __forceinline__ void stage_store(const f32x4 (&v)[C / 16], float *tile, int t) {
    constexpr int Q = C / 4;
#pragma unroll
    for (int k = 0; k < C / 16; ++k) {
        const int idx = t + 256 * k, r = idx / Q, c4 = idx - r * Q;
        *reinterpret_cast<f32x4 *>(tile + r * kPitch + 4 * c4) = v[k];
    }
}

template <int C>
__device__ __forceinline__ void tile_to_global(const float *tile, float *out, int ldo, int64_t row0, int64_t n, int t) {
    constexpr int Q = C / 4;
#pragma unroll
    for (int k = 0; k < C / 16; ++k) {
        const int idx = t + 256 * k, r = idx / Q, c4 = idx - r * Q;
        if (row0 + r < n) *reinterpret_cast<f32x4 *>(out + (row0 + r) * ldo + 4 * c4) = *reinterpret_cast<const f32x4 *>(tile + r * kPitch + 4 * c4);
    }
}

// SubDecoderGeoLossl: x [n, 1] -> W0 (outer product) -> W1 -> cat(., y [n, CY]) -> W2 -> W3
template <int W0, int W1, int CY, int W2, int W3>
__global__ __launch_bounds__(256, 1) void k_mlp_chain_a(WgArgs a) {
    static_assert(W1 == 128 && W2 == 128 && W3 == 128 && W0 % 32 == 0 && CY % 32 == 0, "column blocks of the MFMA layers are split over four waves");
    __shared__ __attribute__((aligned(16))) float s_p[kTileFloats], s_q[kTileFloats], s_y[kTileFloats];
    __shared__ float s_x[kTileRows];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, lh = lane >> 5;
    f32x4 B1[W0 / 8], B2[(W1 + CY) / 8], B3[W2 / 8];
    load_b(B1, a.wp[1], W1 / 32, wv, lane);
    load_b(B2, a.wp[2], W2 / 32, wv, lane);
    load_b(B3, a.wp[3], W3 / 32, wv, lane);
    const int col0 = t % W0;
    const float w0 = a.w1[col0], b0 = a.bias[0] ? a.bias[0][col0] : 0.0f;
    float bia[4], slp[4];
#pragma unroll
    for (int l = 1; l < 4; ++l) bia[l] = a.bias[l] ? a.bias[l][32 * wv + li] : 0.0f;
#pragma unroll
    for (int l = 0; l < 4; ++l) slp[l] = (a.act[l] == FPCC_ACT_PRELU && a.slope[l]) ? a.slope[l][0] : 0.0f;

    f32x4 yv[CY / 16];
    float xv = 0.0f;
    unsigned tile = blockIdx.x;
    if (tile < a.n_tiles) {
        const int64_t row0 = (int64_t)tile * kTileRows;
        stage_load<CY>(yv, a.y, a.ldy, row0, a.n, t);
        if (t < kTileRows) xv = row0 + t < a.n ? a.x[(row0 + t) * a.ldx] : 0.0f;
    }
    for (; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t row0 = (int64_t)tile * kTileRows;
        stage_store<CY>(yv, s_y, t);
        if (t < kTileRows) s_x[t] = xv;
        __syncthreads();
        // the next tile's inputs travel while this one computes
        const unsigned nxt = tile + gridDim.x;
        if (nxt < a.n_tiles) {
            const int64_t r1 = (int64_t)nxt * kTileRows;
            stage_load<CY>(yv, a.y, a.ldy, r1, a.n, t);
            if (t < kTileRows) xv = r1 + t < a.n ? a.x[(r1 + t) * a.ldx] : 0.0f;
        }
        // layer 0: one input channel, an outer product over all 256 threads (thread = column, rows strided)
        with_epilogue(a.act[0], a.clip[0], [&](auto act_c, auto clip_c) {
#pragma unroll
            for (int r = t / W0; r < kTileRows; r += 256 / W0)
                s_p[r * kPitch + col0] = finish_t<decltype(act_c)::value, decltype(clip_c)::value>(fmaf(s_x[r], w0, 0.0f), b0, slp[0], a.clip[0]);
        });
        __syncthreads();
        wg_layer<W0 / 8, 0, 4>(s_p, s_p, s_q, B1, bia[1], a.act[1], slp[1], a.clip[1], wv, li, lh);
        __syncthreads();
        wg_layer<W1 / 8, CY / 8, 4>(s_q, s_y, s_p, B2, bia[2], a.act[2], slp[2], a.clip[2], wv, li, lh);
        __syncthreads();
        wg_layer<W2 / 8, 0, 4>(s_p, s_p, s_q, B3, bia[3], a.act[3], slp[3], a.clip[3], wv, li, lh);
        __syncthreads();
        tile_to_global<W3>(s_q, a.out, a.ldo, row0, a.n, t);
        // s_q is read here and next written by layer 1 of the following tile, two barriers further on; s_y / s_x are rewritten at
        // the top of the loop, after every wave has passed the barrier behind the layer that read them
    }
}

// SubDecoderGeoLossl2: x [n, CX] -> W -> W (two square-ish MFMA layers of equal width)
template <int CX, int W>
__global__ __launch_bounds__(256, 1) void k_mlp_chain_b(WgArgs a) {
    static_assert((W == 128 || W == 64) && CX % 32 == 0, "");
    constexpr int NCB = W / 32;
    __shared__ __attribute__((aligned(16))) float s_p[kTileFloats], s_q[kTileFloats];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, lh = lane >> 5;
    const int cb = wv % NCB;
    f32x4 B0[CX / 8], B1[W / 8];
    load_b(B0, a.wp[0], NCB, cb, lane);
    load_b(B1, a.wp[1], NCB, cb, lane);
    float bia[2], slp[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        bia[l] = a.bias[l] ? a.bias[l][32 * cb + li] : 0.0f;
        slp[l] = (a.act[l] == FPCC_ACT_PRELU && a.slope[l]) ? a.slope[l][0] : 0.0f;
    }
    f32x4 xv[CX / 16];
    unsigned tile = blockIdx.x;
    if (tile < a.n_tiles) stage_load<CX>(xv, a.x, a.ldx, (int64_t)tile * kTileRows, a.n, t);
    for (; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t row0 = (int64_t)tile * kTileRows;
        stage_store<CX>(xv, s_p, t);
        __syncthreads();
        const unsigned nxt = tile + gridDim.x;
        if (nxt < a.n_tiles) stage_load<CX>(xv, a.x, a.ldx, (int64_t)nxt * kTileRows, a.n, t);
        wg_layer<CX / 8, 0, NCB>(s_p, s_p, s_q, B0, bia[0], a.act[0], slp[0], a.clip[0], wv, li, lh);
        __syncthreads();
        wg_layer<W / 8, 0, NCB>(s_q, s_q, s_p, B1, bia[1], a.act[1], slp[1], a.clip[1], wv, li, lh);
        __syncthreads();
        tile_to_global<W>(s_p, a.out, a.ldo, row0, a.n, t);
        __syncthreads();                              // s_p is refilled with the next tile's input right away
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Narrow two-layer head C0 -> C1 -> 1 per point (the decoder's classify block, layers.py:105-110: 16 -> 8 -> 1 on the 8 N
// candidate voxels): far too narrow for MFMA tiles -- evaluated separately the first layer is zero-padded to 32 output columns
// (a [8N, 32] matrix written and a strided quarter of it read back).  One thread per row: C0 inputs as 16-byte loads, the C1
// hidden values in registers, weights through LDS.  The hidden layer's FMA chain follows the order the separate launch would
// use for this row count (order 1: 0,4,1,5,2,6,3,7 inside groups of 8 -- the padded MFMA evaluation; order 0: natural), the
// output layer's the natural order: same bits.
template <int C0, int C1>
__global__ __launch_bounds__(256) void k_pointwise_head(const float *__restrict__ x, int ldx, const float *__restrict__ w1,
                                                        const float *__restrict__ b1, int act1, const float *__restrict__ slope1, int order1,
                                                        const float *__restrict__ w2, const float *__restrict__ b2, int act2,
                                                        const float *__restrict__ slope2, float clip, float *__restrict__ out, int64_t n) {
    __shared__ float s_w1[C0 * C1], s_b1[C1], s_w2[C1];
    for (int e = threadIdx.x; e < C0 * C1; e += 256) {
        // row c of s_w1 = the weights of the c-th channel IN CHAIN ORDER
        const int pos = e / C1, j = e - pos * C1;
        const int c = order1 ? (pos & ~7) + ((pos & 1) ? 4 : 0) + ((pos & 7) >> 1) : pos;
        s_w1[e] = w1[c * C1 + j];
    }
    if (threadIdx.x < C1) { s_b1[threadIdx.x] = b1 ? b1[threadIdx.x] : 0.0f; s_w2[threadIdx.x] = w2[threadIdx.x]; }
    __syncthreads();
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    float xv[C0];
#pragma unroll
    for (int q = 0; q < C0 / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + o * ldx + 4 * q);
        xv[4 * q] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
    }
    float h[C1];
#pragma unroll
    for (int j = 0; j < C1; ++j) h[j] = 0.0f;
#pragma unroll
    for (int pos = 0; pos < C0; ++pos) {
        const int c = order1 ? (pos & ~7) + ((pos & 1) ? 4 : 0) + ((pos & 7) >> 1) : pos;     // compile-time per branch after unrolling
        const float xc = order1 ? xv[(pos & ~7) + ((pos & 1) ? 4 : 0) + ((pos & 7) >> 1)] : xv[pos];
        (void)c;
#pragma unroll
        for (int j = 0; j < C1; ++j) h[j] = fmaf(xc, s_w1[pos * C1 + j], h[j]);
    }
    const float sl1 = (act1 == FPCC_ACT_PRELU && slope1) ? slope1[0] : 0.0f;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < C1; ++j) acc = fmaf(finish(h[j], s_b1[j], act1, sl1, 0.0f), s_w2[j], acc);
    const float sl2 = (act2 == FPCC_ACT_PRELU && slope2) ? slope2[0] : 0.0f;
    out[o] = finish(acc, b2 ? b2[0] : 0.0f, act2, sl2, clip);
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool width_ok(int c) { return c == 32 || c == 64 || c == 128; }

}  // namespace
}  // namespace fpcc

using namespace fpcc;

// 1 (default; FPCC_MLP_CHAIN_WG): the codecs' chain shapes run in the workgroup form, 0: everything in the wave form.  Result-neutral.
static int g_chain_form = [] { const char *e = getenv("FPCC_MLP_CHAIN_WG"); return e ? atoi(e) : 1; }();

extern "C" int fpcc_mlp_chain_set_form(int form) {
    const int before = g_chain_form;
    if (form >= 0) g_chain_form = form;
    return before;
}

extern "C" int fpcc_mlp_chain_f32(const fpcc_mlp_chain *d, void *stream) {
    LaunchBracket timed(stream);
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
    // the chain shapes of the codecs run in the workgroup form (weights in registers); FPCC_MLP_CHAIN_WG=0 keeps the wave form
    if (g_chain_form != 0) {
        WgArgs g{};
        g.x = d->x; g.ldx = d->ldx; g.y = d->y; g.ldy = d->ldy; g.out = d->out; g.ldo = d->ldo; g.n = d->n;
        g.n_tiles = (unsigned)((d->n + kTileRows - 1) / kTileRows);
        for (int l = 0; l < d->n_layers; ++l) {
            g.wp[l] = a.L[l].wp; g.bias[l] = a.L[l].bias; g.slope[l] = a.L[l].slope; g.act[l] = a.L[l].act; g.clip[l] = a.L[l].clip;
        }
        g.w1 = a.L[0].w1;
        const unsigned grid = g.n_tiles < 256u ? g.n_tiles : 256u;
        const fpcc_mlp_layer *L = d->layers;
        const bool y_ok = d->cat_layer == 2 && d->cy == 128 && aligned16(d->y) && d->ldy % 4 == 0;
        if (d->n_layers == 4 && d->cx == 1 && y_ok && L[0].c_out == 64 && L[1].c_out == 128 && L[2].c_out == 128 && L[3].c_out == 128) {
            hipLaunchKernelGGL((k_mlp_chain_a<64, 128, 128, 128, 128>), dim3(grid), dim3(256), 0, as_stream(stream), g);
            return check_hip(hipGetLastError(), "k_mlp_chain_a");
        }
        if (d->n_layers == 2 && d->cat_layer < 0 && d->cx == 128 && L[0].c_out == 128 && L[1].c_out == 128) {
            hipLaunchKernelGGL((k_mlp_chain_b<128, 128>), dim3(grid), dim3(256), 0, as_stream(stream), g);
            return check_hip(hipGetLastError(), "k_mlp_chain_b");
        }
        if (d->n_layers == 2 && d->cat_layer < 0 && d->cx == 64 && L[0].c_out == 64 && L[1].c_out == 64) {
            hipLaunchKernelGGL((k_mlp_chain_b<64, 64>), dim3(grid), dim3(256), 0, as_stream(stream), g);
            return check_hip(hipGetLastError(), "k_mlp_chain_b");
        }
    }
    a.n_row_blocks = (unsigned)((d->n + 31) / 32);
    unsigned waves = 256u * 4u * 2u;                         // two waves per SIMD on the whole chip
    if (waves > a.n_row_blocks) waves = a.n_row_blocks;
    a.n_waves = waves;
    hipLaunchKernelGGL(k_mlp_chain, dim3((waves + 3) / 4), dim3(256), 0, as_stream(stream), a);
    return check_hip(hipGetLastError(), "k_mlp_chain");
}

extern "C" int fpcc_pointwise_head_f32(const float *x, int c0, int ldx, const float *w1, const float *b1, int c1, int act1,
                                       const float *slope1, int order1, const float *w2, const float *b2, int act2,
                                       const float *slope2, float clip, float *out, int64_t n, void *stream) {
    LaunchBracket timed(stream);
    if (n < 0 || ldx < c0 || ldx % 4) return fail_arg("pointwise_head: bad sizes");
    if (!((c0 == 16 && c1 == 8) || (c0 == 8 && c1 == 4))) return fail_arg("pointwise_head: shapes 16->8->1 and 8->4->1");
    if (order1 != 0 && order1 != 1) return fail_arg("pointwise_head: hidden-layer order must be 0 (natural) or 1 (MFMA chain)");
    if (n == 0) return FPCC_OK;
    if (!x || !w1 || !w2 || !out || !aligned16(x)) return fail_arg("pointwise_head: null or unaligned pointer");
    if ((act1 == FPCC_ACT_PRELU && !slope1) || (act2 == FPCC_ACT_PRELU && !slope2)) return fail_arg("pointwise_head: PReLU needs a slope pointer");
    const dim3 grid(blocks_for(n, 256)), block(256);
    hipStream_t s = as_stream(stream);
    if (c0 == 16) hipLaunchKernelGGL((k_pointwise_head<16, 8>), grid, block, 0, s, x, ldx, w1, b1, act1, slope1, order1, w2, b2, act2, slope2, clip, out, n);
    else hipLaunchKernelGGL((k_pointwise_head<8, 4>), grid, block, 0, s, x, ldx, w1, b1, act1, slope1, order1, w2, b2, act2, slope2, clip, out, n);
    return check_hip(hipGetLastError(), "k_pointwise_head");
}
