// fp32 sparse convolution for gfx950: output-stationary gather -> LDS -> v_mfma_f32_32x32x2_f32.
//
// One workgroup (4 waves) owns 128 consecutive output rows (rows are Morton-sorted, so their neighbourhoods overlap and
// the gathered input rows are L2-resident) and ALL output channels.  It walks the kernel offsets that occur in its tile
// and, per offset, the input channels in chunks of CH; a stage = (offset, chunk):
//     global (gathered rows of X, one 128-byte line per row and chunk; the W[k] chunk, contiguous)
//         -> registers (issued one stage ahead) -> LDS (double buffered, XOR-swizzled) -> MFMA fragments.
// Every output element is ONE fp32 FMA chain (offsets ascending, channels in the order documented in fpcc_hip.h), it is
// written once, nothing is scattered or atomically added: results are bitwise reproducible and independent of tiling,
// which the codec needs because the decoder must recompute the encoder's activations exactly.
// Offsets absent from a whole wave's 32 rows are skipped by that wave, offsets absent from the tile by the workgroup
// (adding x*0 would not change the chain's value).
//
// Roofline: 2*L*C_in*C_out algorithmic flop against the fp32 MFMA peak (157 TFLOP/s); the gathered bytes
// (L*C_in*4 from L2/MALL) are ~1/16 of what the MFMAs can consume at C_out = 128.
//
// Shapes the MFMA kernel does not cover (C_out in {1, 8, 16}, C_in = 1, ...) are tiny in this codec and go through the
// VALU kernel below: one thread per output row and block of <= 16 output channels, weights through the scalar cache.
#include "common.h"

namespace fpcc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float *x1; int c1; int ld1;
    const float *x2; int c2; int ld2;
    const int32_t *nbr; int n_off; int64_t nbr_ks; int64_t nbr_os;
    const float *w; const float *bias; int c_out; int groups;
    const int32_t *out_map; int64_t om_os; int64_t om_gs; float *out; int ldo; int64_t n_out;
    int act; const float *slope; float clip;
};

__device__ float g_zero_row[64];   // 256 bytes of zeros: the source of every absent neighbour

__device__ __forceinline__ float finish(float v, float b, int act, float slope, float clip) {
    v = v + b;
    if (act == FPCC_ACT_PRELU) v = v < 0.0f ? v * slope : v;
    else if (act == FPCC_ACT_RELU) v = v < 0.0f ? 0.0f : v;
    if (clip > 0.0f) v = fminf(fmaxf(v, -clip), clip);
    return v;
}

// blockIdx.x -> tile so that tiles adjacent in row order share an XCD (and therefore its L2); bijective for any grid size
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned n) {
    const unsigned q = n / 8, r = n % 8, x = bid % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
}

template <int CH>
__device__ __forceinline__ int swz(int r) {
    // 16-byte piece index XOR so that the 16 lanes of a ds_read_b128 group land on 16 different bank slots
    return CH == 32 ? ((r >> 1) & 7) : ((r >> 2) & 3);
}

constexpr int kMaxOffsets = 27;   // the MFMA kernel keeps the tile's neighbour indices in LDS: [27][rows]

// Tile geometry.  A workgroup of WM x WN waves owns TM = 32*WM output rows and all 32*NBT output columns; wave (wr, wc)
// computes rows [32*wr, 32*wr+32) x column blocks [wc*NBW, (wc+1)*NBW).  Large maps use 128-row tiles (4x1 waves);
// small maps use 64- or 32-row tiles with the waves spread over the columns instead, so that a level with a few
// thousand rows still fills the 256 CUs and its chain of (offset, chunk) stages is 4x shorter per wave.
template <int NBT, int CH, int WM, int WN>
struct MfmaCfg {
    static constexpr int C_OUT = 32 * NBT;
    static constexpr int TM = 32 * WM;
    static constexpr int NBW = NBT / WN;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int PPR = CH / 4;                               // 16-byte pieces per gathered row chunk
    static constexpr int A_TOTAL = TM * PPR;
    static constexpr int A_PIECES = (A_TOTAL + THREADS - 1) / THREADS;
    static constexpr int W_TOTAL = CH * C_OUT / 4;                   // 16-byte pieces of one weight chunk
    static constexpr int W_PIECES = (W_TOTAL + THREADS - 1) / THREADS;
    static_assert(NBT % WN == 0, "column blocks must divide over the waves");
};

// global -> registers for stage (k, cc)
template <typename C, int CH>
__device__ __forceinline__ void stage_fetch(const ConvArgs &a, const float *wg, const int32_t *s_nbr, int c_in, int tid,
                                            int k, int cc, f32x4 (&ra)[C::A_PIECES], f32x4 (&rw)[C::W_PIECES]) {
    // the chunk lies entirely in x1 or entirely in x2 (c1 is a multiple of CH): a scalar choice, no per-lane branch
    const bool in_x1 = cc * CH < a.c1;
    const float *xb = in_x1 ? a.x1 + cc * CH : a.x2 + (cc * CH - a.c1);
    const int64_t ld = in_x1 ? a.ld1 : a.ld2;
#pragma unroll
    for (int j = 0; j < C::A_PIECES; ++j) {
        const int p = tid + C::THREADS * j;
        if (C::A_TOTAL % C::THREADS == 0 || p < C::A_TOTAL) {
            const int r = p / C::PPR, q = p % C::PPR;
            const int32_t idx = s_nbr[k * C::TM + r];
            const float *src = idx >= 0 ? xb + (int64_t)idx * ld + 4 * q : (const float *)g_zero_row;
            ra[j] = *reinterpret_cast<const f32x4 *>(src);
        }
    }
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(wg + ((int64_t)k * c_in + (int64_t)cc * CH) * C::C_OUT);
#pragma unroll
    for (int j = 0; j < C::W_PIECES; ++j) {
        const int p = tid + C::THREADS * j;
        if (C::W_TOTAL % C::THREADS == 0 || p < C::W_TOTAL) rw[j] = wsrc[p];
    }
}

// registers -> LDS buffer
template <typename C, int CH>
__device__ __forceinline__ void stage_stash(float *dA, float *dWf, int tid, const f32x4 (&ra)[C::A_PIECES],
                                            const f32x4 (&rw)[C::W_PIECES]) {
#pragma unroll
    for (int j = 0; j < C::A_PIECES; ++j) {
        const int p = tid + C::THREADS * j;
        if (C::A_TOTAL % C::THREADS == 0 || p < C::A_TOTAL) {
            const int r = p / C::PPR, q = p % C::PPR;
            *reinterpret_cast<f32x4 *>(dA + r * CH + 4 * (q ^ swz<CH>(r))) = ra[j];
        }
    }
    f32x4 *dW = reinterpret_cast<f32x4 *>(dWf);
#pragma unroll
    for (int j = 0; j < C::W_PIECES; ++j) {
        const int p = tid + C::THREADS * j;
        if (C::W_TOTAL % C::THREADS == 0 || p < C::W_TOTAL) dW[p] = rw[j];
    }
}

template <typename C, int CH>
__device__ __forceinline__ void stage_compute(const float *cA, const float *cW, int wr, int wc, int li, int lh,
                                              f32x16 (&acc)[C::NBW]) {
    constexpr int C_OUT = C::C_OUT;
    constexpr int NBW = C::NBW;
    const int r = wr * 32 + li;
#pragma unroll
    for (int g8 = 0; g8 < CH / 8; ++g8) {
        const int q = 2 * g8 + lh;
        const f32x4 av = *reinterpret_cast<const f32x4 *>(cA + r * CH + 4 * (q ^ swz<CH>(r)));
        const float *wrow = cW + (8 * g8 + 4 * lh) * C_OUT + 32 * wc * NBW + li;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wrow[32 * nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wrow[C_OUT + 32 * nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wrow[2 * C_OUT + 32 * nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wrow[3 * C_OUT + 32 * nb], acc[nb], 0, 0, 0);
    }
}

template <int NBT, int CH, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_mfma(ConvArgs a) {
    using C = MfmaCfg<NBT, CH, WM, WN>;
    constexpr int C_OUT = C::C_OUT;
    constexpr int TM = C::TM;
    constexpr int NBW = C::NBW;

    __shared__ __attribute__((aligned(16))) float smem[2 * TM * CH + 2 * CH * C_OUT];
    __shared__ int32_t s_nbr[kMaxOffsets * TM];
    __shared__ unsigned s_mask[WM];
    float *sA = smem;
    float *sW = smem + 2 * TM * CH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WM, wc = wave / WM;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int g = blockIdx.y;
    const int64_t row0 = (int64_t)tile * TM;
    const int c_in = a.c1 + a.c2;
    const int n_chunks = c_in / CH;
    const float *wg = a.w + (int64_t)g * a.n_off * c_in * C_OUT;

    // the tile's neighbour table -> LDS (rows past the end count as absent)
    for (int e = tid; e < a.n_off * TM; e += C::THREADS) {
        const int k = e / TM, r = e % TM;
        const int64_t row = row0 + r;
        int32_t v = -1;
        if (row < a.n_out) v = a.nbr ? a.nbr[(int64_t)k * a.nbr_ks + row * a.nbr_os] : (int32_t)row;
        s_nbr[e] = v;
    }
    __syncthreads();
    // which offsets occur in this wave's 32 rows / in the whole tile
    unsigned wmask = 0;
    for (int k = 0; k < a.n_off; ++k)
        if (__ballot(s_nbr[k * TM + wr * 32 + li] >= 0) != 0ull) wmask |= 1u << k;
    if (lane == 0 && wc == 0) s_mask[wr] = wmask;
    __syncthreads();
    unsigned tmask = 0;
#pragma unroll
    for (int i = 0; i < WM; ++i) tmask |= s_mask[i];
    const int n_stages = __popc(tmask) * n_chunks;

    f32x16 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

    f32x4 ra[C::A_PIECES], rw[C::W_PIECES];

    if (n_stages > 0) {
        // stage iterator over (set bits of tmask) x chunks
        unsigned rest = tmask;
        int k_cur = __ffs(rest) - 1;
        int k_next = k_cur, cc_next = 0;
        stage_fetch<C, CH>(a, wg, s_nbr, c_in, tid, k_cur, 0, ra, rw);
        stage_stash<C, CH>(sA, sW, tid, ra, rw);
        __syncthreads();
        for (int s = 0; s < n_stages; ++s) {
            if (++cc_next == n_chunks) {
                cc_next = 0;
                rest &= rest - 1;
                k_next = rest ? __ffs(rest) - 1 : 0;
            }
            const bool more = s + 1 < n_stages;
            if (more) stage_fetch<C, CH>(a, wg, s_nbr, c_in, tid, k_next, cc_next, ra, rw);
            if ((wmask >> k_cur) & 1u)
                stage_compute<C, CH>(sA + (s & 1) * TM * CH, sW + (s & 1) * CH * C_OUT, wr, wc, li, lh, acc);
            if (more) stage_stash<C, CH>(sA + ((s + 1) & 1) * TM * CH, sW + ((s + 1) & 1) * CH * C_OUT, tid, ra, rw);
            __syncthreads();
            k_cur = k_next;
        }
    }

    const float slope = (a.act == FPCC_ACT_PRELU && a.slope) ? a.slope[0] : 0.0f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t o = row0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (o >= a.n_out) continue;
        const int64_t dst = a.out_map ? (int64_t)a.out_map[o * a.om_os + g * a.om_gs] : o * a.groups + g;
        if (dst < 0) continue;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int col = 32 * (wc * NBW + nb) + li;
            const float b = a.bias ? a.bias[col] : 0.0f;
            a.out[dst * a.ldo + col] = finish(acc[nb][reg], b, a.act, slope, a.clip);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Pair-compacted variant for large maps ("order 2").
//
// The dense tile above executes every row of a 32-row block for every offset present in the block; on surfaces only
// ~48 % of the (row, offset) pairs exist, so it runs ~1.8x the algorithmic flop.  Here a workgroup (8 waves, 128 output
// rows) compacts, per kernel offset, the rows that HAVE that neighbour, runs the MFMAs over ceil(count/32) packed row
// blocks only, and adds the finished per-offset partial sums into an fp32 accumulator tile in LDS (each (row, column)
// has one writer per offset, offsets are separated by barriers: plain read-modify-write, no atomics, reproducible).
// Summation order: per offset an FMA chain from zero (MFMA channel order), partial sums added in ascending offset
// order, then the bias -- "order 2" of fpcc_hip.h, the association of a per-offset gather-GEMM-scatter-add evaluation.
// Loads run two stages ahead through two register sets (one workgroup per CU: 145 KB of LDS).
constexpr int kCmpRows = 128;
constexpr int kCmpThreads = 512;

template <int NBT, int CH>
struct CmpCfg {
    static constexpr int C_OUT = 32 * NBT;
    static constexpr int NPG = 8 / NBT;                              // waves sharing a column block split the row passes
    static constexpr int MAXP = (4 + NPG - 1) / NPG;                 // passes per wave (a tile has at most 4)
    static constexpr int PPR = CH / 4;
    static constexpr int A_TOTAL = kCmpRows * PPR;
    static constexpr int A_PIECES = (A_TOTAL + kCmpThreads - 1) / kCmpThreads;
    static constexpr int W_TOTAL = CH * C_OUT / 4;
    static constexpr int W_PIECES = (W_TOTAL + kCmpThreads - 1) / kCmpThreads;
};

template <typename C, int CH>
__device__ __forceinline__ void cmp_fetch(const ConvArgs &a, const float *wg, const int32_t *s_in, int c_in, int tid, int k,
                                          int cc, int rows, f32x4 (&ra)[C::A_PIECES], f32x4 (&rw)[C::W_PIECES]) {
    const bool in_x1 = cc * CH < a.c1;
    const float *xb = in_x1 ? a.x1 + cc * CH : a.x2 + (cc * CH - a.c1);
    const int64_t ld = in_x1 ? a.ld1 : a.ld2;
#pragma unroll
    for (int j = 0; j < C::A_PIECES; ++j) {
        const int p = tid + kCmpThreads * j;
        const int r = p / C::PPR, q = p % C::PPR;
        // branch-free: rows past the packed count read the zero row (a branch around the load would make hipcc wait
        // vmcnt(0) per piece and serialise the gather)
        const int32_t idx = r < rows ? s_in[k * kCmpRows + r] : -1;
        const float *src = idx >= 0 ? xb + (int64_t)idx * ld + 4 * q : (const float *)g_zero_row;
        ra[j] = *reinterpret_cast<const f32x4 *>(src);
    }
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(wg + ((int64_t)k * c_in + (int64_t)cc * CH) * C::C_OUT);
#pragma unroll
    for (int j = 0; j < C::W_PIECES; ++j) {
        const int p = tid + kCmpThreads * j;
        if (C::W_TOTAL % kCmpThreads == 0 || p < C::W_TOTAL) rw[j] = wsrc[p];
    }
}

template <typename C, int CH>
__device__ __forceinline__ void cmp_stash(float *dA, float *dWf, int tid, int /*rows*/, const f32x4 (&ra)[C::A_PIECES],
                                          const f32x4 (&rw)[C::W_PIECES]) {
#pragma unroll
    for (int j = 0; j < C::A_PIECES; ++j) {
        const int p = tid + kCmpThreads * j;
        const int r = p / C::PPR, q = p % C::PPR;
        *reinterpret_cast<f32x4 *>(dA + r * CH + 4 * (q ^ swz<CH>(r))) = ra[j];
    }
    f32x4 *dW = reinterpret_cast<f32x4 *>(dWf);
#pragma unroll
    for (int j = 0; j < C::W_PIECES; ++j) {
        const int p = tid + kCmpThreads * j;
        if (C::W_TOTAL % kCmpThreads == 0 || p < C::W_TOTAL) dW[p] = rw[j];
    }
}

template <int NBT, int CH>
__global__ __launch_bounds__(kCmpThreads) void k_conv_cmp(ConvArgs a) {
    using C = CmpCfg<NBT, CH>;
    constexpr int C_OUT = C::C_OUT;
    constexpr int TM = kCmpRows;
    constexpr int MAXP = C::MAXP;

    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    float *sA = reinterpret_cast<float *>(dyn_smem);                       // [2][TM*CH]
    float *sW = sA + 2 * TM * CH;                                          // [2][CH*C_OUT]
    float *sOut = sW + 2 * CH * C_OUT;                                     // [TM][C_OUT]
    int32_t *s_in = reinterpret_cast<int32_t *>(sOut + TM * C_OUT);        // [27][TM] compact input rows (-1 padded)
    uint8_t *s_row = reinterpret_cast<uint8_t *>(s_in + kMaxOffsets * TM); // [27][TM] compact -> local output row
    int32_t *s_cnt = reinterpret_cast<int32_t *>(s_row + kMaxOffsets * TM);// [27]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int cb = wave % NBT, pg = wave / NBT;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)tile * TM;
    const int c_in = a.c1 + a.c2;
    const int n_chunks = c_in / CH;
    const float *wg = a.w;

    // --- per offset: compact the rows that have this neighbour (one wave per offset, 2 x 64 rows) ---------------------
    for (int k = wave; k < a.n_off; k += 8) {
        int base = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 64 * h + lane;
            const int64_t row = row0 + r;
            int32_t v = -1;
            if (row < a.n_out) v = a.nbr[(int64_t)k * a.nbr_ks + row * a.nbr_os];
            const unsigned long long m = __ballot(v >= 0);
            if (v >= 0) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                s_in[k * TM + pos] = v;
                s_row[k * TM + pos] = (uint8_t)r;
            }
            base += __popcll(m);
        }
        const int padded = (base + 31) & ~31;
        for (int j = base + lane; j < padded; j += 64) s_in[k * TM + j] = -1;
        if (lane == 0) s_cnt[k] = base;
    }
    for (int e = tid; e < TM * C_OUT; e += kCmpThreads) sOut[e] = 0.0f;
    __syncthreads();

    unsigned kmask = 0;
    for (int k = 0; k < a.n_off; ++k) if (s_cnt[k] > 0) kmask |= 1u << k;
    const int n_stages = __popc(kmask) * n_chunks;

    f32x16 acc[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    f32x4 ra0[C::A_PIECES], rw0[C::W_PIECES], ra1[C::A_PIECES], rw1[C::W_PIECES];

    // stage cursor: (offset, chunk) in ascending offset order
    struct Cursor { unsigned rest; int k, cc; };
    auto first = [&](Cursor &c) { c.rest = kmask; c.k = __ffs(c.rest) - 1; c.cc = 0; };
    auto next = [&](Cursor &c) {
        if (++c.cc == n_chunks) { c.cc = 0; c.rest &= c.rest - 1; c.k = c.rest ? __ffs(c.rest) - 1 : 0; }
    };
    auto rows_of = [&](int k) { return (s_cnt[k] + 31) & ~31; };

    auto compute = [&](int buf, int k, int cc) {
        const float *cA = sA + buf * TM * CH;
        const float *cW = sW + buf * CH * C_OUT;
        const int cnt = s_cnt[k];
        const int npass = (cnt + 31) >> 5;
        if (pg < npass) {
#pragma unroll
            for (int g8 = 0; g8 < CH / 8; ++g8) {
                const float *wrow = cW + (8 * g8 + 4 * lh) * C_OUT + 32 * cb + li;
                const float b0 = wrow[0], b1 = wrow[C_OUT], b2 = wrow[2 * C_OUT], b3 = wrow[3 * C_OUT];
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    const int p = pg + i * C::NPG;
                    if (p < npass) {
                        const int r = 32 * p + li;
                        const int q = 2 * g8 + lh;
                        const f32x4 av = *reinterpret_cast<const f32x4 *>(cA + r * CH + 4 * (q ^ swz<CH>(r)));
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc[i], 0, 0, 0);
                    }
                }
            }
            if (cc == n_chunks - 1) {
                // this offset's partial sums are complete: add them into the accumulator tile
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    const int p = pg + i * C::NPG;
                    if (p < npass) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int j = 32 * p + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                            if (j < cnt) {
                                float *dst = sOut + (int)s_row[k * TM + j] * C_OUT + 32 * cb + li;
                                *dst = *dst + acc[i][reg];
                            }
                            acc[i][reg] = 0.0f;
                        }
                    }
                }
            }
        }
    };

    if (n_stages > 0) {
        // Software pipeline, two slots per iteration (static register-set names): stage s is FETCHED in slot s, written to
        // LDS at the end of slot s+1 and COMPUTED in slot s+2, so a gather has two slots to land.  Every global load is
        // unconditional and the loop is entered with nothing in flight (the first two slots only fetch): with loads
        // under a branch, or in flight across the loop entry, hipcc cannot count the younger loads and waits vmcnt(0)
        // before the LDS writes, which would serialise the pipeline.  Past the last stage the cursor parks on a valid
        // stage whose data is never used.
        Cursor cf;
        first(cf);
        int k0 = 0, c0 = 0, k1 = 0, c1 = 0;          // stage held by register set 0 / 1
        int kb0 = 0, cb0 = 0, kb1 = 0, cb1 = 0;      // stage resident in LDS buffer 0 / 1
        for (int it = 0; it < n_stages + 2; it += 2) {
            {   // even slot: fetch stage `it` -> set 0; compute stage it-2 (buffer 0); set 1 (stage it-1) -> buffer 1
                const int ke = cf.k, ce = cf.cc;
                cmp_fetch<C, CH>(a, wg, s_in, c_in, tid, ke, ce, rows_of(ke), ra0, rw0);
                next(cf);
                if (it >= 2) compute(0, kb0, cb0);
                if (it >= 1) {
                    cmp_stash<C, CH>(sA + TM * CH, sW + CH * C_OUT, tid, 0, ra1, rw1);
                    kb1 = k1;
                    cb1 = c1;
                }
                k0 = ke;
                c0 = ce;
                __syncthreads();
            }
            {   // odd slot: fetch stage it+1 -> set 1; compute stage it-1 (buffer 1); set 0 (stage it) -> buffer 0
                const int ko = cf.k, co = cf.cc;
                cmp_fetch<C, CH>(a, wg, s_in, c_in, tid, ko, co, rows_of(ko), ra1, rw1);
                next(cf);
                if (it >= 1 && it - 1 < n_stages) compute(1, kb1, cb1);
                cmp_stash<C, CH>(sA, sW, tid, 0, ra0, rw0);
                kb0 = k0;
                cb0 = c0;
                k1 = ko;
                c1 = co;
                __syncthreads();
            }
        }
    }
    __syncthreads();

    // --- epilogue: accumulator tile -> bias, activation, clamp -> global, 16 bytes per thread ---------------------------
    const float slope = (a.act == FPCC_ACT_PRELU && a.slope) ? a.slope[0] : 0.0f;
    constexpr int VPR = C_OUT / 4;
    for (int e = tid; e < TM * VPR; e += kCmpThreads) {
        const int r = e / VPR, v = e % VPR;
        const int64_t o = row0 + r;
        if (o >= a.n_out) continue;
        const int64_t dst = a.out_map ? (int64_t)a.out_map[o * a.om_os] : o;
        if (dst < 0) continue;
        const f32x4 t = *reinterpret_cast<const f32x4 *>(sOut + r * C_OUT + 4 * v);
        f32x4 res;
        res.x = finish(t.x, a.bias ? a.bias[4 * v] : 0.0f, a.act, slope, a.clip);
        res.y = finish(t.y, a.bias ? a.bias[4 * v + 1] : 0.0f, a.act, slope, a.clip);
        res.z = finish(t.z, a.bias ? a.bias[4 * v + 2] : 0.0f, a.act, slope, a.clip);
        res.w = finish(t.w, a.bias ? a.bias[4 * v + 3] : 0.0f, a.act, slope, a.clip);
        float *orow = a.out + dst * a.ldo + 4 * v;
        if ((a.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {
            *reinterpret_cast<f32x4 *>(orow) = res;
        } else {
            orow[0] = res.x; orow[1] = res.y; orow[2] = res.z; orow[3] = res.w;
        }
    }
}

template <int NBT, int CH>
constexpr size_t cmp_lds_bytes() {
    return sizeof(float) * (2 * kCmpRows * CH + 2 * CH * 32 * NBT + kCmpRows * 32 * NBT) +
           sizeof(int32_t) * kMaxOffsets * kCmpRows + kMaxOffsets * kCmpRows + sizeof(int32_t) * 32;
}

template <int NBT, int CH>
int launch_cmp(const ConvArgs &a, hipStream_t s) {
    const unsigned tiles = (unsigned)((a.n_out + kCmpRows - 1) / kCmpRows);
    constexpr size_t lds = cmp_lds_bytes<NBT, CH>();
    static bool configured = false;
    if (!configured) {
        if (int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_cmp<NBT, CH>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                               "hipFuncSetAttribute(k_conv_cmp)"))
            return rc;
        configured = true;
    }
    hipLaunchKernelGGL((k_conv_cmp<NBT, CH>), dim3(tiles), dim3(kCmpThreads), lds, s, a);
    return check_hip(hipGetLastError(), "k_conv_cmp");
}

// ---------------------------------------------------------------------------------------------------------------
// VALU path: thread = (output row, group, block of JB output channels); natural channel order.
template <int JB>
__global__ __launch_bounds__(256) void k_conv_valu(ConvArgs a, int n_jb) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y / n_jb, jb = blockIdx.y % n_jb;
    if (o >= a.n_out) return;
    const int64_t dst = a.out_map ? (int64_t)a.out_map[o * a.om_os + g * a.om_gs] : o * a.groups + g;
    if (dst < 0) return;
    const int c_in = a.c1 + a.c2;
    const int j0 = jb * JB;
    const float *wg = a.w + (int64_t)g * a.n_off * c_in * a.c_out;

    float acc[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) acc[j] = 0.0f;

    for (int k = 0; k < a.n_off; ++k) {
        const int32_t idx = a.nbr ? a.nbr[(int64_t)k * a.nbr_ks + o * a.nbr_os] : (int32_t)o;
        if (idx < 0) continue;
        const float *xr1 = a.x1 + (int64_t)idx * a.ld1;
        const float *wk = wg + (int64_t)k * c_in * a.c_out + j0;
        for (int c = 0; c < a.c1; ++c) {
            const float xv = xr1[c];
            const float *wr = wk + (int64_t)c * a.c_out;
#pragma unroll
            for (int j = 0; j < JB; ++j)
                if (JB == 1 || j0 + j < a.c_out) acc[j] = fmaf(xv, wr[j], acc[j]);
        }
        if (a.c2 > 0) {
            const float *xr2 = a.x2 + (int64_t)idx * a.ld2;
            for (int c = 0; c < a.c2; ++c) {
                const float xv = xr2[c];
                const float *wr = wk + (int64_t)(a.c1 + c) * a.c_out;
#pragma unroll
                for (int j = 0; j < JB; ++j)
                    if (JB == 1 || j0 + j < a.c_out) acc[j] = fmaf(xv, wr[j], acc[j]);
            }
        }
    }
    const float slope = (a.act == FPCC_ACT_PRELU && a.slope) ? a.slope[0] : 0.0f;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        if (j0 + j < a.c_out) {
            const float b = a.bias ? a.bias[j0 + j] : 0.0f;
            a.out[dst * a.ldo + j0 + j] = finish(acc[j], b, a.act, slope, a.clip);
        }
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 0: VALU kernel, 16 / 32: MFMA kernel with that chunk size
int mfma_chunk(int c1, int c2, int c_out) {
    if (c_out != 32 && c_out != 64 && c_out != 128) return 0;
    const int c_in = c1 + c2;
    if (c_in % 32 == 0 && c1 % 32 == 0) return 32;
    if (c_in % 16 == 0 && c1 % 16 == 0) return 16;
    return 0;
}

template <int NBT, int CH, int WM, int WN>
int launch_mfma_cfg(const ConvArgs &a, hipStream_t s) {
    constexpr int TM = 32 * WM;
    const unsigned tiles = (unsigned)((a.n_out + TM - 1) / TM);
    hipLaunchKernelGGL((k_conv_mfma<NBT, CH, WM, WN>), dim3(tiles, a.groups), dim3(64 * WM * WN), 0, s, a);
    return check_hip(hipGetLastError(), "k_conv_mfma");
}

// pick the tile height so that the launch has at least ~2 workgroups per CU when the map allows it
template <int NBT, int CH>
int launch_mfma(const ConvArgs &a, hipStream_t s) {
    const int64_t work = a.n_out * a.groups;
    constexpr int WNS = NBT >= 2 ? 2 : 1;       // 64-row tile: 2 x WNS waves
    if (work >= 128 * 1024) return launch_mfma_cfg<NBT, CH, 4, 1>(a, s);
    if (work >= 64 * 512) return launch_mfma_cfg<NBT, CH, 2, WNS>(a, s);
    return launch_mfma_cfg<NBT, CH, 1, NBT>(a, s);
}

template <int JB>
int launch_valu(const ConvArgs &a, hipStream_t s) {
    const int n_jb = (a.c_out + JB - 1) / JB;
    hipLaunchKernelGGL((k_conv_valu<JB>), dim3(blocks_for(a.n_out, 256), a.groups * n_jb), dim3(256), 0, s, a, n_jb);
    return check_hip(hipGetLastError(), "k_conv_valu");
}

}  // namespace
}  // namespace fpcc

using namespace fpcc;

// rows from which the pair-compacted kernel is used for multi-offset convolutions
constexpr int64_t kCmpMinRows = 32 * 1024;

// Measured on MI355X (profiles/r01): the compacted kernel executes ~35 % fewer MFMAs but runs one workgroup per CU
// (145 KB of LDS); it wins for C_in >= 128, C_out = 128 and loses to the dense tile for narrower layers.
static bool use_cmp(int c1, int c2, int c_out, int n_offsets, int groups, int64_t n_out) {
    return mfma_chunk(c1, c2, c_out) == 32 && c1 + c2 >= 128 && c_out == 128 && n_offsets >= 8 &&
           n_offsets <= kMaxOffsets && groups == 1 && n_out >= kCmpMinRows;
}

extern "C" int fpcc_conv_f32_order(int c1, int c2, int c_out) { return mfma_chunk(c1, c2, c_out) ? 1 : 0; }

extern "C" int fpcc_conv_f32_order_ex(int c1, int c2, int c_out, int n_offsets, int groups, int64_t n_out) {
    if (use_cmp(c1, c2, c_out, n_offsets, groups, n_out)) return 2;
    return mfma_chunk(c1, c2, c_out) ? 1 : 0;
}

extern "C" int fpcc_conv_f32(const float *x1, int c1, int ld1, const float *x2, int c2, int ld2, const int32_t *nbr,
                             int n_offsets, int64_t nbr_ks, int64_t nbr_os, const float *w, const float *bias, int c_out,
                             int groups, const int32_t *out_map, int64_t om_os, int64_t om_gs, float *out, int ldo,
                             int64_t n_out, int act, const float *slope, float clip, void *stream) {
    if (n_out < 0 || c1 < 1 || c2 < 0 || c_out < 1 || groups < 1 || n_offsets < 1 || n_offsets > 32)
        return fail_arg("conv_f32: sizes out of range (n_offsets must be 1..32)");
    if (n_out == 0) return FPCC_OK;
    if (!x1 || !w || !out || (c2 > 0 && !x2)) return fail_arg("conv_f32: null pointer");
    if (!nbr && n_offsets != 1) return fail_arg("conv_f32: identity map needs n_offsets == 1");
    if (ld1 < c1 || (c2 > 0 && ld2 < c2) || ldo < c_out) return fail_arg("conv_f32: row stride smaller than the row");
    if (act != FPCC_ACT_NONE && act != FPCC_ACT_PRELU && act != FPCC_ACT_RELU) return fail_arg("conv_f32: unknown activation");
    if (act == FPCC_ACT_PRELU && !slope) return fail_arg("conv_f32: PReLU needs a slope pointer");
    if ((int64_t)groups * ((c_out + 15) / 16) > 65535) return fail_arg("conv_f32: too many groups");
    if (n_out == 0) return FPCC_OK;

    ConvArgs a{x1, c1, ld1, x2, c2, ld2, nbr, n_offsets, nbr_ks, nbr_os, w, bias, c_out, groups,
               out_map, om_os, om_gs, out, ldo, n_out, act, slope, clip};
    hipStream_t s = as_stream(stream);
    int ch = mfma_chunk(c1, c2, c_out);
    if (ch && n_offsets > kMaxOffsets) return fail_arg("conv_f32: the MFMA path supports at most 27 kernel offsets");
    if (ch && !(aligned16(x1) && ld1 % 4 == 0 && aligned16(w) && (c2 == 0 || (aligned16(x2) && ld2 % 4 == 0))))
        return fail_arg("conv_f32: the MFMA path needs 16-byte aligned inputs and row strides that are multiples of 4");
    if (nbr && use_cmp(c1, c2, c_out, n_offsets, groups, n_out)) {
        if (c_out == 128) return launch_cmp<4, 32>(a, s);
        if (c_out == 64) return launch_cmp<2, 32>(a, s);
        return launch_cmp<1, 32>(a, s);
    }
    if (ch == 32) {
        if (c_out == 128) return launch_mfma<4, 32>(a, s);
        if (c_out == 64) return launch_mfma<2, 32>(a, s);
        return launch_mfma<1, 32>(a, s);
    }
    if (ch == 16) {
        if (c_out == 128) return launch_mfma<4, 16>(a, s);
        if (c_out == 64) return launch_mfma<2, 16>(a, s);
        return launch_mfma<1, 16>(a, s);
    }
    if (c_out == 1) return launch_valu<1>(a, s);
    if (c_out <= 4) return launch_valu<4>(a, s);
    if (c_out <= 8) return launch_valu<8>(a, s);
    return launch_valu<16>(a, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Row gather (features re-ordered into the canonical Morton row order of a coordinate map).
namespace fpcc {
namespace {
__global__ void k_gather_rows(const float *__restrict__ x, int c, int ld, const int32_t *__restrict__ index, int64_t n,
                              float *__restrict__ out, int ldo) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * c) return;
    const int64_t r = e / c;
    const int j = (int)(e - r * c);
    out[r * ldo + j] = x[(int64_t)index[r] * ld + j];
}
}  // namespace
}  // namespace fpcc

extern "C" int fpcc_gather_rows_f32(const float *x, int c, int ld, const int32_t *index, int64_t n, float *out, int ldo,
                                    void *stream) {
    if (n < 0 || c < 1 || (n > 0 && (!x || !index || !out))) return fail_arg("gather_rows: null pointer or bad size");
    if (n == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks_for(n * c, 256)), dim3(256), 0, as_stream(stream), x, c, ld, index, n,
                       out, ldo);
    return check_hip(hipGetLastError(), "k_gather_rows");
}

// ---------------------------------------------------------------------------------------------------------------
// Second half of a single-output-channel 3x3x3 convolution.  With C_out == 1 the convolution is
//      out[o] = sum_k < X[nbr_k(o), :], w_k >  =  sum_k Y[nbr_k(o), k],      Y = X @ [w_0 | w_1 | ... | w_26]
// so the dot products run once per INPUT row on the MFMA kernel (a pointwise GEMM with 27 -> 32 output columns) and this
// kernel only gathers 27 scalars per output row: 27 x 4 B instead of 27 x C_in x 4 B of gather traffic.
// Summation order ("order 2" of fpcc_hip.h): per offset the dot product is its own FMA chain from zero, the offsets'
// partial sums are added in ascending offset order, then the bias -- which is the order of a per-offset
// gather-GEMM-scatter-add evaluation.
namespace fpcc {
namespace {
__global__ __launch_bounds__(256) void k_gather_sum(const float *__restrict__ y, int ldy, const int32_t *__restrict__ nbr,
                                                    int n_off, int64_t nbr_ks, int64_t nbr_os, int64_t n,
                                                    const float *__restrict__ bias, int act,
                                                    const float *__restrict__ slope, float clip, float *__restrict__ out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    float acc = 0.0f;
    for (int k = 0; k < n_off; ++k) {
        const int32_t idx = nbr[(int64_t)k * nbr_ks + o * nbr_os];
        if (idx >= 0) acc = acc + y[(int64_t)idx * ldy + k];
    }
    const float sl = (act == FPCC_ACT_PRELU && slope) ? slope[0] : 0.0f;
    out[o] = finish(acc, bias ? bias[0] : 0.0f, act, sl, clip);
}
}  // namespace
}  // namespace fpcc

extern "C" int fpcc_gather_sum_f32(const float *y, int ldy, const int32_t *nbr, int n_offsets, int64_t nbr_ks,
                                   int64_t nbr_os, int64_t n, const float *bias, int act, const float *slope, float clip,
                                   float *out, void *stream) {
    if (n < 0 || n_offsets < 1 || ldy < n_offsets) return fail_arg("gather_sum: bad sizes");
    if (n == 0) return FPCC_OK;
    if (!y || !nbr || !out) return fail_arg("gather_sum: null pointer");
    if (act == FPCC_ACT_PRELU && !slope) return fail_arg("gather_sum: PReLU needs a slope pointer");
    hipLaunchKernelGGL(k_gather_sum, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), y, ldy, nbr, n_offsets,
                       nbr_ks, nbr_os, n, bias, act, slope, clip, out);
    return check_hip(hipGetLastError(), "k_gather_sum");
}
