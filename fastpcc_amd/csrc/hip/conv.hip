// fp32 sparse convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 157.3 TFLOP/s dense peak).
//
// Every output element is ONE fp32 FMA chain -- kernel offsets ascending, channels in the order documented in
// fpcc_hip.h ("summation order") -- written once, never scattered or atomically added, so results are bitwise
// reproducible and independent of tiling, row order and which of the kernels below ran.  The codec needs that: the
// decoder must recompute the encoder's activations exactly.  An absent neighbour contributes x*0 from a zero row, or is
// skipped where a whole wave lacks the offset (neither changes the chain's value).
//
// Three kernels, chosen per launch (launch_conv):
//   k_conv_wave   wave-autonomous, the default for 32-multiple channel counts with packed weights (fpcc_conv_f32_pk):
//                 a wave owns 32 output rows x NBW 32-column blocks; per (offset, 32-channel chunk) it gathers its A
//                 fragment straight from the input rows into registers in MFMA layout (8 dword loads per lane, absent
//                 rows redirected to the zero row by a bitwise pointer select) and streams B from the packed weights
//                 wp[offset][chunk][g8][nb][h][i][j] as 16-byte coalesced loads that stay in L2.  No LDS, no barrier; the
//                 operands of group g are refilled in place right after the MFMAs that consumed them, with
//                 sched_barrier pinning the load issue points (hipcc otherwise sinks them to their first use).
//   k_conv_mfma   workgroup-tiled: 4 waves own RT*32 rows and all output columns, X rows and W chunks go through
//                 double-buffered XOR-swizzled LDS.  Used when weights are not packed (training, changing weights),
//                 for 16-channel chunks and for transposed maps (out_map).
//   k_conv_valu   shapes MFMA tiles do not cover (C_out in {1, 8, 16}, C_in = 1 ...): one thread per output row and
//                 <= 16 output channels, weights through the scalar cache.  Bandwidth-bound, tiny in this codec.
// Multi-offset layers with 32-multiple channel counts are evaluated GROUPED (summation order 3 -- part of the stream format, see
// FPCC_NUMERICS_VERSION): the kernel offsets form four fixed groups, one wave of a workgroup each, partial sums added in group
// order -- by four waves of a workgroup that meet in LDS (k_conv_wave<..., OG = 4>: maps below 100 K rows, which need the
// parallelism) or by one wave that folds its accumulator into a running sum at every group boundary (k_conv_wave<..., FOLD>: large
// maps); both leave the same bits.
//
// Roofline: 2 * pairs * C_in * C_out algorithmic flop against the fp32 MFMA peak.  Measured limits of this design are in
// profiles/r02/wave_kernel_sweeps.md and profiles/r03/{sq_counters,grouped_occupancy,grouped_fold,clock_ramp}.md: each VMEM
// instruction costs ~30 SIMD cycles of issue that more waves do not hide, the B stream (256 B per MFMA from L2) caps the wave
// kernel at ~100 TFLOP/s on the 272 K-row maps at 2.4 GHz, and inside a codec step the power management grants ~2.16 GHz.
#include "conv_common.h"
#include <atomic>
#include <cstdlib>

namespace fpcc {
namespace {

// Tile geometry.  A workgroup of WM x WN waves owns TM = 32*WM output rows and all 32*NBT output columns; wave (wr, wc)
// computes rows [32*wr, 32*wr+32) x column blocks [wc*NBW, (wc+1)*NBW).  Large maps use 128-row tiles (4x1 waves);
// small maps use 64- or 32-row tiles with the waves spread over the columns instead, so that a level with a few
// thousand rows still fills the 256 CUs and its chain of (offset, chunk) stages is 4x shorter per wave.
//
// Operand paths.  A (gathered input rows): global -> REGISTERS, in the MFMA operand layout -- lane (i, h) of a wave
// holds channels [8g+4h, 8g+4h+4) of row i as one 16-byte load per group g of 8 channels, which is exactly the A
// operand of the two k-steps that group feeds; no LDS, no barrier, and a wave whose 32 rows lack the current kernel
// offset issues no loads at all.  B (weights of one (offset, chunk): CH x C_OUT): global -> registers -> LDS, shared by
// the waves of the workgroup, double buffered, one barrier per stage.  LDS per workgroup is 2*CH*C_OUT*4 + 27*TM*4
// bytes (46 KB for the 128 x 128 tile), so three workgroups share a CU.
template <int NBT, int CH, int WM, int WN>
struct MfmaCfg {
    static constexpr int C_OUT = 32 * NBT;
    static constexpr int TM = 32 * WM;
    static constexpr int NBW = NBT / WN;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int G8 = CH / 8;                                // groups of 8 channels per chunk
    static constexpr int W_TOTAL = CH * C_OUT / 4;                   // 16-byte pieces of one weight chunk
    static constexpr int W_PIECES = (W_TOTAL + THREADS - 1) / THREADS;
    static_assert(NBT % WN == 0, "column blocks must divide over the waves");
};

// A operand of stage (k, cc) for this wave's 32 rows: global -> registers
template <typename C, int CH>
__device__ __forceinline__ void fetch_a(const ConvArgs &a, const int32_t *nbr_rows, int cc, int li, int lh,
                                        f32x4 (&ra)[C::G8]) {
    // the chunk lies entirely in x1 or entirely in x2 (c1 is a multiple of CH): a scalar choice, no per-lane branch
    const bool in_x1 = cc * CH < a.c1;
    const float *xb = in_x1 ? a.x1 + cc * CH : a.x2 + (cc * CH - a.c1);
    const int64_t ld = in_x1 ? a.ld1 : a.ld2;
    const int32_t idx = nbr_rows[li];
    const float *src = (idx >= 0 ? xb + (int64_t)idx * ld : (const float *)g_zero_row) + 4 * lh;
#pragma unroll
    for (int g8 = 0; g8 < C::G8; ++g8) ra[g8] = *reinterpret_cast<const f32x4 *>(src + 8 * g8);
}

// B operand of stage (k, cc): global -> registers (whole workgroup)
template <typename C, int CH>
__device__ __forceinline__ void fetch_w(const float *wg, int c_in, int tid, int k, int cc, f32x4 (&rw)[C::W_PIECES]) {
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(wg + ((int64_t)k * c_in + (int64_t)cc * CH) * C::C_OUT);
#pragma unroll
    for (int j = 0; j < C::W_PIECES; ++j) {
        const int p = tid + C::THREADS * j;
        if (C::W_TOTAL % C::THREADS == 0 || p < C::W_TOTAL) rw[j] = wsrc[p];
    }
}

// registers -> LDS buffer
template <typename C>
__device__ __forceinline__ void stash_w(float *dWf, int tid, const f32x4 (&rw)[C::W_PIECES]) {
    f32x4 *dW = reinterpret_cast<f32x4 *>(dWf);
#pragma unroll
    for (int j = 0; j < C::W_PIECES; ++j) {
        const int p = tid + C::THREADS * j;
        if (C::W_TOTAL % C::THREADS == 0 || p < C::W_TOTAL) dW[p] = rw[j];
    }
}

template <typename C, int CH>
__device__ __forceinline__ void stage_compute(const f32x4 (&ra)[C::G8], const float *cW, int wc, int li, int lh,
                                              f32x16 (&acc)[C::NBW]) {
    constexpr int C_OUT = C::C_OUT;
    constexpr int NBW = C::NBW;
#pragma unroll
    for (int g8 = 0; g8 < C::G8; ++g8) {
        const f32x4 av = ra[g8];
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

// DBG (timing experiments only, results are wrong): bit 0 = every gathered row is row 0, bit 1 = every stage reads the weights of
// chunk 0, bit 2 = no per-stage barrier (races on the W buffers)
template <int NBT, int CH, int WM, int WN, int DBG = 0>
__global__ __launch_bounds__(64 * WM * WN, 3) void k_conv_mfma(ConvArgs a) {
    using C = MfmaCfg<NBT, CH, WM, WN>;
    constexpr int C_OUT = C::C_OUT;
    constexpr int TM = C::TM;
    constexpr int NBW = C::NBW;

    __shared__ __attribute__((aligned(16))) float sW[2 * CH * C_OUT];
    __shared__ int32_t s_nbr[kMaxOffsets * TM];
    __shared__ int32_t s_row[TM];          // output row of each tile position (-1 past the end)
    __shared__ unsigned s_mask[WM];
    __shared__ int32_t s_zero_idx[32];     // DBG bit 0: thirty-two times row 0
    if (DBG & 1) { if (threadIdx.x < 32) s_zero_idx[threadIdx.x] = 0; }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WM, wc = wave / WM;
    const int li = lane & 31, lh = lane >> 5;
    // Morton order: contiguous tile ranges per XCD (shared L2).  With a row order the gathers have no locality to keep and
    // the tiles come heaviest first (fpcc_conv_tile_keys): they are taken in dispatch order, so consecutive -- equally
    // heavy -- tiles land on different XCDs and the light tiles fill the tail of the launch (longest-processing-time-first:
    // 272 K-row layer 1193 -> 1132 us, 70 K-row 429 -> 390 us, 18 K-row 213 -> 200 us against a strided deal).
    const unsigned tile = a.row_order ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int g = blockIdx.y;
    const int n_off = a.n_off;
    const int64_t row0 = (int64_t)tile * TM;
    const int c_in = a.c1 + a.c2;
    const int n_chunks = c_in / CH;
    const float *wg = a.w + (int64_t)g * a.n_off * c_in * C_OUT;

    for (int r = tid; r < TM; r += C::THREADS)
        s_row[r] = row0 + r < a.n_out ? (a.row_order ? a.row_order[row0 + r] : (int32_t)(row0 + r)) : -1;
    // the tile's neighbour table -> LDS (rows past the end count as absent)
    const bool by_pos = a.row_order && table_is_row_major(a);
    for (int e = tid; e < n_off * TM; e += C::THREADS) {
        const int k = e / TM, r = e % TM;
        int32_t v = -1;
        if (row0 + r < a.n_out) {
            const int64_t row = a.row_order ? (int64_t)a.row_order[row0 + r] : row0 + r;
            // (a row-major table beside a row order holds its rows in position order: conv_common.h)
            v = a.nbr ? a.nbr[(int64_t)k * a.nbr_ks + (by_pos ? row0 + r : row) * a.nbr_os] : (int32_t)row;
        }
        s_nbr[e] = v;
    }
    __syncthreads();
    // which offsets occur in this wave's 32 rows / in the whole tile
    unsigned wmask = 0;
    for (int k = 0; k < n_off; ++k)
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

    f32x4 ra_cur[C::G8], ra_nxt[C::G8], rw[C::W_PIECES];
#pragma unroll
    for (int g8 = 0; g8 < C::G8; ++g8) ra_nxt[g8] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int32_t *my_nbr = s_nbr + wr * 32;

    if (n_stages > 0) {
        // stage iterator over (set bits of tmask) x chunks
        unsigned rest = tmask;
        int k_cur = __ffs(rest) - 1;
        int k_next = k_cur, cc_next = 0;
        // Every fetch below is unconditional (a wave without the offset reads the zero row, the stage after the last
        // re-reads a valid one): loads under a branch make hipcc drain vmcnt(0) right after issuing them.
        fetch_a<C, CH>(a, my_nbr + k_cur * TM, 0, li, lh, ra_nxt);
        fetch_w<C, CH>(wg, c_in, tid, k_cur, 0, rw);
        stash_w<C>(sW, tid, rw);
        __syncthreads();
        for (int s = 0; s < n_stages; ++s) {
            if (++cc_next == n_chunks) {
                cc_next = 0;
                rest &= rest - 1;
                k_next = rest ? __ffs(rest) - 1 : k_cur;
            }
#pragma unroll
            for (int g8 = 0; g8 < C::G8; ++g8) ra_cur[g8] = ra_nxt[g8];
            fetch_a<C, CH>(a, (DBG & 1) ? s_zero_idx : my_nbr + k_next * TM, cc_next, li, lh, ra_nxt);
            fetch_w<C, CH>(wg, c_in, tid, (DBG & 2) ? 0 : k_next, (DBG & 2) ? 0 : cc_next, rw);
            if ((wmask >> k_cur) & 1u) stage_compute<C, CH>(ra_cur, sW + (s & 1) * CH * C_OUT, wc, li, lh, acc);
            stash_w<C>(sW + ((s + 1) & 1) * CH * C_OUT, tid, rw);
            if (!(DBG & 4)) __syncthreads();
            k_cur = k_next;
        }
    }

    // Every load the epilogue needs (slope, bias, output-map entries) BEFORE the first store: a load placed between two stores makes
    // the wave wait for it with vmcnt(0), i.e. for every store issued so far -- a full memory round trip per output row
    // (profiles/r04/prologue_epilogue.md: the stores of a wave took 36-70 K cycles on a loaded chip).
    const float slope = (a.act == FPCC_ACT_PRELU && a.slope) ? a.slope[0] : 0.0f;
    float bias[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) bias[nb] = a.bias ? a.bias[32 * (wc * NBW + nb) + li] : 0.0f;
    int64_t dsts[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t o = s_row[wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh];
        dsts[reg] = o < 0 ? -1 : a.out_map ? (int64_t)a.out_map[o * a.om_os + g * a.om_gs] : o * a.groups + g;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t dst = dsts[reg];
        if (dst < 0) continue;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
            a.out[dst * a.ldo + 32 * (wc * NBW + nb) + li] = finish(acc[nb][reg], bias[nb], a.act, slope, a.clip);
    }
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
    const bool vec4 = a.c1 % 4 == 0 && a.ld1 % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x1) & 15) == 0;

    if (c_in == 1 && a.n_off == 27 && a.nbr) {
        // one input channel (the first layer): 27 independent index loads, 27 independent gathers, then the chain -- the
        // general loop below issues them one dependent pair at a time
        int32_t idx[27];
        float xv[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) idx[k] = a.nbr[(int64_t)k * a.nbr_ks + o * a.nbr_os];
#pragma unroll
        for (int k = 0; k < 27; ++k) xv[k] = a.x1[(int64_t)(idx[k] >= 0 ? idx[k] : 0) * a.ld1];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const float *wr = wg + (int64_t)k * a.c_out + j0;
#pragma unroll
            for (int j = 0; j < JB; ++j)
                if (JB == 1 || j0 + j < a.c_out) acc[j] = idx[k] >= 0 ? fmaf(xv[k], wr[j], acc[j]) : acc[j];
        }
    } else
    for (int k = 0; k < a.n_off; ++k) {
        const int32_t idx = a.nbr ? a.nbr[(int64_t)k * a.nbr_ks + o * a.nbr_os] : (int32_t)o;
        if (idx < 0) continue;
        const float *xr1 = a.x1 + (int64_t)idx * a.ld1;
        const float *wk = wg + (int64_t)k * c_in * a.c_out + j0;
        if (JB == 1 && vec4) {
            // one output channel: the input row as 16-byte loads (same natural-order chain)
            for (int c = 0; c < a.c1; c += 4) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr1 + c);
                acc[0] = fmaf(xv.x, wk[(int64_t)c * a.c_out], acc[0]);
                acc[0] = fmaf(xv.y, wk[(int64_t)(c + 1) * a.c_out], acc[0]);
                acc[0] = fmaf(xv.z, wk[(int64_t)(c + 2) * a.c_out], acc[0]);
                acc[0] = fmaf(xv.w, wk[(int64_t)(c + 3) * a.c_out], acc[0]);
            }
        } else
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

// ---------------------------------------------------------------------------------------------------------------
// Wave-autonomous MFMA path ("wave kernel").  Every wave is its own work unit: 32 output rows x NBW column blocks of 32,
// no LDS operand staging and NO workgroup barrier anywhere in the main loop.
//   A (gathered rows): global -> registers in MFMA operand layout, one stage ahead (as in k_conv_mfma);
//   B (weights): global/L2 -> registers from a PACKED copy of the weights (fpcc_conv_pack_weights_f32) in which the four
//     B operands lane (i, h) needs for one group of 8 channels and one column block are 16 contiguous bytes and the 64
//     lanes of a wave read 1 KB contiguous:  wp[m][cc][g8][nb][h][i][j] = w[m][32 cc + 8 g8 + 4 h + j][32 nb + i].
//     The weights of a layer (<= 1.8 MB) stay L2-resident; a wave re-loads the B registers of group g8 for the NEXT stage
//     right after the MFMAs of group g8 of the current one, so every load has a whole stage of MFMAs to land.
// The 4 waves of a workgroup only share the launch; each walks the kernel offsets present in ITS 32 rows.  What this buys
// over the workgroup-tiled kernel: no barrier skew between waves whose 32-row blocks have different offsets, no
// LDS round trip of W, work units of 1/2 .. 1/8 the size (shorter tail of a launch), and with NBW = 4 every gathered row is
// fetched once instead of twice.  Same summation order (order 1).
template <int NBW, int CH>
struct WaveCfg {
    static constexpr int G8 = CH / 8;
    static constexpr int REGS = 16 * NBW + 16 * (1 + NBW) + 28;              // accumulators + operands + addresses
    static constexpr int MIN_WAVES = REGS <= 96 ? 5 : REGS <= 128 ? 4 : 3;   // per SIMD
};

// Operand registers: ONE A buffer and ONE B buffer refilled in place group by group -- right after the MFMAs of group g8 have
// consumed ra[g8] / rb[g8][*] the same registers are loaded with group g8 of the NEXT stage, so every load has a whole stage
// of MFMAs to land and nothing is copied.  (A ring of 2-4 whole-stage buffers, refilled after the stage's last MFMA, was
// 30-50 % slower on every map size: profiles/r02/wave_kernel_sweeps.md.)
// DBG (timing experiments only, results are wrong): bit 0 = every gathered row is row 0 (no gather traffic), bit 1 = every
// stage reads the weights of chunk 0 (B stream stays in the vector L1)
// OG = 4 ("grouped" evaluation, summation order 3): the four waves of a workgroup share ONE unit and split its kernel offsets into
// the four fixed contiguous groups of offset_group_begin(); each wave runs the order-1 chain of its group from zero, the partial
// sums meet in LDS and are added as ((g0 + g1) + g2) + g3, then bias / activation.  A unit's serial chain of (offset, chunk)
// stages is four times shorter and the launch has four times the waves: what maps of a few thousand to a few ten thousand rows
// lack (one partial round of waves, each latency-bound on its own chain).  The groups are a function of the offset index alone,
// so a row's result does not depend on which rows share its block.
// DBG bit 4 (value 16): s_memtime stamps of one wave's life -- kernel entry, neighbour table read, first operands requested, the top
// of every (offset, chunk) stage, loop end, partial sums exchanged, outputs stored -- for profiles/r04/small_level_stage.md.  A stamp
// is one LDS store by lane 0 with the exec mask narrowed in place (no branch: a branch in the stage loop makes hipcc drain vmcnt);
// every wave copies its kStampSlots stamps to the buffer set with fpcc_conv_debug_stamps() when it ends.

// FOLD (with OG == 1; "folded" evaluation of summation order 3): ONE wave walks all offsets of its unit as in order 1, but at every
// boundary between two offset groups it adds the accumulator to a running sum t and restarts the accumulator from zero:
// t = p0, t = t + p1, t = t + p2, t = t + p3 with p_g = 0 for a group without a present offset -- bit for bit what the four waves of
// OG == 4 leave behind, without their LDS exchange, barrier and wait for the slowest group.  For maps with many row blocks, where
// the fourfold parallelism of OG == 4 buys nothing (272 K rows: 100 against 95 TFLOP/s, profiles/r03/grouped_fold.md).
// ASTAGE (experiment, same bits): the A fragments of the NEXT stage are requested all at once at the top of a stage into a second
// register set (the four 32-byte pieces of a gathered 128-byte line are then touched back to back instead of a quarter stage apart).
// PERSIST (round 4; needs a row-major table, conv_common.h): a wave (OG == 1) / workgroup (OG == 4) walks units first, first + stride, ...
// of the launch order instead of one, and requests the NEXT unit's table rows and output rows while it computes the current one: on a
// loaded chip the prologue's dependent loads (row order -> table rows -> gather addresses) took a wave 20-70 K cycles and its stores
// 20-40 K, a fifth of its life in which it feeds no MFMA (profiles/r04/prologue_epilogue.md).  Same chains, same bits.
// WPW (OG == 1 only): waves per workgroup.  The waves of a workgroup share nothing but the launch, and a workgroup's slot on the CU is
// held until its SLOWEST wave has finished: with one wave per workgroup a finished wave's slot is refilled at once.
template <int NBW, int CH, int SB, int DBG = 0, int OG = 1, bool FOLD = false, bool ASTAGE = false, bool PERSIST = false, int WPW = 4>
__global__ __launch_bounds__(64 * WPW, ((FOLD || PERSIST) ? 3 : WaveCfg<NBW, CH>::MIN_WAVES)) void k_conv_wave(ConvArgs a, const float *__restrict__ wp,
                                                                                                           int nbt, unsigned n_units,
                                                                                                           unsigned *unit_counter = nullptr) {
    constexpr int G8 = CH / 8;
    static_assert(OG == 1 || WPW == 4, "the four waves of a grouped workgroup are its four offset groups");
    __shared__ int32_t s_nbr_all[WPW][32 * 32];    // [offset (padded to 32)][row] per wave
    __shared__ float s_part[OG == 4 ? 4 * 16 * NBW * 64 : 1];
    __shared__ unsigned long long s_stamp[(DBG & 16) ? 4 * kStampSlots : 1];
    __shared__ unsigned s_next[(PERSIST && OG == 4) ? 4 : 1];   // OG == 4: {unit, pass tag} x 2, written by wave 0 for its workgroup
#define FPCC_STAMP(i) do { if (DBG & 16) stamp_lds(&s_stamp[wv * kStampSlots + (i)]); } while (0)

    // the wave index is wave-uniform, but hipcc only knows that when told: everything derived from it (the wave's offset group, its
    // offset mask, the stage iterator) otherwise lives in VGPRs, and the loop control below becomes ~25 vector instructions and
    // three exec-mask branches per stage instead of scalar code beside the MFMAs
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    if (DBG & 16) {
        for (int i = lane; i < kStampSlots; i += 64) s_stamp[wv * kStampSlots + i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    FPCC_STAMP(0);
    const unsigned n_cg = (unsigned)(nbt / NBW);
    // unit = (32-row block, column group); the column groups of one row block are adjacent units (same workgroup: their A
    // rows hit the CU's vector L1).  Natural order: contiguous unit ranges per XCD; with a row order: dispatch order.
    const unsigned blk = (PERSIST || a.row_order) ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const unsigned slot = OG == 4 ? blk : blk * (unsigned)WPW + (unsigned)wv;
    const unsigned n_slots = OG == 4 ? gridDim.x : gridDim.x * (unsigned)WPW;
    // PERSIST: the first unit of a slot is its own index; every further one is drawn from a counter (*unit_counter starts at the number
    // of slots) WHILE the current unit is computed -- the units come heaviest first (fpcc_conv_tile_keys) and take different times, so
    // a fixed stride or a serpentine would leave the slots up to 10 % apart at the end (measured: slower than one unit per workgroup)
    unsigned pass = 0, unit_next = 0xffffffffu;
    unsigned unit = slot;
    if (unit >= n_units) return;                    // OG == 1: no barrier below, a wave may leave on its own (OG == 4: whole workgroups)
    if (PERSIST && OG == 4) {
        if (threadIdx.x < 4) s_next[threadIdx.x] = 0xffffffffu;
        __syncthreads();
    }
    const int g = blockIdx.y;
    const int c_in = a.c1 + a.c2;
    const int n_chunks = c_in / CH;
    int32_t *s_nbr = s_nbr_all[wv];
    const int k_lo = OG == 4 ? offset_group_begin(wv, a.n_off) : 0, k_hi = OG == 4 ? offset_group_begin(wv + 1, a.n_off) : a.n_off;
    const bool row_major = table_is_row_major(a);   // always so when PERSIST
    // Row-major table: lane (i, h) fetches entries [16 h, 16 h + 16) of its row as four 16-byte pieces of the row's line (pieces past the
    // row's last one re-read it; entries past n_off are discarded) and the output row of position i; no load depends on another.
    auto fetch_unit = [&](unsigned u, i32x4 (&q)[4], int32_t &orow) {
        const int64_t p = (int64_t)(u / n_cg) * 32 + li;
        const int64_t pc = p < a.n_out ? p : a.n_out - 1;                        // past the end: re-read the last row, marked absent below
        const int32_t o = a.row_order ? a.row_order[pc] : (int32_t)pc;
        orow = p < a.n_out ? o : -1;
        const i32x4 *rowp = reinterpret_cast<const i32x4 *>(a.nbr + (a.row_order ? pc : (int64_t)o) * a.nbr_os);   // by position beside a row order
        const int last_piece = (a.n_off - 1) >> 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = rowp[min(4 * lh + j, last_piece)];
    };
    i32x4 q_next[4];
    int32_t row_next = -1;
    if (PERSIST) fetch_unit(unit, q_next, row_next);

  for (;;) {                                        // one pass unless PERSIST
    const unsigned rb_ = unit / n_cg, cg = unit - rb_ * n_cg;
    const int64_t row0 = (int64_t)rb_ * 32;

    int32_t my_row = -1;
    // neighbour rows of my 32 output rows -> this wave's LDS slice
    unsigned wmask = 0;
    if (row_major) {
        i32x4 q[4];
        if (PERSIST) {                                                       // requested after the previous unit's last stage
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = q_next[j];
            my_row = row_next;
            // draw the next unit now: the answer is back long before this unit's last stage
            if (OG == 4) {
                if (wv == 0) {
                    unsigned n = 0;
                    if (lane == 0) n = __hip_atomic_fetch_add(unit_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    n = __builtin_amdgcn_readfirstlane(n);
                    if (lane == 0) {
                        volatile unsigned *sn = s_next;
                        sn[2 * (pass & 1u)] = n;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        sn[2 * (pass & 1u) + 1] = pass;                      // the tag after the value
                    }
                }
            } else {
                unsigned n = 0;
                if (lane == 0) n = __hip_atomic_fetch_add(unit_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unit_next = __builtin_amdgcn_readfirstlane(n);
            }
        } else {
            fetch_unit(unit, q, my_row);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * lh + 4 * j + e;                           // this lane's offset; k0 = the offset of lane half 0
                const int k0 = 4 * j + e;
                const int32_t v = (k < a.n_off && my_row >= 0) ? q[j][e] : -1;
                s_nbr[k * 32 + li] = v;                                      // slots 27 .. 31 exist and are never read
                const unsigned long long b = __ballot(v >= 0);
                if (b & 0xffffffffull) wmask |= 1u << k0;
                if (b >> 32) wmask |= 1u << (k0 + 16);
            }
        if (OG == 4) wmask &= (k_hi >= 32 ? ~0u : (1u << k_hi) - 1u) & ~((1u << k_lo) - 1u);
    } else {
        if (row0 + li < a.n_out) my_row = a.row_order ? a.row_order[row0 + li] : (int32_t)(row0 + li);
        for (int k0 = k_lo; k0 < k_hi; k0 += 2) {                            // lane half h takes the offsets of parity h
            const int k = k0 + lh;
            int32_t v = -1;
            if (k < k_hi && my_row >= 0) v = a.nbr ? a.nbr[(int64_t)k * a.nbr_ks + (int64_t)my_row * a.nbr_os] : my_row;
            if (k < k_hi) s_nbr[k * 32 + li] = v;
            const unsigned long long b = __ballot(v >= 0);
            if (b & 0xffffffffull) wmask |= 1u << k0;
            if (b >> 32) wmask |= 1u << (k0 + 1);
        }
    }
    __builtin_amdgcn_wave_barrier();               // LDS operations of one wave execute in order; keep the compiler from reordering
    wmask = __builtin_amdgcn_readfirstlane(wmask);
    FPCC_STAMP(1);

    f32x16 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

    // FOLD: running sum over the offset groups finished so far (n_folded of them), see the template's comment
    f32x16 tsum[FOLD ? NBW : 1];
    int n_folded = 0, cur_g = 0;
    auto fold_acc = [&]() {
#pragma unroll
        for (int nb = 0; nb < (FOLD ? NBW : 0); ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tsum[nb][r] = n_folded ? tsum[nb][r] + acc[nb][r] : acc[nb][r];
                acc[nb][r] = 0.0f;
            }
        ++n_folded;
    };
    auto fold_zero = [&]() {                              // a group none of whose offsets is present: its partial sum is +0
#pragma unroll
        for (int nb = 0; nb < (FOLD ? NBW : 0); ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tsum[nb][r] = n_folded ? tsum[nb][r] + 0.0f : 0.0f;
        ++n_folded;
    };
    unsigned rest_c = wmask;
    int cc_c = 0;
    if (FOLD && wmask) {
        cur_g = offset_group_of(__ffs(wmask) - 1, a.n_off);
        for (int gz = 0; gz < cur_g; ++gz) fold_zero();
    }

    const int n_stages = __popc(wmask) * n_chunks;
    if (n_stages > 0) {
        // packed weights of (group g, offset k, chunk cc): G8 * nbt KB-blocks; this wave reads blocks [cg*NBW, cg*NBW + NBW) of each g8
        const int64_t chunk_floats = (int64_t)G8 * nbt * 256;
        const float *wp_g = wp + (int64_t)g * a.n_off * n_chunks * chunk_floats + ((int64_t)cg * NBW) * 256 + lane * 4;
        // Branch-free on purpose: divergent control flow inside the stage loop makes hipcc drain vmcnt(0) at the join, which
        // would wait for the loads issued a moment ago.  Loop invariants are taken out by hand (the compiler re-reads kernel
        // arguments and the zero row's address through scalar loads -- and waits for them -- inside the loop otherwise).
        const float *const zero = (const float *)g_zero_row + 4 * lh;
        const float *const x1b = a.x1 + 4 * lh, *const x2b = a.x2 ? a.x2 + 4 * lh : zero;
        const int64_t ld1 = a.ld1, ld2 = a.ld2;
        const int n1 = a.c1 / CH;                                // chunks [0, n1) lie in x1, the rest in x2
        // Per kernel offset: this lane's row in x1 / x2 (one LDS read, two 64-bit multiply-adds), then one add per chunk.
        // An absent neighbour reads the zero row with a zero chunk step.
        const float *a1, *a2, *bpk;
        int step;
        auto set_offset = [&](int k) {
            int32_t idx = s_nbr[k * 32 + li];
            if (DBG & 1) idx = idx < 0 ? idx : 0;
            const int32_t neg = idx >> 31;                        // 0 or ~0; a select here comes back as a branch (if-conversion)
            const uint64_t m = (uint64_t)(int64_t)neg, z = reinterpret_cast<uint64_t>(zero) & m;
            const int64_t row = idx & ~neg;
            a1 = reinterpret_cast<const float *>((reinterpret_cast<uint64_t>(x1b + row * ld1) & ~m) | z);
            a2 = reinterpret_cast<const float *>((reinterpret_cast<uint64_t>(x2b + row * ld2) & ~m) | z);
            step = CH & ~neg;
            bpk = (DBG & 2) ? wp_g : wp_g + (int64_t)k * n_chunks * chunk_floats;
        };
        // fetch position: the next (offset, chunk) stage whose operands are to be requested; past the last stage it stays
        // on the last one (re-read, never used)
        unsigned rest = wmask;
        int cc_f = 0;
        set_offset(__ffs(rest) - 1);
        const float *ap, *bp;
        auto next_stage = [&]() {
            const bool in1 = cc_f < n1;                           // wave-uniform
            ap = (in1 ? a1 : a2) + (in1 ? cc_f : cc_f - n1) * step;
            bp = (DBG & 2) ? bpk : bpk + cc_f * chunk_floats;
            if (cc_f + 1 < n_chunks) {
                ++cc_f;
            } else {
                const unsigned r2 = rest & (rest - 1);
                if (r2) { rest = r2; cc_f = 0; set_offset(__ffs(r2) - 1); }
            }
        };
        // hipcc's scheduler would sink every load down to its first use (it minimises live ranges), exposing a full L2 / HBM
        // latency per stage; sched_barrier pins the issue points.  Mask SB lets the scalar / vector address arithmetic of the
        // next stage float between the MFMAs (0: nothing crosses).
        f32x4 ra[G8], rb[G8][NBW];
        // The first stage's operands, issued in the SAME order as the refills inside the loop (group by group, A then B): the
        // wait counts at the loop head are merged over both ways into it, and a different order here (the scheduler reverses
        // it if left alone) turns them into vmcnt(0) on every iteration.
        next_stage();
        f32x4 ran[ASTAGE ? G8 : 1];
        if (ASTAGE) {
#pragma unroll
            for (int g8 = 0; g8 < G8; ++g8) {
                __builtin_amdgcn_sched_barrier(0);
                ra[g8] = *reinterpret_cast<const f32x4 *>(ap + 8 * g8);
            }
        }
#pragma unroll
        for (int g8 = 0; g8 < G8; ++g8) {
            __builtin_amdgcn_sched_barrier(0);
            if (!ASTAGE) ra[g8] = *reinterpret_cast<const f32x4 *>(ap + 8 * g8);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                __builtin_amdgcn_sched_barrier(0);
                rb[g8][nb] = *reinterpret_cast<const f32x4 *>(bp + ((int64_t)g8 * nbt + nb) * 256);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        FPCC_STAMP(2);
        for (int s = 0; s < n_stages; ++s) {
            FPCC_STAMP(3 + (s < 36 ? s : 36));
            if (FOLD) {
                // compute position (one stage behind the fetch position): entering the first chunk of an offset of a later group
                if (cc_c == 0) {
                    const int gk = offset_group_of(__ffs(rest_c) - 1, a.n_off);
                    if (gk != cur_g) {
                        fold_acc();
                        for (int gz = cur_g + 1; gz < gk; ++gz) fold_zero();
                        cur_g = gk;
                    }
                }
                if (++cc_c == n_chunks) { cc_c = 0; rest_c &= rest_c - 1; }
            }
            next_stage();                                       // stage s + 1
            __builtin_amdgcn_sched_barrier(SB);
            if (ASTAGE) {
#pragma unroll
                for (int g8 = 0; g8 < G8; ++g8) {
                    ran[g8] = *reinterpret_cast<const f32x4 *>(ap + 8 * g8);
                    __builtin_amdgcn_sched_barrier(SB);
                }
            }
#pragma unroll
            for (int g8 = 0; g8 < G8; ++g8) {
                const f32x4 av = ra[g8];
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, rb[g8][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, rb[g8][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, rb[g8][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, rb[g8][nb].w, acc[nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(SB);
                // the registers of this group are free now: refill them for the next stage
                if (!ASTAGE) ra[g8] = *reinterpret_cast<const f32x4 *>(ap + 8 * g8);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    __builtin_amdgcn_sched_barrier(SB);
                    rb[g8][nb] = *reinterpret_cast<const f32x4 *>(bp + ((int64_t)g8 * nbt + nb) * 256);
                }
                __builtin_amdgcn_sched_barrier(SB);
            }
            if (ASTAGE) {
#pragma unroll
                for (int g8 = 0; g8 < G8; ++g8) ra[g8] = ran[g8];
                __builtin_amdgcn_sched_barrier(SB);
            }
        }
    }

    FPCC_STAMP(40);
    if (PERSIST) {
        // the next unit's table rows and output rows travel while this unit's sums are finished and stored (requested here and not at the
        // unit's start: sixteen more registers through the stage loop would spill)
        if (OG == 4) {                                                       // wave 0 drew it at this unit's start
            volatile unsigned *sn = s_next;
            while (sn[2 * (pass & 1u) + 1] != pass) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            unit_next = __builtin_amdgcn_readfirstlane(sn[2 * (pass & 1u)]);
        }
        fetch_unit(unit_next < n_units ? unit_next : unit, q_next, row_next);
    }
    if (FOLD) {
        if (wmask) { fold_acc(); ++cur_g; }
        for (int gz = cur_g; gz < 4; ++gz) fold_zero();
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[nb] = tsum[nb];
    }
    // output rows of my accumulator registers: register r holds row (r & 3) + 8 (r >> 2) + 4 h of the block
    const float slope = (a.act == FPCC_ACT_PRELU && a.slope) ? a.slope[0] : 0.0f;
    if (OG == 4) {
        // partial sums of the four offset groups -> LDS; wave q then finishes accumulator registers [4q, 4q + 4) of every column
        // block: ((g0 + g1) + g2) + g3 in that order, whatever the groups held
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) s_part[((wv * NBW + nb) * 16 + reg) * 64 + lane] = acc[nb][reg];
        // bias and output-map entries before the barrier and the stores (see k_conv_mfma's epilogue: no load between two stores)
        float bias4[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) bias4[nb] = a.bias ? a.bias[32 * ((int)cg * NBW + nb) + li] : 0.0f;
        int64_t dst4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int reg = 4 * wv + q;
            const int64_t o = __shfl(my_row, (reg & 3) + 8 * (reg >> 2) + 4 * lh);
            dst4[q] = o < 0 ? -1 : a.out_map ? (int64_t)a.out_map[o * a.om_os + g * a.om_gs] : o * a.groups + g;
        }
        __syncthreads();
        FPCC_STAMP(41);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int reg = 4 * wv + q;
            const int64_t dst = dst4[q];
            if (dst < 0) continue;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const float *p = s_part + (nb * 16 + reg) * 64 + lane;
                constexpr int kStride = NBW * 16 * 64;
                const float v = ((p[0] + p[kStride]) + p[2 * kStride]) + p[3 * kStride];
                a.out[dst * a.ldo + 32 * ((int)cg * NBW + nb) + li] = finish(v, bias4[nb], a.act, slope, a.clip);
            }
        }
        if (DBG & 16) {
            FPCC_STAMP(42);
            if (lane == 0) {
                s_stamp[wv * kStampSlots + 44] = (unsigned long long)n_stages;
                s_stamp[wv * kStampSlots + 45] = ((unsigned long long)blockIdx.x << 8) | (unsigned)wv;
                s_stamp[wv * kStampSlots + 46] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
                s_stamp[wv * kStampSlots + 47] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
            }
            __builtin_amdgcn_wave_barrier();
            const long long w_id = (long long)blockIdx.x * 4 + wv;
            if (g_stamp_buf && (w_id + 1) * kStampSlots <= g_stamp_cap && lane < kStampSlots)
                g_stamp_buf[w_id * kStampSlots + lane] = s_stamp[wv * kStampSlots + lane];
        }
        if (!PERSIST) return;
        unit = unit_next;
        ++pass;
        if (unit >= n_units) return;                // the same decision in all four waves
        __syncthreads();                            // the partial sums are read: the next unit may overwrite them
        continue;
    }
    float bias1[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) bias1[nb] = a.bias ? a.bias[32 * ((int)cg * NBW + nb) + li] : 0.0f;
    int64_t dsts[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {            // lane rr (< 32) holds that row's output index; all lanes active here
        const int64_t o = __shfl(my_row, (reg & 3) + 8 * (reg >> 2) + 4 * lh);
        dsts[reg] = o < 0 ? -1 : a.out_map ? (int64_t)a.out_map[o * a.om_os + g * a.om_gs] : o * a.groups + g;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {            // no load from here on (see k_conv_mfma's epilogue)
        const int64_t dst = dsts[reg];
        if (dst < 0) continue;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
            a.out[dst * a.ldo + 32 * ((int)cg * NBW + nb) + li] = finish(acc[nb][reg], bias1[nb], a.act, slope, a.clip);
    }
    if (!PERSIST) return;
    unit = unit_next;
    ++pass;
    if (unit >= n_units) return;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 64 x 64 wave tile ("2 x 2"): the wave-autonomous kernel with TWO 32-row blocks and TWO 32-column blocks per wave.  Every A
// fragment (16 bytes per lane, gathered) and every B fragment (16 bytes per lane, packed weights) feeds two MFMAs per k-step
// instead of one resp. two: 4 vector-memory instructions per 16 MFMAs and 256 bytes of operands per MFMA, against 3 per 8 and
// 384 bytes in the 32 x 64 unit -- the operand stream out of L1 / L2, not the matrix pipe, is what bounds the 32 x 64 unit on the
// large maps (profiles/r03/wave22_sweeps.md).  Four independent accumulators per wave.  The price: a 64-row unit executes every
// kernel offset that any of its 64 rows has (in neighbour-pattern row order adjacent blocks have like patterns), and a
// launch has half as many units -- so this tile is for maps with many row blocks.  Same FMA chain per output element (order 1).
template <int SB>
__global__ __launch_bounds__(256, 3) void k_conv_wave22(ConvArgs a, const float *__restrict__ wp, int nbt, unsigned n_units) {
    constexpr int CH = 32, G8 = 4;
    __shared__ int32_t s_nbr_all[4][kMaxOffsets * 64];

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned n_cg = (unsigned)(nbt / 2);
    const unsigned blk = a.row_order ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const unsigned unit = blk * 4u + (unsigned)wv;
    if (unit >= n_units) return;                    // no barrier below: a wave may leave on its own
    const unsigned rb_ = unit / n_cg, cg = unit - rb_ * n_cg;
    const int g = blockIdx.y;
    const int64_t row0 = (int64_t)rb_ * 64;
    const int c_in = a.c1 + a.c2;
    const int n_chunks = c_in / CH;
    int32_t *s_nbr = s_nbr_all[wv];

    // lane = position `lane` of the unit's 64 rows here (one neighbour-table entry per lane and offset); in the MFMA phase lane
    // (i, h) serves rows i (first block) and 32 + i (second block)
    int32_t my_row = -1;
    if (row0 + lane < a.n_out) my_row = a.row_order ? a.row_order[row0 + lane] : (int32_t)(row0 + lane);
    // (a row-major table beside a row order holds its rows in position order: conv_common.h)
    const int64_t table_row = (a.row_order && table_is_row_major(a)) ? row0 + lane : (int64_t)my_row;
    unsigned wmask = 0;
    for (int k = 0; k < a.n_off; ++k) {
        int32_t v = -1;
        if (my_row >= 0) v = a.nbr ? a.nbr[(int64_t)k * a.nbr_ks + table_row * a.nbr_os] : my_row;
        s_nbr[k * 64 + lane] = v;
        if (__ballot(v >= 0) != 0ull) wmask |= 1u << k;
    }
    __builtin_amdgcn_wave_barrier();

    f32x16 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.0f;

    const int n_stages = __popc(wmask) * n_chunks;
    if (n_stages > 0) {
        const int64_t chunk_floats = (int64_t)G8 * nbt * 256;
        const float *wp_g = wp + (int64_t)g * a.n_off * n_chunks * chunk_floats + ((int64_t)cg * 2) * 256 + lane * 4;
        const float *const zero = (const float *)g_zero_row + 4 * lh;
        const float *const x1b = a.x1 + 4 * lh, *const x2b = a.x2 ? a.x2 + 4 * lh : zero;
        const int64_t ld1 = a.ld1, ld2 = a.ld2;
        const int n1 = a.c1 / CH;
        const float *a1[2], *a2[2], *bpk;
        int step[2];
        auto set_offset = [&](int k) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int32_t idx = s_nbr[k * 64 + 32 * rb + li];
                const int32_t neg = idx >> 31;
                const uint64_t m = (uint64_t)(int64_t)neg, z = reinterpret_cast<uint64_t>(zero) & m;
                const int64_t row = idx & ~neg;
                a1[rb] = reinterpret_cast<const float *>((reinterpret_cast<uint64_t>(x1b + row * ld1) & ~m) | z);
                a2[rb] = reinterpret_cast<const float *>((reinterpret_cast<uint64_t>(x2b + row * ld2) & ~m) | z);
                step[rb] = CH & ~neg;
            }
            bpk = wp_g + (int64_t)k * n_chunks * chunk_floats;
        };
        unsigned rest = wmask;
        int cc_f = 0;
        set_offset(__ffs(rest) - 1);
        const float *ap[2], *bp;
        auto next_stage = [&]() {
            const bool in1 = cc_f < n1;                           // wave-uniform
            const int c = in1 ? cc_f : cc_f - n1;
            ap[0] = (in1 ? a1[0] : a2[0]) + c * step[0];
            ap[1] = (in1 ? a1[1] : a2[1]) + c * step[1];
            bp = bpk + cc_f * chunk_floats;
            if (cc_f + 1 < n_chunks) {
                ++cc_f;
            } else {
                const unsigned r2 = rest & (rest - 1);
                if (r2) { rest = r2; cc_f = 0; set_offset(__ffs(r2) - 1); }
            }
        };
        f32x4 ra[2][G8], rbv[G8][2];
        next_stage();
#pragma unroll
        for (int g8 = 0; g8 < G8; ++g8) {
            __builtin_amdgcn_sched_barrier(0);
            ra[0][g8] = *reinterpret_cast<const f32x4 *>(ap[0] + 8 * g8);
            __builtin_amdgcn_sched_barrier(0);
            ra[1][g8] = *reinterpret_cast<const f32x4 *>(ap[1] + 8 * g8);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                __builtin_amdgcn_sched_barrier(0);
                rbv[g8][nb] = *reinterpret_cast<const f32x4 *>(bp + ((int64_t)g8 * nbt + nb) * 256);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int s = 0; s < n_stages; ++s) {
            next_stage();                                       // stage s + 1
            __builtin_amdgcn_sched_barrier(SB);
#pragma unroll
            for (int g8 = 0; g8 < G8; ++g8) {
                const f32x4 av0 = ra[0][g8], av1 = ra[1][g8];
                const f32x4 b0 = rbv[g8][0], b1 = rbv[g8][1];
#define FPCC_STEP(c)                                                                                     \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.c, b0.c, acc[0][0], 0, 0, 0);       \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.c, b1.c, acc[0][1], 0, 0, 0);       \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.c, b0.c, acc[1][0], 0, 0, 0);       \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.c, b1.c, acc[1][1], 0, 0, 0);
                FPCC_STEP(x) FPCC_STEP(y) FPCC_STEP(z) FPCC_STEP(w)
#undef FPCC_STEP
                __builtin_amdgcn_sched_barrier(SB);
                ra[0][g8] = *reinterpret_cast<const f32x4 *>(ap[0] + 8 * g8);
                __builtin_amdgcn_sched_barrier(SB);
                ra[1][g8] = *reinterpret_cast<const f32x4 *>(ap[1] + 8 * g8);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    __builtin_amdgcn_sched_barrier(SB);
                    rbv[g8][nb] = *reinterpret_cast<const f32x4 *>(bp + ((int64_t)g8 * nbt + nb) * 256);
                }
                __builtin_amdgcn_sched_barrier(SB);
            }
        }
    }

    const float slope = (a.act == FPCC_ACT_PRELU && a.slope) ? a.slope[0] : 0.0f;
    float bias2[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bias2[nb] = a.bias ? a.bias[32 * ((int)cg * 2 + nb) + li] : 0.0f;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        int64_t dsts[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t o = __shfl(my_row, 32 * rb + (reg & 3) + 8 * (reg >> 2) + 4 * lh);
            dsts[reg] = o < 0 ? -1 : a.out_map ? (int64_t)a.out_map[o * a.om_os + g * a.om_gs] : o * a.groups + g;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t dst = dsts[reg];
            if (dst < 0) continue;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                a.out[dst * a.ldo + 32 * ((int)cg * 2 + nb) + li] = finish(acc[rb][nb][reg], bias2[nb], a.act, slope, a.clip);
        }
    }
}

__global__ __launch_bounds__(256) void k_pack_weights(const float *__restrict__ w, int64_t n_mats, int c_in, int c_out,
                                                      float *__restrict__ wp) {
    // one thread per packed element: [m][cc][g8][nb][h][i][j] <- w[m][32 cc + 8 g8 + 4 h + j][32 nb + i]
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_mat = (int64_t)c_in * c_out;
    if (e >= n_mats * per_mat) return;
    const int nbt = c_out / 32;
    const int64_t m = e / per_mat;
    int64_t r = e - m * per_mat;
    const int j = (int)(r & 3); r >>= 2;
    const int i = (int)(r & 31); r >>= 5;
    const int h = (int)(r & 1); r >>= 1;
    const int nb = (int)(r % nbt); r /= nbt;
    const int g8 = (int)(r & 3); r >>= 2;
    const int cc = (int)r;
    wp[e] = w[m * per_mat + (int64_t)(32 * cc + 8 * g8 + 4 * h + j) * c_out + 32 * nb + i];
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

int knob(int k);

template <int NBT, int CH, int WM, int WN>
int launch_mfma_cfg(ConvArgs a, hipStream_t s) {
    constexpr int TM = 32 * WM;
    const unsigned tiles = (unsigned)((a.n_out + TM - 1) / TM);
    const dim3 grid(tiles, a.groups), block(64 * WM * WN);
    const int dbg = (NBT == 4 && CH == 32) ? knob(3) : 0;          // kKnobWaveDbg: experiments on the 128-column shapes only
    if (dbg == 1) hipLaunchKernelGGL((k_conv_mfma<NBT, CH, WM, WN, 1>), grid, block, 0, s, a);
    else if (dbg == 2) hipLaunchKernelGGL((k_conv_mfma<NBT, CH, WM, WN, 2>), grid, block, 0, s, a);
    else if (dbg == 3) hipLaunchKernelGGL((k_conv_mfma<NBT, CH, WM, WN, 3>), grid, block, 0, s, a);
    else if (dbg == 4) hipLaunchKernelGGL((k_conv_mfma<NBT, CH, WM, WN, 4>), grid, block, 0, s, a);
    else if (dbg == 7) hipLaunchKernelGGL((k_conv_mfma<NBT, CH, WM, WN, 7>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_conv_mfma<NBT, CH, WM, WN>), grid, block, 0, s, a);
    return check_hip(hipGetLastError(), "k_conv_mfma");
}

// Tuning knobs (fpcc_conv_set_tuning; initial values from the environment).  None of them changes a result EXCEPT
// kKnobGroupedOff (experiments only: 1 = multi-offset layers in order 1 on the plain wave kernel instead of grouped / order 3),
// which is therefore refused unless the process runs with FPCC_EXPERIMENT=1 and has no environment variable.
enum { kKnobWaveOn = 0, kKnobWaveNbw = 1, kKnobWaveSb = 2, kKnobWaveDbg = 3, kKnobGroupedFoldRows = 4, kKnobMfmaCfg = 5, kKnobPointwiseRows = 6,
       kKnobGroupedOff = 7, kKnobGroupedNbw = 8, kKnobWave22Rows = 9, kKnobLdsRows = 10, kKnobLdsRowBlocks = 11, kKnobPersist = 12, kKnobCount = 13 };
int g_knob[kKnobCount] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
int knob(int k) {
    if (g_knob[k] < 0) {
        static const char *names[kKnobCount] = {"FPCC_CONV_WAVE", "FPCC_WAVE_NBW", "FPCC_WAVE_SB", "FPCC_WAVE_DBG", "FPCC_GROUPED_FOLD_ROWS",
                                                "FPCC_MFMA_TILE", "FPCC_POINTWISE_MIN_ROWS", "", "FPCC_GROUPED_NBW",
                                                "FPCC_WAVE22_MIN_ROWS", "FPCC_LDS_MIN_ROWS", "FPCC_LDS_ROW_BLOCKS",
                                                "FPCC_CONV_PERSIST"};
        static const int defaults[kKnobCount] = {1, 0, 1, 0, 100 * 1024, 0, 32 * 1024, 0, 0, 0, 0, 2, 0};
        const char *e = k == kKnobGroupedOff ? nullptr : getenv(names[k]);
        g_knob[k] = e ? atoi(e) : defaults[k];
    }
    return g_knob[k];
}

// Tile height by map size, measured on MI355X (profiles/r01, FPCC_MFMA_CFG sweeps): 64-row tiles (2x2 waves) from 32 Ki
// rows up -- against 128-row tiles they halve the tail of the last wave of workgroups and execute fewer (tile, offset)
// stages -- 128-row tiles (4x1) only for C_out <= 64 on the largest maps, 32-row tiles (1xNBT) below 32 Ki rows.
// Knob FPCC_MFMA_TILE = 1|2|3 forces 128|64|32 rows (0: by size).
template <int NBT, int CH>
int launch_mfma(const ConvArgs &a, hipStream_t s) {
    const int forced = knob(kKnobMfmaCfg) - 1;
    const int64_t work = a.n_out * a.groups;
    constexpr int WNS = NBT >= 2 ? 2 : 1;       // 64-row tile: 2 x WNS waves
    const int cfg = forced >= 0 ? forced : (NBT <= 2 && work >= 128 * 1024) ? 0 : work >= 32 * 1024 ? 1 : 2;
    if (cfg == 0) return launch_mfma_cfg<NBT, CH, 4, 1>(a, s);
    if (cfg == 1) return launch_mfma_cfg<NBT, CH, 2, WNS>(a, s);
    return launch_mfma_cfg<NBT, CH, 1, NBT>(a, s);
}

// per-point layer with ONE input channel (the first MLP layer on a residual / logit column): an outer product, written with
// consecutive threads on consecutive output columns (the row-per-thread kernel above stores with a row-stride between lanes)
__global__ __launch_bounds__(256) void k_conv_c1_pointwise(ConvArgs a) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n_out * a.c_out) return;
    const int64_t o = e / a.c_out;
    const int j = (int)(e - o * a.c_out);
    const float slope = (a.act == FPCC_ACT_PRELU && a.slope) ? a.slope[0] : 0.0f;
    a.out[o * a.ldo + j] = finish(fmaf(a.x1[o * a.ld1], a.w[j], 0.0f), a.bias ? a.bias[j] : 0.0f, a.act, slope, a.clip);
}

template <int NBW>
int launch_wave_cfg(const ConvArgs &a, const float *wp, int nbt, hipStream_t s) {
    const int64_t row_blocks = (a.n_out + 31) / 32;
    const int64_t units = row_blocks * (nbt / NBW);
    if (units > 0x7fffffffll) return fail_arg("conv_f32: too many work units");
    const dim3 grid((unsigned)((units + 3) / 4), a.groups);
    const int dbg = knob(kKnobWaveDbg);
    // knob FPCC_WAVE_SB=1: the address arithmetic of the next stage may be scheduled between the MFMAs
    if (dbg == 1) hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6, 1>), grid, dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    else if (dbg == 2) hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6, 2>), grid, dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    else if (dbg == 3) hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6, 3>), grid, dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    else if (knob(kKnobWaveSb)) hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6>), grid, dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    else hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0>), grid, dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    return check_hip(hipGetLastError(), "k_conv_wave");
}

// cached per device, initialised from any host thread (several frame threads launch concurrently)
constexpr int kMaxDevices = 16;
inline int current_device() {
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices) ? dev : -1;
}
int cu_count() {
    static std::atomic<int> cached[kMaxDevices];
    const int dev = current_device();
    if (dev < 0) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Persistent form (knob 12 / FPCC_CONV_PERSIST = workgroups per CU, 0 = off): when a launch has more workgroups than the chip holds
// at once and the table is row-major, that many workgroups walk the units with a stride and prefetch the next unit's table rows.
// Unit counters of the persistent launches: a ring of words in device memory; a launch takes the next one, sets it to its number of
// slots on its stream (a 4-byte fill ahead of the kernel) and its slots draw their further units from it.
constexpr int kUnitCounters = 256;
__device__ unsigned g_unit_counters[kUnitCounters];
unsigned *next_unit_counter(unsigned start, hipStream_t s, int *rc) {
    static std::atomic<unsigned> seq{0};           // launches come from several host threads
    static std::atomic<unsigned *> bases[kMaxDevices];   // the symbol has one address per device
    const int dev = current_device();
    if (dev < 0) {
        *rc = fail_arg("unit counters: no current device");
        return nullptr;
    }
    unsigned *base = bases[dev].load(std::memory_order_acquire);
    if (!base) {
        if (hipGetSymbolAddress(reinterpret_cast<void **>(&base), HIP_SYMBOL(g_unit_counters)) != hipSuccess) {
            *rc = check_hip(hipGetLastError(), "unit counters");
            return nullptr;
        }
        bases[dev].store(base, std::memory_order_release);
    }
    unsigned *c = base + (seq++ % kUnitCounters);
    *rc = check_hip(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c), (int)start, 1, s), "unit counter fill");
    return c;
}

inline int64_t persist_slots() {                 // knob values above 16 are an absolute number of workgroups (tests)
    const int v = knob(kKnobPersist);
    return v > 16 ? v : (int64_t)v * cu_count();
}
inline bool persist_ok(const ConvArgs &a) {
    return knob(kKnobPersist) > 0 && a.nbr && a.nbr_ks == 1 && (a.nbr_os & 3) == 0 && a.nbr_os >= ((a.n_off + 3) & ~3) &&
           (reinterpret_cast<uintptr_t>(a.nbr) & 15) == 0 && a.groups == 1;
}

// Grouped evaluation (summation order 3): one workgroup per (32-row block, column group), its four waves = the four offset groups.
template <int NBW>
int launch_grouped_cfg(const ConvArgs &a, const float *wp, int nbt, hipStream_t s) {
    const int64_t row_blocks = (a.n_out + 31) / 32;
    const int64_t units = row_blocks * (nbt / NBW);
    if (units > 0x7fffffffll) return fail_arg("conv_f32: too many work units");
    const int dbg = knob(kKnobWaveDbg);
    const int64_t slots = persist_slots();
    if (dbg == 0 && persist_ok(a) && units > slots) {
        int rc = FPCC_OK;
        unsigned *counter = next_unit_counter((unsigned)slots, s, &rc);
        if (rc != FPCC_OK) return rc;
        hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6, 0, 4, false, false, true>), dim3((unsigned)slots, 1), dim3(256), 0, s, a, wp, nbt, (unsigned)units,
                           counter);
        return check_hip(hipGetLastError(), "k_conv_wave(grouped, persistent)");
    }
    if (dbg == 16)                      // stage stamps (fpcc_conv_debug_stamps)
        hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6, 16, 4>), dim3((unsigned)units, a.groups), dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    else if (NBW == 1 && dbg == 17)     // + no gather traffic / weights from one chunk / both (results wrong)
        hipLaunchKernelGGL((k_conv_wave<1, 32, 0x6, 17, 4>), dim3((unsigned)units, a.groups), dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    else if (NBW == 1 && dbg == 18)
        hipLaunchKernelGGL((k_conv_wave<1, 32, 0x6, 18, 4>), dim3((unsigned)units, a.groups), dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    else if (NBW == 1 && dbg == 19)
        hipLaunchKernelGGL((k_conv_wave<1, 32, 0x6, 19, 4>), dim3((unsigned)units, a.groups), dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    else
        hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6, 0, 4>), dim3((unsigned)units, a.groups), dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    return check_hip(hipGetLastError(), "k_conv_wave(grouped)");
}

// Folded evaluation of the same order (FOLD): one wave per unit, for maps with at least kKnobGroupedFoldRows rows.
template <int NBW>
int launch_folded_cfg(const ConvArgs &a, const float *wp, int nbt, hipStream_t s) {
    const int64_t units = ((a.n_out + 31) / 32) * (nbt / NBW);
    if (units > 0x7fffffffll) return fail_arg("conv_f32: too many work units");
    const int64_t slots = persist_slots();
    if (knob(kKnobWaveDbg) == 0 && persist_ok(a) && (units + 3) / 4 > slots) {
        int rc = FPCC_OK;
        unsigned *counter = next_unit_counter((unsigned)(4 * slots), s, &rc);
        if (rc != FPCC_OK) return rc;
        hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6, 0, 1, true, false, true>), dim3((unsigned)slots, 1), dim3(256), 0, s, a, wp, nbt, (unsigned)units,
                           counter);
        return check_hip(hipGetLastError(), "k_conv_wave(folded, persistent)");
    }
    if (knob(kKnobWaveDbg) == 64) {                  // experiment: one wave per workgroup
        hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6, 0, 1, true, false, false, 1>), dim3((unsigned)units, a.groups), dim3(64), 0, s, a, wp, nbt,
                           (unsigned)units);
        return check_hip(hipGetLastError(), "k_conv_wave(folded, one wave per workgroup)");
    }
    if (NBW == 2 && knob(kKnobWaveDbg) == 32)        // experiment: A fragments a whole stage at a time
        hipLaunchKernelGGL((k_conv_wave<2, 32, 0x6, 0, 1, true, true>), dim3((unsigned)((units + 3) / 4), a.groups), dim3(256), 0, s, a, wp, nbt,
                           (unsigned)units);
    else
        hipLaunchKernelGGL((k_conv_wave<NBW, 32, 0x6, 0, 1, true>), dim3((unsigned)((units + 3) / 4), a.groups), dim3(256), 0, s, a, wp, nbt,
                           (unsigned)units);
    return check_hip(hipGetLastError(), "k_conv_wave(folded)");
}

int launch_grouped(const ConvArgs &a, const float *wp, hipStream_t s) {
    const int nbt = a.c_out / 32;
    // both operands through LDS (conv_lds.hip) on maps of at least FPCC_LDS_MIN_ROWS rows (knob 10; 0 = never); same order 3
    const int64_t lds_rows = knob(kKnobLdsRows);
    if (lds_rows > 0 && a.n_out >= lds_rows) {
        const int rc = launch_conv_lds(a, wp, knob(kKnobLdsRowBlocks), knob(kKnobWaveDbg) & 63, s);
        if (rc != -1) return rc;
    }
    const int64_t fold_rows = knob(kKnobGroupedFoldRows);
    if (fold_rows > 0 && a.n_out >= fold_rows && knob(kKnobGroupedNbw) <= 0)
        return nbt % 2 == 0 ? launch_folded_cfg<2>(a, wp, nbt, s) : launch_folded_cfg<1>(a, wp, nbt, s);
    int nbw = knob(kKnobGroupedNbw);
    if (nbw <= 0) nbw = a.n_out >= 40 * 1024 ? 2 : 1;        // measured: 64 columns per workgroup from ~40 K rows (profiles/r03/grouped_probe.md)
    while (nbw > nbt || nbt % nbw) nbw >>= 1;
    return nbw == 2 ? launch_grouped_cfg<2>(a, wp, nbt, s) : launch_grouped_cfg<1>(a, wp, nbt, s);
}

// Unit width (column blocks per wave) by map size, measured on MI355X (profiles/r02/wave_kernel_sweeps.md): two column blocks
// from 32 Ki rows up, one below -- on a 18 K-row map one-block units are 20 % faster (4x the units for the same chip), on
// 272 K rows two-block units gather every row half as often.  Four-block units (every row gathered once) are never the
// fastest: the B stream is 256 bytes per MFMA at any width, and three waves per SIMD hide less than five.
// Knob FPCC_WAVE_NBW = 1|2|4 forces.
int launch_wave22(const ConvArgs &a, const float *wp, int nbt, hipStream_t s) {
    const int64_t units = ((a.n_out + 63) / 64) * (nbt / 2);
    if (units > 0x7fffffffll) return fail_arg("conv_f32: too many work units");
    const dim3 grid((unsigned)((units + 3) / 4), a.groups);
    if (knob(kKnobWaveSb)) hipLaunchKernelGGL((k_conv_wave22<0x6>), grid, dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    else hipLaunchKernelGGL((k_conv_wave22<0>), grid, dim3(256), 0, s, a, wp, nbt, (unsigned)units);
    return check_hip(hipGetLastError(), "k_conv_wave22");
}

int launch_wave(const ConvArgs &a, const float *wp, hipStream_t s) {
    const int nbt = a.c_out / 32;
    const int64_t work = a.n_out * a.groups;
    // 64 x 64 wave tiles on multi-offset layers of maps with at least FPCC_WAVE22_MIN_ROWS rows (knob 9; 0 = never)
    const int64_t rows22 = knob(kKnobWave22Rows);
    if (rows22 > 0 && a.n_off > 1 && nbt % 2 == 0 && work >= rows22 && knob(kKnobWaveNbw) <= 0) return launch_wave22(a, wp, nbt, s);
    int nbw = knob(kKnobWaveNbw);
    if (nbw <= 0) nbw = work >= 32 * 1024 ? 2 : 1;
    while (nbw > nbt || nbt % nbw) nbw >>= 1;
    if (nbw >= 4) return launch_wave_cfg<4>(a, wp, nbt, s);
    if (nbw == 2) return launch_wave_cfg<2>(a, wp, nbt, s);
    return launch_wave_cfg<1>(a, wp, nbt, s);
}


// ---------------------------------------------------------------------------------------------------------------
// Per-point layers (ONE kernel offset, identity row map: MinkowskiLinear, 1x1x1 convolutions) on large maps.  In k_conv_wave
// such a layer is 4-8 stages per wave: the operand fetches of a 32-row unit are not amortised and the weights are streamed
// from L2 once per unit.  Here a wave keeps the B operands of its NBW column blocks for ALL input channels in registers
// (16 * NCH * NBW VGPRs) and walks many 32-row blocks: per block only the A rows are loaded (16-byte loads in MFMA operand
// layout, rows are contiguous in memory) and the outputs stored.  Two to three waves per SIMD overlap one wave's loads with
// another's MFMAs.  Same FMA chain per output element as k_conv_wave / k_conv_mfma (chunks ascending, groups of 8 channels
// ascending, 0,4,1,5,2,6,3,7 inside a group): summation order 1, results bit-identical.
template <int NBW, int NCH, int DBG = 0>
__global__ __launch_bounds__(256, 2) void k_pointwise_wave(ConvArgs a, const float *__restrict__ wp, int nbt, unsigned n_row_blocks,
                                                           unsigned waves_per_cg) {
    constexpr int AC = NCH < 4 ? NCH : 4;          // chunks of A held at a time (16 VGPRs each)
    constexpr int TW = 32 * NBW + 4;               // row pitch of the output tile in LDS (floats; +4: rows start on different banks)
    __shared__ float s_tile[4][32 * TW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned n_cg = (unsigned)(nbt / NBW);
    const unsigned w = blockIdx.x * 4u + (unsigned)wv;          // global wave: column group = w % n_cg, row blocks strided
    const unsigned cg = w % n_cg, slot = w / n_cg;
    if (slot >= waves_per_cg) return;

    // B operands of this wave's column blocks, all chunks: wp[0][cc][g8][nb][h][i][j]
    f32x4 rb[NCH][4][NBW];
    {
        const float *wpl = wp + ((int64_t)cg * NBW) * 256 + lane * 4;
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                    rb[cc][g8][nb] = *reinterpret_cast<const f32x4 *>(wpl + (((int64_t)cc * 4 + g8) * nbt + nb) * 256);
    }
    const int n1 = a.c1 / 32;                                   // chunks [0, n1) lie in x1, the rest in x2
    const float slope = (a.act == FPCC_ACT_PRELU && a.slope) ? a.slope[0] : 0.0f;
    float bias[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) bias[nb] = a.bias ? a.bias[32 * ((int)cg * NBW + nb) + li] : 0.0f;

    // Operand stream: stage t = (row block of this wave, chunk); the AC chunk slots of `ra` form a ring -- right after the MFMAs
    // of a group of 8 channels have consumed ra[slot][g8] the same registers are reloaded with stage t + AC (the same chunk of
    // the wave's NEXT row block when all chunks are resident), so every load has a whole row block of MFMAs to land.
    auto row_ptrs = [&](unsigned rbk, const float *&q1, const float *&q2) {
        int64_t row = (int64_t)rbk * 32 + li;
        if (row >= a.n_out) row = a.n_out - 1;                  // tail block / past the end: re-read the last row, never stored
        if (DBG & 1) row = lane & 1;                            // timing experiment: (almost) no A traffic
        q1 = a.x1 + row * a.ld1 + 4 * lh;
        q2 = a.x2 ? a.x2 + row * a.ld2 + 4 * lh : q1;
    };
    auto fetch = [&](const float *q1, const float *q2, int cc, int g8) -> f32x4 {
        if (DBG & 8) { const float v = (float)(lane + g8 + cc); return f32x4{v, v + 1.0f, v + 2.0f, v + 3.0f}; }
        const float *src = cc < n1 ? q1 + 32 * cc : q2 + 32 * (cc - n1);             // wave-uniform
        return *reinterpret_cast<const f32x4 *>(src + 8 * g8);
    };
    const float *c1p, *c2p, *n1p, *n2p;
    row_ptrs(slot, c1p, c2p);
    f32x4 ra[AC][4];
#pragma unroll
    for (int c = 0; c < AC; ++c)
#pragma unroll
        for (int g8 = 0; g8 < 4; ++g8) {
            __builtin_amdgcn_sched_barrier(0);
            ra[c][g8] = fetch(c1p, c2p, c, g8);
        }
    __builtin_amdgcn_sched_barrier(0);
    for (unsigned rbk = slot; rbk < n_row_blocks; rbk += waves_per_cg) {
        const int64_t row0 = (int64_t)rbk * 32;
        row_ptrs(rbk + waves_per_cg, n1p, n2p);                 // (clamped past the end: loaded, never used)
        f32x16 acc[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
            constexpr int kNone = 0;
            (void)kNone;
            const int sl = cc % AC;
            const bool wraps = cc + AC >= NCH;                  // compile-time after unrolling
            const int cn = (cc + AC) % NCH;
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8) {
                const f32x4 av = ra[sl][g8];
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, rb[cc][g8][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, rb[cc][g8][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, rb[cc][g8][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, rb[cc][g8][nb].w, acc[nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0x6);
                ra[sl][g8] = wraps ? fetch(n1p, n2p, cn, g8) : fetch(c1p, c2p, cn, g8);       // stage t + AC
                __builtin_amdgcn_sched_barrier(0x6);
            }
        }
        c1p = n1p;
        c2p = n2p;
        // Epilogue through a private LDS tile: the accumulator layout (register r = row (r & 3) + 8 (r >> 2) + 4 h, lane = column)
        // would store one dword per lane, 256 bytes per instruction; transposed through LDS every lane stores 16 bytes and an
        // instruction writes whole 128-byte lines of four rows (a quarter of the store instructions).
        float *tile = s_tile[wv];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) tile[r * TW + 32 * nb + li] = finish(acc[nb][reg], bias[nb], a.act, slope, a.clip);
        }
        __builtin_amdgcn_wave_barrier();
        constexpr int LPR = 8 * NBW;                            // lanes per row (16 bytes each)
        constexpr int RPI = 64 / LPR;                           // rows per store instruction
        const int qr = lane / LPR, qc = lane % LPR;
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += RPI) {
            const int r = r0 + qr;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + r * TW + 4 * qc);
            const int64_t o = row0 + r;
            if (o < a.n_out && !((DBG & 4) && v.x != 12345.678f))
                *reinterpret_cast<f32x4 *>(a.out + o * a.ldo + 32 * (int)cg * NBW + 4 * qc) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// rows from which the persistent per-point kernel is used (below, a launch has too few 32-row blocks for every wave to reuse
// its weights several times and k_conv_wave's finer units fill the chip better): knob 6 / FPCC_POINTWISE_MIN_ROWS, 0 = never

template <int NBW, int NCH>
int launch_pointwise_cfg(const ConvArgs &a, const float *wp, int nbt, hipStream_t s) {
    const unsigned row_blocks = (unsigned)((a.n_out + 31) / 32);
    const unsigned n_cg = (unsigned)(nbt / NBW);
    unsigned waves_per_cg = (256u * 4u * 2u) / n_cg;            // two waves per SIMD on the whole chip
    if (waves_per_cg > row_blocks) waves_per_cg = row_blocks;
    const unsigned waves = waves_per_cg * n_cg;
    const int dbg = (NBW == 2 && NCH == 4) ? knob(kKnobWaveDbg) : 0;       // experiments on the 128 -> 128 shape only
    if (dbg == 1) hipLaunchKernelGGL((k_pointwise_wave<NBW, NCH, 1>), dim3((waves + 3) / 4), dim3(256), 0, s, a, wp, nbt, row_blocks, waves_per_cg);
    else if (dbg == 4) hipLaunchKernelGGL((k_pointwise_wave<NBW, NCH, 4>), dim3((waves + 3) / 4), dim3(256), 0, s, a, wp, nbt, row_blocks, waves_per_cg);
    else if (dbg == 5) hipLaunchKernelGGL((k_pointwise_wave<NBW, NCH, 5>), dim3((waves + 3) / 4), dim3(256), 0, s, a, wp, nbt, row_blocks, waves_per_cg);
    else if (dbg == 12) hipLaunchKernelGGL((k_pointwise_wave<NBW, NCH, 12>), dim3((waves + 3) / 4), dim3(256), 0, s, a, wp, nbt, row_blocks, waves_per_cg);
    else hipLaunchKernelGGL((k_pointwise_wave<NBW, NCH>), dim3((waves + 3) / 4), dim3(256), 0, s, a, wp, nbt, row_blocks, waves_per_cg);
    return check_hip(hipGetLastError(), "k_pointwise_wave");
}

// -1: shape not covered
int launch_pointwise(const ConvArgs &a, const float *wp, hipStream_t s) {
    const int nbt = a.c_out / 32, nch = (a.c1 + a.c2) / 32;
    switch (nch) {
        case 1: return nbt % 4 == 0 ? launch_pointwise_cfg<4, 1>(a, wp, nbt, s) : nbt % 2 == 0 ? launch_pointwise_cfg<2, 1>(a, wp, nbt, s)
                                                                                                : launch_pointwise_cfg<1, 1>(a, wp, nbt, s);
        case 2: return nbt % 4 == 0 ? launch_pointwise_cfg<4, 2>(a, wp, nbt, s) : nbt % 2 == 0 ? launch_pointwise_cfg<2, 2>(a, wp, nbt, s)
                                                                                                : launch_pointwise_cfg<1, 2>(a, wp, nbt, s);
        case 4: return nbt % 2 == 0 ? launch_pointwise_cfg<2, 4>(a, wp, nbt, s) : launch_pointwise_cfg<1, 4>(a, wp, nbt, s);
        case 8: return launch_pointwise_cfg<1, 8>(a, wp, nbt, s);
        default: return -1;
    }
}


// 3x3x3 convolution of a CONSTANT-ONE one-channel input (the codec's first layer: every voxel carries the feature 1): the sum
// over the neighbours that exist of w[k][j], in ascending offset order -- the same chain the general kernel evaluates with
// x = 1 (fmaf(1, w, acc) == acc + w), so the result is bit-identical -- read from the row's 27-bit presence mask instead of
// the 108-byte neighbour row.
// Round 6: ONE THREAD PER ROW, all CO columns: the row's mask bit k becomes the float 0 / 1 once (two integer instructions) and
// feeds CO / 2 packed FMAs whose other operand -- w[k][2p], w[k][2p + 1] -- is the same for every lane, i.e. lives in SGPRs streamed
// by scalar loads; an absent neighbour contributes fmaf(0, w, acc) == acc (acc is never -0: it starts at +0 and a sum that cancels
// is +0), so there is no divergent branch and the chain is the general kernel's.
// 4 bytes read and 4 CO bytes written per row: HBM-bound (rounds 2-5: one thread per (row, 4 columns), weights re-read from LDS for
// every row behind 27 exec-mask branches, 24 % of 8 TB/s).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int CO, int R>
__global__ __launch_bounds__(256) void k_conv_ones_k3(const uint32_t *__restrict__ masks, int64_t n, const float *__restrict__ w,
                                                      const float *__restrict__ bias, int act, const float *__restrict__ slope, float clip,
                                                      float *__restrict__ out, int ldo) {
    const int64_t base = (int64_t)blockIdx.x * (256 * R) + threadIdx.x;
    uint32_t m[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { const int64_t r = base + i * 256; m[i] = r < n ? masks[r] : 0u; }
    f32x2_t acc[R][CO / 2];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int p = 0; p < CO / 2; ++p) acc[i][p] = {0.0f, 0.0f};
    // A ROLLED loop over the offsets with the next offset's weights requested before the current one's FMAs: unrolled, the compiler
    // gathers all 27 x CO scalar weight loads at the top and spills the SGPRs they do not fit into VGPR lanes (3 440 v_readlane in
    // the ISA; 3x slower than round 5's kernel on the GPU) -- a scheduling barrier per offset does not stop it
    f32x2_t wcur[CO / 2];
#pragma unroll
    for (int p = 0; p < CO / 2; ++p) wcur[p] = {w[2 * p], w[2 * p + 1]};
#pragma unroll 1
    for (int k = 0; k < 27; ++k) {
        const float *wn = w + (k < 26 ? k + 1 : 26) * CO;                    // uniform address: scalar loads
        f32x2_t wnext[CO / 2];
#pragma unroll
        for (int p = 0; p < CO / 2; ++p) wnext[p] = {wn[2 * p], wn[2 * p + 1]};
        f32x2_t bb[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            // bit k -> 0.0f / 1.0f: sign-extend the bit to 0 / ~0 and keep the bits of 1.0f
            const float bit = __int_as_float((((int32_t)(m[i] << (31 - k))) >> 31) & 0x3f800000);
            bb[i] = {bit, bit};
        }
#pragma unroll
        for (int p = 0; p < CO / 2; ++p)
#pragma unroll
            for (int i = 0; i < R; ++i) acc[i][p] = __builtin_elementwise_fma(bb[i], wcur[p], acc[i][p]);
#pragma unroll
        for (int p = 0; p < CO / 2; ++p) wcur[p] = wnext[p];
    }
    const float sl = (act == FPCC_ACT_PRELU && slope) ? slope[0] : 0.0f;
    // The finished rows leave through LDS so that a wave's store instruction writes consecutive 16-byte pieces (1 KB when the rows are
    // contiguous): a thread storing its own row as CO / 4 pieces writes 16 bytes at a 4 CO-byte stride per instruction, which held the
    // kernel at 2.6 TB/s.  Row stride CO + 1 words: conflict-free for the writes (lane = row) and 2-way at worst for the reads.
    static_assert(R == 1, "one row per thread");
    __shared__ float s_rows[256 * (CO + 1)];
    float *mine = s_rows + threadIdx.x * (CO + 1);
#pragma unroll
    for (int h = 0; h < CO / 4; ++h) {
        mine[4 * h] = finish(acc[0][2 * h].x, bias ? bias[4 * h] : 0.0f, act, sl, clip);
        mine[4 * h + 1] = finish(acc[0][2 * h].y, bias ? bias[4 * h + 1] : 0.0f, act, sl, clip);
        mine[4 * h + 2] = finish(acc[0][2 * h + 1].x, bias ? bias[4 * h + 2] : 0.0f, act, sl, clip);
        mine[4 * h + 3] = finish(acc[0][2 * h + 1].y, bias ? bias[4 * h + 3] : 0.0f, act, sl, clip);
    }
    __syncthreads();
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    constexpr int kPieces = CO / 4;
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
        const int e = j * 256 + (int)threadIdx.x;
        const int r = e / kPieces, piece = e - r * kPieces;
        if (row0 + r < n) {
            const float *src = s_rows + r * (CO + 1) + 4 * piece;
            f32x4 o;
            o.x = src[0]; o.y = src[1]; o.z = src[2]; o.w = src[3];
            *reinterpret_cast<f32x4 *>(out + (row0 + r) * ldo + 4 * piece) = o;
        }
    }
}

template <int CO>
int launch_ones_k3(const uint32_t *masks, int64_t n, const float *w, const float *bias, int act, const float *slope, float clip,
                   float *out, int ldo, hipStream_t s) {
    // one row per thread (the offset loop is rolled and double-buffers its scalar weight loads; see the kernel)
    hipLaunchKernelGGL((k_conv_ones_k3<CO, 1>), dim3(blocks_for(n, 256)), dim3(256), 0, s, masks, n, w, bias, act, slope, clip, out, ldo);
    return check_hip(hipGetLastError(), "k_conv_ones_k3");
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

// Which multi-offset shapes are evaluated grouped (summation order 3): a property of the shape alone -- not of the row count, the
// caller's arguments or a tuning knob (knob 7 exists for A/B experiments under FPCC_EXPERIMENT=1).
static bool use_grouped(int c1, int c2, int c_out, int n_offsets, int groups) {
    return mfma_chunk(c1, c2, c_out) == 32 && n_offsets >= 8 && n_offsets <= kMaxOffsets && groups == 1 && !knob(kKnobGroupedOff);
}

extern "C" int fpcc_conv_f32_order(int c1, int c2, int c_out) { return mfma_chunk(c1, c2, c_out) ? 1 : 0; }

extern "C" int64_t fpcc_conv_f32_ws_bytes(int c1, int c2, int c_out, int n_offsets, int groups, int64_t n_out) {
    // grouped shapes run on the wave kernel, which reads packed weights: without a packed copy from the caller they are packed
    // into the workspace on every call
    if (n_out > 0 && use_grouped(c1, c2, c_out, n_offsets, groups)) return (int64_t)n_offsets * (c1 + c2) * c_out * 4;
    return 0;
}

extern "C" int fpcc_conv_f32_order_ex(int c1, int c2, int c_out, int n_offsets, int groups, int64_t n_out) {
    (void)n_out;                                    // the order is a function of the shape alone (numerics version 2)
    if (use_grouped(c1, c2, c_out, n_offsets, groups)) return 3;
    return mfma_chunk(c1, c2, c_out) ? 1 : 0;
}

extern "C" int fpcc_numerics_version(void) { return FPCC_NUMERICS_VERSION; }

extern "C" int fpcc_conv_ones_k3_f32(const uint32_t *masks, int64_t n, const float *w, const float *bias, int c_out, int act,
                                     const float *slope, float clip, float *out, int ldo, void *stream) {
    if (n < 0 || c_out < 4 || c_out > 32 || c_out % 4 || ldo < c_out || ldo % 4) return fail_arg("conv_ones_k3: 4 <= c_out <= 32, multiple of 4");
    if (n == 0) return FPCC_OK;
    if (!masks || !w || !out || !aligned16(out)) return fail_arg("conv_ones_k3: null or unaligned pointer");
    if (act == FPCC_ACT_PRELU && !slope) return fail_arg("conv_ones_k3: PReLU needs a slope pointer");
    hipStream_t st = as_stream(stream);
    switch (c_out) {
        case 4: return launch_ones_k3<4>(masks, n, w, bias, act, slope, clip, out, ldo, st);
        case 8: return launch_ones_k3<8>(masks, n, w, bias, act, slope, clip, out, ldo, st);
        case 12: return launch_ones_k3<12>(masks, n, w, bias, act, slope, clip, out, ldo, st);
        case 16: return launch_ones_k3<16>(masks, n, w, bias, act, slope, clip, out, ldo, st);
        case 20: return launch_ones_k3<20>(masks, n, w, bias, act, slope, clip, out, ldo, st);
        case 24: return launch_ones_k3<24>(masks, n, w, bias, act, slope, clip, out, ldo, st);
        case 28: return launch_ones_k3<28>(masks, n, w, bias, act, slope, clip, out, ldo, st);
        default: return launch_ones_k3<32>(masks, n, w, bias, act, slope, clip, out, ldo, st);
    }
}

extern "C" int fpcc_conv_debug_stamps(unsigned long long *buf, int64_t n_u64) {
    if (n_u64 < 0 || (n_u64 > 0 && !buf)) return fail_arg("conv_debug_stamps: null buffer");
    const long long cap = buf ? n_u64 : 0;
    FPCC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)));
    FPCC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_cap), &cap, sizeof(cap)));
    return set_lds_stamp_buffer(buf, cap);
}

extern "C" int fpcc_conv_set_tuning(int which, int value) {
    if (which < 0 || which >= kKnobCount) return fail_arg("conv_set_tuning: unknown knob");
    if (which == kKnobGroupedOff) {
        const char *e = getenv("FPCC_EXPERIMENT");
        if (!e || atoi(e) != 1)
            return fail_arg("conv_set_tuning: the summation order is part of the stream format (FPCC_EXPERIMENT=1 to override)");
    }
    const int before = knob(which);
    g_knob[which] = value < 0 ? 0 : value;
    return before;
}

extern "C" int64_t fpcc_conv_packed_floats(int c1, int c2, int c_out, int n_offsets, int groups) {
    // the wave kernel takes the 32-channel-chunk shapes of the MFMA path
    if (mfma_chunk(c1, c2, c_out) != 32 || n_offsets < 1 || n_offsets > kMaxOffsets || groups < 1) return 0;
    return (int64_t)groups * n_offsets * (c1 + c2) * c_out;
}

extern "C" int fpcc_conv_pack_weights_f32(const float *w, int64_t n_mats, int c_in, int c_out, float *w_packed, void *stream) {
    if (n_mats < 0 || c_in < 32 || c_in % 32 || (c_out != 32 && c_out != 64 && c_out != 128))
        return fail_arg("conv_pack_weights: c_in must be a multiple of 32 and c_out one of 32, 64, 128");
    if (n_mats == 0) return FPCC_OK;
    if (!w || !w_packed) return fail_arg("conv_pack_weights: null pointer");
    hipLaunchKernelGGL(k_pack_weights, dim3(blocks_for(n_mats * c_in * c_out, 256)), dim3(256), 0, as_stream(stream), w, n_mats,
                       c_in, c_out, w_packed);
    return check_hip(hipGetLastError(), "k_pack_weights");
}

extern "C" int fpcc_conv_f32(const float *x1, int c1, int ld1, const float *x2, int c2, int ld2, const int32_t *nbr,
                             int n_offsets, int64_t nbr_ks, int64_t nbr_os, const float *w, const float *bias, int c_out,
                             int groups, const int32_t *out_map, int64_t om_os, int64_t om_gs, float *out, int ldo,
                             int64_t n_out, int act, const float *slope, float clip, const int32_t *row_order,
                             void *ws, int64_t ws_bytes, void *stream) {
    return fpcc_conv_f32_pk(x1, c1, ld1, x2, c2, ld2, nbr, n_offsets, nbr_ks, nbr_os, w, nullptr, bias, c_out, groups, out_map,
                            om_os, om_gs, out, ldo, n_out, act, slope, clip, row_order, ws, ws_bytes, stream);
}

namespace fpcc {
PendingEvents &pending_events() {
    static thread_local PendingEvents p;
    return p;
}
}  // namespace fpcc

extern "C" int fpcc_time_next_launch(void *start_event, void *end_event) {
    PendingEvents &p = pending_events();
    p.ev0 = start_event;
    p.ev1 = end_event;
    return FPCC_OK;
}

extern "C" int fpcc_conv_f32_pk(const float *x1, int c1, int ld1, const float *x2, int c2, int ld2, const int32_t *nbr,
                                int n_offsets, int64_t nbr_ks, int64_t nbr_os, const float *w, const float *w_packed,
                                const float *bias, int c_out, int groups, const int32_t *out_map, int64_t om_os,
                                int64_t om_gs, float *out, int ldo, int64_t n_out, int act, const float *slope, float clip,
                                const int32_t *row_order, void *ws, int64_t ws_bytes, void *stream) {
    LaunchBracket timed(stream);
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
               out_map, om_os, om_gs, out, ldo, n_out, act, slope, clip, row_order};
    hipStream_t s = as_stream(stream);
    int ch = mfma_chunk(c1, c2, c_out);
    if (row_order && !ch) return fail_arg("conv_f32: row_order is a feature of the MFMA path (fpcc_conv_f32_order() != 0)");
    if (ch && n_offsets > kMaxOffsets) return fail_arg("conv_f32: the MFMA path supports at most 27 kernel offsets");
    if (ch && !(aligned16(x1) && ld1 % 4 == 0 && aligned16(w) && (c2 == 0 || (aligned16(x2) && ld2 % 4 == 0))))
        return fail_arg("conv_f32: the MFMA path needs 16-byte aligned inputs and row strides that are multiples of 4");
    if (nbr && use_grouped(c1, c2, c_out, n_offsets, groups)) {
        // summation order 3, whatever the caller passed (the order is a property of the shape)
        if (out_map) return fail_arg("conv_f32: grouped shapes do not take an output map");
        const float *wp = w_packed;
        if (!wp) {
            if (!ws || ws_bytes < fpcc_conv_f32_ws_bytes(c1, c2, c_out, n_offsets, groups, n_out) || !aligned16(ws))
                return fail_arg("conv_f32: this shape needs packed weights or a 16-byte aligned workspace of fpcc_conv_f32_ws_bytes() bytes");
            if (int rc = fpcc_conv_pack_weights_f32(w, n_offsets, c1 + c2, c_out, static_cast<float *>(ws), stream)) return rc;
            wp = static_cast<const float *>(ws);
        } else if (!aligned16(wp)) {
            return fail_arg("conv_f32: packed weights must be 16-byte aligned");
        }
        return launch_grouped(a, wp, s);
    }
    // Narrow layers with 48 input channels (3 chunks of 16: 8 MFMAs between two barriers) on large maps are bound by the
    // gathers and the barriers: one stage per kernel offset (CH = 48) has a third as many.  Same per-element chain (the
    // chunks were walked in sequence anyway).  With 64 input channels (2 chunks of 32) the same change loses 2-15 %.
    if (ch == 16 && c2 == 0 && c1 == 48 && c_out <= 64 && n_out * (int64_t)groups >= 32 * 1024)
        return c_out == 64 ? launch_mfma_cfg<2, 48, 4, 1>(a, s) : launch_mfma_cfg<1, 48, 4, 1>(a, s);
    if (ch == 32 && w_packed) {
        if (!aligned16(w_packed)) return fail_arg("conv_f32: packed weights must be 16-byte aligned");
        if (knob(kKnobWaveOn)) {
            const int64_t pw_rows = knob(kKnobPointwiseRows);
            if (n_offsets == 1 && !nbr && !row_order && groups == 1 && !out_map && pw_rows > 0 && n_out >= pw_rows && aligned16(out) &&
                ldo % 4 == 0) {                                   // 16-byte output stores
                const int rc = launch_pointwise(a, w_packed, s);
                if (rc != -1) return rc;
            }
            return launch_wave(a, w_packed, s);
        }
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
    if (c1 == 1 && c2 == 0 && n_offsets == 1 && !nbr && groups == 1 && !out_map && c_out >= 8) {
        hipLaunchKernelGGL(k_conv_c1_pointwise, dim3(blocks_for(n_out * c_out, 256)), dim3(256), 0, s, a);
        return check_hip(hipGetLastError(), "k_conv_c1_pointwise");
    }
    if (c_out == 1) return launch_valu<1>(a, s);
    if (c_out <= 4) return launch_valu<4>(a, s);
    if (c_out <= 8) return launch_valu<8>(a, s);
    return launch_valu<16>(a, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Row order for the MFMA tiles.  A 32-row MFMA block executes every kernel offset that ANY of its rows has; in Morton
// order a block of 32 surface voxels has ~24 of the 27 offsets between them although each row has only ~13, so the
// kernel runs ~1.8x the algorithmic flop.  Sorting the rows of a window of 2^window_log2 consecutive rows by their
// 27-bit neighbour-presence pattern (ranked as a Gray code, so that consecutive patterns differ in few bits) puts rows
// with like patterns into the same block: ~15-16 offsets per block.  Windows keep the gathers inside an L2-sized
// neighbourhood.  Results do not depend on the order (every output row is its own summation chain).
namespace fpcc {
namespace {
__global__ void k_conv_row_keys(const int32_t *__restrict__ nbr, int n_off, int64_t nbr_ks, int64_t nbr_os, int64_t n,
                                int window_log2, int64_t *__restrict__ keys, uint32_t *__restrict__ masks) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    uint32_t m = 0;
    for (int k = 0; k < n_off; ++k) m |= (nbr[(int64_t)k * nbr_ks + o * nbr_os] >= 0 ? 1u : 0u) << k;
    if (masks) masks[o] = m;
    // rank of m in the binary-reflected Gray sequence
    m ^= m >> 1; m ^= m >> 2; m ^= m >> 4; m ^= m >> 8; m ^= m >> 16;
    keys[o] = ((o >> window_log2) << 32) | (int64_t)m;
}

// the same keys from the rows' presence masks (written by the table's producer: fpcc_nbr27_from_parent_ex) -- 4 bytes read per row
// instead of the row's 27 table entries
__global__ void k_conv_row_keys_masks(const uint32_t *__restrict__ masks, int64_t n, int window_log2, int64_t *__restrict__ keys) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    uint32_t m = masks[o];
    m ^= m >> 1; m ^= m >> 2; m ^= m >> 4; m ^= m >> 8; m ^= m >> 16;
    keys[o] = ((o >> window_log2) << 32) | (int64_t)m;
}

// rows of a row-major table in position order: out[p] = rows[order[p]], whole 16-byte pieces (ld / 4 per row): eight consecutive lanes
// move one 128-byte line, a wave's loads are eight scattered lines, its stores 1 KB contiguous
__global__ __launch_bounds__(256) void k_gather_table_rows(const int4 *__restrict__ rows, const int32_t *__restrict__ order, int64_t n,
                                                           int pieces, int4 *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * pieces) return;
    const int64_t p = e / pieces;
    const int j = (int)(e - p * pieces);
    out[e] = rows[(int64_t)order[p] * pieces + j];
}

// one wave per group of `group` (<= 64) consecutive positions of row_order: key = (offsets the group lacks) << 32 | group
__global__ __launch_bounds__(256) void k_conv_tile_keys(const uint32_t *__restrict__ row_masks, int n_off,
                                                        const int32_t *__restrict__ row_order, int64_t n, int group,
                                                        int64_t n_groups, int64_t *__restrict__ keys) {
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= n_groups) return;
    const int64_t pos = g * group + lane;
    uint32_t m = 0;
    if (lane < group && pos < n) m = row_masks[row_order ? (int64_t)row_order[pos] : pos];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o);
    if (lane == 0) keys[g] = ((int64_t)(n_off - __popc(m)) << 32) | g;
}

// Permutation that sorts <= 16384 group keys (fpcc_conv_tile_keys: (lacking offsets) << 32 | group) -- i.e. a STABLE counting sort
// of the groups by their <= 33 possible weights.  One workgroup, three barriers; replaces a 64-bit merge sort of ~10 launches
// per coordinate map (rocprim sorts arrays this small with block sort + log2(n / block) merge passes).
template <int kPer>
__global__ __launch_bounds__(1024) void k_group_counting_order(const int64_t *__restrict__ keys, int n, int32_t *__restrict__ perm) {
    constexpr int kBins = 34;
    __shared__ uint16_t s_wave[16][kBins];
    __shared__ uint32_t s_base[kBins];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = (n + 1023) / 1024;                           // consecutive groups per thread (<= kPer)
    const int g0 = t * per;
    uint8_t bin[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int g = g0 + j;
        bin[j] = (j < per && g < n) ? (uint8_t)min((int)(keys[g] >> 32), kBins - 1) : 255;
    }
    uint16_t before[kBins];                                      // groups of this bin in earlier threads of my wave
#pragma unroll
    for (int b = 0; b < kBins; ++b) {
        int c = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) c += bin[j] == b;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        before[b] = (uint16_t)(incl - c);
        if (lane == 63) s_wave[wv][b] = (uint16_t)incl;
    }
    __syncthreads();
    if (t < kBins) {                                             // per bin: exclusive prefix over the 16 waves, and the bin's total
        uint32_t run = 0;
        for (int w = 0; w < 16; ++w) { const uint32_t c = s_wave[w][t]; s_wave[w][t] = (uint16_t)run; run += c; }
        s_base[t] = run;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int b = 0; b < kBins; ++b) { const uint32_t c = s_base[b]; s_base[b] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        if (bin[j] == 255) continue;
        uint32_t pos = 0;
#pragma unroll
        for (int b = 0; b < kBins; ++b)
            if (bin[j] == b) { pos = s_base[b] + s_wave[wv][b] + before[b]; before[b] += 1; }
        perm[pos] = g0 + j;
    }
}

__global__ void k_conv_regroup_rows(const int32_t *__restrict__ row_order, const int32_t *__restrict__ group_perm, int64_t n,
                                    int group, int64_t n_groups, int32_t *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t g = p / group;
    const int64_t src = g < n_groups ? (int64_t)group_perm[g] * group + (p - g * group) : p;    // the ragged tail stays last
    out[p] = row_order ? row_order[src] : (int32_t)src;
}
}  // namespace
}  // namespace fpcc

extern "C" int fpcc_conv_tile_keys(const uint32_t *row_masks, int n_offsets, const int32_t *row_order, int64_t n, int group,
                                   int64_t *keys_out, void *stream) {
    if (n < 0 || n_offsets < 1 || n_offsets > 32 || group < 1 || group > 64) return fail_arg("conv_tile_keys: sizes out of range");
    const int64_t n_groups = n / group;
    if (n_groups == 0) return FPCC_OK;
    if (!row_masks || !keys_out) return fail_arg("conv_tile_keys: null pointer");
    hipLaunchKernelGGL(k_conv_tile_keys, dim3(blocks_for(n_groups, 4)), dim3(256), 0, as_stream(stream), row_masks, n_offsets,
                       row_order, n, group, n_groups, keys_out);
    return check_hip(hipGetLastError(), "k_conv_tile_keys");
}

extern "C" int fpcc_conv_group_order(const int64_t *group_keys, int64_t n_groups, int32_t *perm_out, void *stream) {
    if (n_groups < 0 || n_groups > 16384) return fail_arg("conv_group_order: at most 16384 groups");
    if (n_groups == 0) return FPCC_OK;
    if (!group_keys || !perm_out) return fail_arg("conv_group_order: null pointer");
    const int per = (int)((n_groups + 1023) / 1024);
    if (per <= 2) hipLaunchKernelGGL(k_group_counting_order<2>, dim3(1), dim3(1024), 0, as_stream(stream), group_keys, (int)n_groups, perm_out);
    else if (per <= 5) hipLaunchKernelGGL(k_group_counting_order<5>, dim3(1), dim3(1024), 0, as_stream(stream), group_keys, (int)n_groups, perm_out);
    else hipLaunchKernelGGL(k_group_counting_order<16>, dim3(1), dim3(1024), 0, as_stream(stream), group_keys, (int)n_groups, perm_out);
    return check_hip(hipGetLastError(), "k_group_counting_order");
}

extern "C" int fpcc_conv_regroup_rows(const int32_t *row_order, const int32_t *group_perm, int64_t n, int group,
                                      int32_t *row_order_out, void *stream) {
    if (n < 0 || group < 1) return fail_arg("conv_regroup_rows: sizes out of range");
    if (n == 0) return FPCC_OK;
    if (!row_order_out || (n / group > 0 && !group_perm)) return fail_arg("conv_regroup_rows: null pointer");
    hipLaunchKernelGGL(k_conv_regroup_rows, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), row_order, group_perm, n,
                       group, n / group, row_order_out);
    return check_hip(hipGetLastError(), "k_conv_regroup_rows");
}

extern "C" int fpcc_conv_row_keys_masks(const uint32_t *masks, int64_t n, int window_log2, int64_t *keys_out, void *stream) {
    if (n < 0 || window_log2 < 5 || window_log2 > 31) return fail_arg("conv_row_keys_masks: bad sizes");
    if (n == 0) return FPCC_OK;
    if (!masks || !keys_out) return fail_arg("conv_row_keys_masks: null pointer");
    hipLaunchKernelGGL(k_conv_row_keys_masks, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), masks, n, window_log2, keys_out);
    return check_hip(hipGetLastError(), "k_conv_row_keys_masks");
}

extern "C" int fpcc_gather_table_rows_i32(const int32_t *rows, int ld, const int32_t *order, int64_t n, int32_t *out, void *stream) {
    if (n < 0 || ld < 4 || ld % 4) return fail_arg("gather_table_rows: ld must be a positive multiple of 4");
    if (n == 0) return FPCC_OK;
    if (!rows || !order || !out || !aligned16(rows) || !aligned16(out)) return fail_arg("gather_table_rows: null or unaligned pointer");
    hipLaunchKernelGGL(k_gather_table_rows, dim3(blocks_for(n * (ld / 4), 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const int4 *>(rows), order, n, ld / 4, reinterpret_cast<int4 *>(out));
    return check_hip(hipGetLastError(), "k_gather_table_rows");
}

extern "C" int fpcc_conv_row_keys(const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os, int64_t n,
                                  int window_log2, int64_t *keys_out, uint32_t *masks_out, void *stream) {
    if (n < 0 || n_offsets < 1 || n_offsets > 32 || window_log2 < 5 || window_log2 > 30)
        return fail_arg("conv_row_keys: sizes out of range");
    if (n == 0) return FPCC_OK;
    if (!nbr || !keys_out) return fail_arg("conv_row_keys: null pointer");
    hipLaunchKernelGGL(k_conv_row_keys, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), nbr, n_offsets, nbr_ks,
                       nbr_os, n, window_log2, keys_out, masks_out);
    return check_hip(hipGetLastError(), "k_conv_row_keys");
}

// ---------------------------------------------------------------------------------------------------------------
// Offset-major table [n_off][n] -> row-major [n][ld] (entries n_off .. ld - 1 of a row = -1): the layout in which a convolution's
// prologue fetches the entries of a row as pieces of one cache line (conv_common.h, table_is_row_major).
namespace fpcc {
namespace {
__global__ __launch_bounds__(256) void k_transpose_table(const int32_t *__restrict__ src, int n_off, int64_t n, int32_t *__restrict__ dst, int ld) {
    __shared__ int32_t s_t[256 * 33];
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const int tid = threadIdx.x;
    for (int k = 0; k < 32; ++k) {
        int32_t v = -1;
        if (k < n_off && row0 + tid < n) v = src[(int64_t)k * n + row0 + tid];          // consecutive threads, consecutive rows
        s_t[tid * 33 + k] = v;
    }
    __syncthreads();
    for (int e = tid; e < 256 * ld; e += 256) {                                            // consecutive threads, consecutive words
        const int r = e / ld, k = e - r * ld;
        if (row0 + r < n) dst[(row0 + r) * ld + k] = k < 32 ? s_t[r * 33 + k] : -1;
    }
}
}  // namespace
}  // namespace fpcc

extern "C" int fpcc_transpose_table_i32(const int32_t *table, int n_offsets, int64_t n, int32_t *rows_out, int ld, void *stream) {
    if (n < 0 || n_offsets < 1 || n_offsets > 32 || ld < n_offsets || ld % 4) return fail_arg("transpose_table: 1 <= n_offsets <= 32, ld >= n_offsets, ld % 4 == 0");
    if (n == 0) return FPCC_OK;
    if (!table || !rows_out) return fail_arg("transpose_table: null pointer");
    hipLaunchKernelGGL(k_transpose_table, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), table, n_offsets, n, rows_out, ld);
    return check_hip(hipGetLastError(), "k_transpose_table");
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
    if (n_off == 27) {
        // all 27 indices, then all 27 gathers (an absent neighbour reads row 0 and is not added), then the chain: the loads
        // are independent of each other and of the sum
        int32_t idx[27];
        float v[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) idx[k] = nbr[(int64_t)k * nbr_ks + o * nbr_os];
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = y[(int64_t)(idx[k] >= 0 ? idx[k] : 0) * ldy + k];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc = idx[k] >= 0 ? acc + v[k] : acc;
    } else {
        for (int k = 0; k < n_off; ++k) {
            const int32_t idx = nbr[(int64_t)k * nbr_ks + o * nbr_os];
            if (idx >= 0) acc = acc + y[(int64_t)idx * ldy + k];
        }
    }
    const float sl = (act == FPCC_ACT_PRELU && slope) ? slope[0] : 0.0f;
    out[o] = finish(acc, bias ? bias[0] : 0.0f, act, sl, clip);
}
}  // namespace
}  // namespace fpcc

// The same on a GENERATED set (row = 8 parent + octant: the decoder side's 8 candidate children of every voxel) WITHOUT its 27-entry
// table: the neighbours of a candidate follow from the parent level's table -- 8 parent rows looked up once, neighbour d of a row with
// octant o is child c of block parent b with, per axis, 2 b + c = d + 2 - o (d in {-1, 0, 1}) -- so the indices are computed in
// registers and only the gathers remain.  Saves writing and re-reading 108 bytes per row (3.8 GB for the 35 M candidates of a batch of
// 16 frames: fpcc_nbr27_from_parent 1.77 ms + the table read inside k_gather_sum).  Same gathers, same ascending-offset sum: same bits.
namespace fpcc {
namespace {
__global__ __launch_bounds__(256) void k_gather_sum_generated(const float *__restrict__ y, int ldy, const int32_t *__restrict__ pnbr, int64_t m,
                                                              int64_t n, const float *__restrict__ bias, int act,
                                                              const float *__restrict__ slope, float clip, float *__restrict__ out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    const int oct = (int)(o & 7);
    const int ox = oct & 1, oy = (oct >> 1) & 1, oz = oct >> 2;
    const int64_t p = o >> 3;
    // the 2 x 2 x 2 block of parents {ox - 1, ox} x {oy - 1, oy} x {oz - 1, oz} (parent offsets): their rows, -1 = absent
    int32_t q[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int px = ox - 1 + (b & 1), py = oy - 1 + ((b >> 1) & 1), pz = oz - 1 + (b >> 2);
        const int pd = (px + 1) + 3 * (py + 1) + 9 * (pz + 1);
        q[b] = pd == 13 ? (int32_t)p : pnbr[(int64_t)pd * m + p];
    }
    int32_t idx[27];
#pragma unroll
    for (int d = 0; d < 27; ++d) {
        // per axis v = d_axis + 2 - o_axis in {0 .. 3}: block bit b = v >> 1, child bit c = v & 1 (d compile-time, o per lane)
        const int vx = (d % 3) + 1 - ox, vy = (d / 3) % 3 + 1 - oy, vz = d / 9 + 1 - oz;
        const int bx = vx >> 1, by = vy >> 1, bz = vz >> 1;
        const int c = (vx & 1) | ((vy & 1) << 1) | ((vz & 1) << 2);
        // q[bx + 2 by + 4 bz] without dynamic register indexing: a select tree over the three block bits
        const int32_t q00 = bx ? q[1] : q[0], q01 = bx ? q[3] : q[2], q10 = bx ? q[5] : q[4], q11 = bx ? q[7] : q[6];
        const int32_t q0 = by ? q01 : q00, q1 = by ? q11 : q10;
        const int32_t qq = bz ? q1 : q0;
        idx[d] = qq < 0 ? -1 : qq * 8 + c;
    }
    float v[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = y[(int64_t)(idx[k] >= 0 ? idx[k] : 0) * ldy + k];
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 27; ++k) acc = idx[k] >= 0 ? acc + v[k] : acc;
    const float sl = (act == FPCC_ACT_PRELU && slope) ? slope[0] : 0.0f;
    out[o] = finish(acc, bias ? bias[0] : 0.0f, act, sl, clip);
}
}  // namespace
}  // namespace fpcc

extern "C" int fpcc_gather_sum_generated_f32(const float *y, int ldy, const int32_t *parent_nbr, int64_t m, const float *bias, int act,
                                             const float *slope, float clip, float *out, void *stream) {
    if (m < 0 || ldy < 27) return fail_arg("gather_sum_generated: bad sizes");
    if (m == 0) return FPCC_OK;
    if (8 * m > (int64_t)INT32_MAX) return fail_arg("gather_sum_generated: more than 2^31 candidates");
    if (!y || !parent_nbr || !out) return fail_arg("gather_sum_generated: null pointer");
    if (act == FPCC_ACT_PRELU && !slope) return fail_arg("gather_sum_generated: PReLU needs a slope pointer");
    hipLaunchKernelGGL(k_gather_sum_generated, dim3(blocks_for(8 * m, 256)), dim3(256), 0, as_stream(stream), y, ldy, parent_nbr, m, 8 * m,
                       bias, act, slope, clip, out);
    return check_hip(hipGetLastError(), "k_gather_sum_generated");
}

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
