// fp32 sparse convolution, multi-offset layers of the LARGE maps, both MFMA operands through LDS (round 4).
//
// k_conv_wave (conv.hip) gathers its A fragments straight into registers -- 32 rows x 32 bytes per instruction, every gathered
// 128-byte line touched by four instructions a quarter stage apart -- and every wave streams its own B fragments from L2 (256 bytes
// per MFMA).  Here a workgroup of 2 R waves owns R row blocks of 32 output rows x all output columns and walks the (kernel offset,
// 32-channel chunk) stages of the UNION of its row blocks' offsets in lockstep:
//   * per stage the B tile (32 x C_out weights, 16 KB at C_out = 128, as it lies in the packed weights) is brought into LDS ONCE per
//     workgroup and each row block's A tile (32 gathered rows x 128 bytes) ONCE, by global_load_lds_dwordx4 (LDS-DMA: no VGPR round
//     trip, no ds_write): an A instruction moves 8 whole 128-byte lines, a B instruction 1 KB contiguous;
//   * wave (r, c) = row block r x column half c takes its MFMA operands from LDS by ds_read_b128: the A fragment of a group of 8
//     channels is 16 bytes of row i at piece 2 g + h -- stored at piece position (2 g + h) ^ ((i >> 1) & 7), the swizzle applied on
//     the SOURCE side of the DMA (its LDS side is lane-linear), so that the 16-lane groups of a ds_read_b128 hit 16 different
//     16-byte slots; the B fragment is lane-linear (1 KB per (group, column block), conflict-free as it lies);
//   * two LDS buffers: the DMAs of stage s + 1 are issued right after the barrier that opens stage s and have the whole stage to land;
//     one `s_waitcnt vmcnt(0)` + one raw s_barrier per stage;
//   * a wave whose row block lacks the stage's offset issues no MFMA for it (it still moves its share of the B tile); the MFMA pipe is
//     kept busy by the other workgroups of the CU meanwhile.
// Summation order 3, folded (conv.hip, k_conv_wave<.., FOLD>): the same FMA chain per output element, bit for bit -- tested against
// the oracle and against the wave kernel (tests/test_gpu_conv.py).
#include "conv_common.h"

namespace fpcc {
namespace {

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int R, int NBW>
struct LdsCfg {
    static constexpr int WAVES = 2 * R, THREADS = 64 * WAVES, ROWS = 32 * R;
    static constexpr int A_BYTES = R * 4096;                  // R row blocks x 32 rows x 128 bytes
    static constexpr int B_HALF = 4 * NBW * 1024;             // one column half: 4 groups of 8 channels x NBW column blocks x 1 KB
    static constexpr int BUF = A_BYTES + 2 * B_HALF;
    static constexpr int LDS = 2 * BUF + 64;                  // two staging buffers + the row blocks' offset masks (+ stamps when traced)
    static constexpr int WGS = (160 * 1024) / LDS > 4 ? 4 : (160 * 1024) / LDS;          // workgroups per CU that LDS admits
    static constexpr int MIN_WAVES = WGS * WAVES / 4 > 4 ? 4 : WGS * WAVES / 4;          // per SIMD
};

// DBG (timing experiments only, results are wrong): bit 0 = every gathered row is row 0, bit 1 = every stage reads the weights of chunk 0,
// bit 2 = no fragment reads (the MFMAs run on whatever the registers hold), bit 3 = no DMAs, bit 4 = no wait / barrier per stage;
// bit 5 (results stay exact without the other bits) = stamps: [0] entry, [1] masks known, [2] first DMAs issued, [40] loop end, [41] folded,
// [42] stored, [43] stages this wave computed, [44] stages of the workgroup, [45] workgroup << 8 | wave, [46] HW_ID, [47] XCC_ID
template <int R, int NBW, int DBG>
__global__ __launch_bounds__(128 * R, (LdsCfg<R, NBW>::MIN_WAVES)) void k_conv_lds(ConvArgs a, const float *__restrict__ wp, int nbt,
                                                                               unsigned n_tiles) {
    using C = LdsCfg<R, NBW>;
    constexpr int ROWS = C::ROWS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS + ((DBG & 32) ? C::WAVES * kStampSlots * 8 : 0)];   // ONE array
    unsigned *const s_mask = reinterpret_cast<unsigned *>(smem + 2 * C::BUF);
    unsigned long long *const s_stamp = reinterpret_cast<unsigned long long *>(smem + C::LDS);
#define FPCC_STAMP(i) do { if (DBG & 32) stamp_lds(&s_stamp[wv * kStampSlots + (i)]); } while (0)

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = wv >> 1, c = wv & 1;                          // row block, column half of this wave
    const int li = lane & 31, lh = lane >> 5;
    if (DBG & 32) {
        for (int i = lane; i < kStampSlots; i += 64) s_stamp[wv * kStampSlots + i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    FPCC_STAMP(0);
    if (DBG & 32) stamp_lds_realtime(&s_stamp[wv * kStampSlots + 38]);
    int n_computed = 0;
    const unsigned tile = a.row_order ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= n_tiles) return;
    const int64_t row0 = (int64_t)tile * ROWS;
    const int c_in = a.c1 + a.c2;
    const int n_chunks = c_in / 32, n1 = a.c1 / 32;
    const int n_off = a.n_off;

    // The neighbour table stays in global memory (an LDS copy of 27 x ROWS entries would cost a workgroup per CU): every wave reads the
    // entries of its own row block once for the offset masks, and per kernel offset two entries per lane for the rows it gathers.
    auto out_row = [&](int pos) -> int32_t {                    // tile position -> output row (-1 past the end)
        const int64_t p = row0 + pos;
        if (p >= a.n_out) return -1;
        return a.row_order ? a.row_order[p] : (int32_t)p;
    };
    const bool by_pos = table_is_row_major(a) && a.row_order;  // row-major table beside a row order: indexed by tile position
    auto nbr_of = [&](int k, int32_t row, int pos) -> int32_t {
        if (row < 0) return -1;
        return a.nbr[(int64_t)k * a.nbr_ks + (by_pos ? row0 + pos : (int64_t)row) * a.nbr_os];
    };
    const int32_t my_row = out_row(32 * r + li);                // output row of lane (i, *)
    unsigned wmask = 0;                                         // offsets present in MY row block
    if (table_is_row_major(a)) {
        // lane (i, h) fetches entries [16 h, 16 h + 16) of row i as four 16-byte pieces of the row's one cache line (conv_common.h)
        const int last_piece = (n_off - 1) >> 2;
        const int64_t trow = my_row < 0 ? 0 : by_pos ? row0 + 32 * r + li : (int64_t)my_row;
        const i32x4 *rowp = reinterpret_cast<const i32x4 *>(a.nbr + trow * a.nbr_os);
        i32x4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = rowp[min(4 * lh + j, last_piece)];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * lh + 4 * j + e, k0 = 4 * j + e;
                const unsigned long long b = __ballot(k < n_off && my_row >= 0 && q[j][e] >= 0);
                if (b & 0xffffffffull) wmask |= 1u << k0;
                if (b >> 32) wmask |= 1u << (k0 + 16);
            }
    } else {
        for (int k = 0; k < n_off; ++k)
            if (__ballot(nbr_of(k, my_row, 32 * r + li) >= 0) != 0ull) wmask |= 1u << k;
    }
    wmask = __builtin_amdgcn_readfirstlane(wmask);
    if (lane == 0 && c == 0) s_mask[r] = wmask;
    __syncthreads();
    unsigned tmask = 0;                                         // offsets present anywhere in the tile: the stages of the workgroup
#pragma unroll
    for (int i = 0; i < R; ++i) tmask |= s_mask[i];
    tmask = __builtin_amdgcn_readfirstlane(tmask);
    FPCC_STAMP(1);

    f32x16 acc[NBW], tsum[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[nb][q] = 0.0f; tsum[nb][q] = 0.0f; }
    int n_folded = 0, cur_g = 0;
    auto fold_acc = [&]() {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                tsum[nb][q] = n_folded ? tsum[nb][q] + acc[nb][q] : acc[nb][q];
                acc[nb][q] = 0.0f;
            }
        ++n_folded;
    };
    auto fold_zero = [&]() {                                    // a group none of whose offsets is present: its partial sum is +0
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) tsum[nb][q] = n_folded ? tsum[nb][q] + 0.0f : 0.0f;
        ++n_folded;
    };
    if (wmask) {
        cur_g = offset_group_of(__ffs(wmask) - 1, n_off);
        for (int gz = 0; gz < cur_g; ++gz) fold_zero();
    }

    const int n_stages = __popc(tmask) * n_chunks;
    if (n_stages > 0) {
        // --- what this wave moves per stage -------------------------------------------------------------------------------------
        // A: pieces p = 2 c, 2 c + 1 of row block r (8 rows x 128 bytes each): lane -> row 8 p + lane / 8, 16-byte piece
        //    (lane % 8) ^ swizzle(row) of that row's chunk; B: the 1-KB blocks q = r, r + R, ... of column half c
        const float *const zero = (const float *)g_zero_row;
        const int64_t ld1 = a.ld1, ld2 = a.ld2;
        const float *a1[2], *a2[2];
        int step[2];
        int ppiece[2], gpos[2];
        int32_t grow[2];                                                     // the output rows whose neighbours this lane gathers
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int prow = 8 * (2 * c + j) + (lane >> 3);
            ppiece[j] = 4 * ((lane & 7) ^ ((prow >> 1) & 7));                // in floats
            gpos[j] = 32 * r + prow;
            grow[j] = out_row(gpos[j]);
        }
        const float *const x2 = a.x2 ? a.x2 : zero;
        int32_t idx_n[2];                                                    // neighbour rows for the NEXT offset, requested one offset ahead
        auto load_offset = [&](int k) {
#pragma unroll
            for (int j = 0; j < 2; ++j) idx_n[j] = nbr_of(k, grow[j], gpos[j]);
        };
        auto set_offset = [&]() {                                            // idx_n -> gather addresses of the current offset
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int32_t idx = idx_n[j];
                if (DBG & 1) idx = idx < 0 ? idx : 0;
                const bool ok = idx >= 0;
                a1[j] = (ok ? a.x1 + (int64_t)idx * ld1 : zero) + ppiece[j];
                a2[j] = (ok ? x2 + (int64_t)idx * ld2 : zero) + ppiece[j];
                step[j] = ok ? 32 : 0;
            }
        };
        const int64_t chunk_floats = (int64_t)4 * nbt * 256;
        const float *const wp_l = wp + ((int64_t)c * NBW) * 256 + lane * 4;
        auto issue = [&](int k, int cc, int buf) {
            if (DBG & 8) return;
            unsigned char *const base = smem + buf * C::BUF;
            const bool in1 = cc < n1;                                        // wave-uniform
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float *src = in1 ? a1[j] + cc * step[j] : a2[j] + (cc - n1) * step[j];
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + r * 4096 + (2 * c + j) * 1024), 16, 0, 0);
            }
            const float *bsrc = wp_l + ((DBG & 2) ? 0 : ((int64_t)k * n_chunks + cc) * chunk_floats);
#pragma unroll
            for (int t = 0; t < (4 * NBW + R - 1) / R; ++t) {
                const int q = r + t * R;                                     // wave-uniform
                if ((4 * NBW) % R == 0 || q < 4 * NBW) {
                    const int g8 = q / NBW, nb = q - g8 * NBW;
                    __builtin_amdgcn_global_load_lds((gptr_t)(bsrc + ((int64_t)g8 * nbt + nb) * 256),
                                                     (lptr_t)(base + C::A_BYTES + c * C::B_HALF + q * 1024), 16, 0, 0);
                }
            }
        };
        // fragment addresses (bytes, inside a buffer)
        const int a_frag = r * 4096 + li * 128;
        const int a_swz = (li >> 1) & 7;
        const int b_frag = C::A_BYTES + c * C::B_HALF + lane * 16;

        unsigned rest = tmask;
        int k = __ffs(rest) - 1, cc = 0;
        load_offset(k);
        set_offset();
        {
            const unsigned r2 = rest & (rest - 1);
            load_offset(r2 ? __ffs(r2) - 1 : k);                             // the second offset's rows
        }
        issue(k, 0, 0);
        FPCC_STAMP(2);
        for (int s = 0; s < n_stages; ++s) {
            // stage s has landed everywhere, and everyone is done with stage s - 1 (whose buffer the next DMAs overwrite)
            if (!(DBG & 16)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            int k_n = k, cc_n = cc + 1;
            if (cc_n == n_chunks) {
                cc_n = 0;
                rest &= rest - 1;
                k_n = rest ? __ffs(rest) - 1 : k;
            }
            if (s + 1 < n_stages) {
                if (cc_n == 0) {
                    set_offset();                                            // rows of k_n were requested an offset ago
                    const unsigned r2 = rest & (rest - 1);
                    load_offset(r2 ? __ffs(r2) - 1 : k_n);
                }
                issue(k_n, cc_n, (s + 1) & 1);
            }
            if ((wmask >> k) & 1u) {                                         // wave-uniform
                if (DBG & 32) ++n_computed;
                if (cc == 0) {
                    const int gk = offset_group_of(k, n_off);
                    if (gk != cur_g) {
                        fold_acc();
                        for (int gz = cur_g + 1; gz < gk; ++gz) fold_zero();
                        cur_g = gk;
                    }
                }
                const unsigned char *const base = smem + (s & 1) * C::BUF;
                // fragments of group g8 + 1 are requested before the MFMAs of group g8 (two register sets): hipcc otherwise reads each
                // group right before its use and exposes the LDS latency four times per stage
                f32x4 av[2], bv[2][NBW];
                if (DBG & 4) {
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        av[sl] = f32x4{1.0f, 2.0f, 3.0f, 4.0f} * (float)lane;
#pragma unroll
                        for (int nb = 0; nb < NBW; ++nb) bv[sl][nb] = f32x4{0.5f, 0.25f, 0.125f, 1.0f} * (float)(nb + lane);
                    }
                }
                auto read_group = [&](int g8, int slot) {
                    if (DBG & 4) { asm volatile("" : "+v"(av[slot])); return; }
                    av[slot] = *reinterpret_cast<const f32x4 *>(base + a_frag + (((2 * g8 + lh) ^ a_swz) << 4));
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb)
                        bv[slot][nb] = *reinterpret_cast<const f32x4 *>(base + b_frag + (g8 * NBW + nb) * 1024);
                };
                read_group(0, 0);
#pragma unroll
                for (int g8 = 0; g8 < 4; ++g8) {
                    const int sl = g8 & 1;
                    if (g8 < 3) read_group(g8 + 1, sl ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl].x, bv[sl][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl].y, bv[sl][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl].z, bv[sl][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl].w, bv[sl][nb].w, acc[nb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            k = k_n;
            cc = cc_n;
        }
    }

    FPCC_STAMP(40);
    if (wmask) { fold_acc(); ++cur_g; }
    for (int gz = cur_g; gz < 4; ++gz) fold_zero();
    FPCC_STAMP(41);

    // output rows of my accumulator registers: register q holds row (q & 3) + 8 (q >> 2) + 4 h of the block.  Slope and bias are read
    // BEFORE the first store: a load between two stores waits (vmcnt(0)) for every store issued so far.
    const float slope = (a.act == FPCC_ACT_PRELU && a.slope) ? a.slope[0] : 0.0f;
    float bias[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) bias[nb] = a.bias ? a.bias[32 * (c * NBW + nb) + li] : 0.0f;
    int32_t orow[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) orow[q] = __shfl(my_row, (q & 3) + 8 * (q >> 2) + 4 * lh);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int64_t o = orow[q];
        if (o < 0) continue;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
            a.out[o * a.ldo + 32 * (c * NBW + nb) + li] = finish(tsum[nb][q], bias[nb], a.act, slope, a.clip);
    }
    if (DBG & 32) {
        FPCC_STAMP(42);
        stamp_lds_realtime(&s_stamp[wv * kStampSlots + 39]);
        if (lane == 0) {
            s_stamp[wv * kStampSlots + 43] = (unsigned long long)n_computed;
            s_stamp[wv * kStampSlots + 44] = (unsigned long long)n_stages;
            s_stamp[wv * kStampSlots + 45] = ((unsigned long long)blockIdx.x << 8) | (unsigned)wv;
            s_stamp[wv * kStampSlots + 46] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            s_stamp[wv * kStampSlots + 47] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        }
        __builtin_amdgcn_wave_barrier();
        const long long w_id = (long long)blockIdx.x * C::WAVES + wv;
        if (g_stamp_buf && (w_id + 1) * kStampSlots <= g_stamp_cap && lane < kStampSlots)
            g_stamp_buf[w_id * kStampSlots + lane] = s_stamp[wv * kStampSlots + lane];
    }
#undef FPCC_STAMP
}

template <int R, int NBW>
int launch_cfg(const ConvArgs &a, const float *wp, int nbt, int dbg, hipStream_t s) {
    constexpr int ROWS = 32 * R;
    const int64_t tiles = (a.n_out + ROWS - 1) / ROWS;
    if (tiles > 0x7fffffffll) return fail_arg("conv_f32: too many tiles");
    const dim3 grid((unsigned)tiles), block(128 * R);
    switch (dbg) {
#define FPCC_LDS_CASE(D) case D: hipLaunchKernelGGL((k_conv_lds<R, NBW, D>), grid, block, 0, s, a, wp, nbt, (unsigned)tiles); break;
        FPCC_LDS_CASE(1) FPCC_LDS_CASE(2) FPCC_LDS_CASE(3) FPCC_LDS_CASE(4) FPCC_LDS_CASE(8) FPCC_LDS_CASE(16) FPCC_LDS_CASE(24) FPCC_LDS_CASE(28) FPCC_LDS_CASE(32) FPCC_LDS_CASE(60)
#undef FPCC_LDS_CASE
        default: hipLaunchKernelGGL((k_conv_lds<R, NBW, 0>), grid, block, 0, s, a, wp, nbt, (unsigned)tiles);
    }
    return check_hip(hipGetLastError(), "k_conv_lds");
}

}  // namespace

int set_lds_stamp_buffer(unsigned long long *buf, long long cap) {
    FPCC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)));
    FPCC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_cap), &cap, sizeof(cap)));
    return FPCC_OK;
}

int launch_conv_lds(const ConvArgs &a, const float *wp, int row_blocks, int dbg, hipStream_t s) {
    const int nbt = a.c_out / 32;
    if ((nbt != 2 && nbt != 4) || a.groups != 1 || a.out_map || !a.nbr || a.n_off > kMaxOffsets || (a.c1 + a.c2) % 32 || a.c1 % 32) return -1;
    if (nbt == 4) {
        if (row_blocks == 2) return launch_cfg<2, 2>(a, wp, nbt, dbg, s);
        if (row_blocks == 3) return launch_cfg<3, 2>(a, wp, nbt, dbg, s);
        if (row_blocks == 4) return launch_cfg<4, 2>(a, wp, nbt, dbg, s);
    } else {
        if (row_blocks == 2) return launch_cfg<2, 1>(a, wp, nbt, dbg, s);
        if (row_blocks == 3) return launch_cfg<3, 1>(a, wp, nbt, dbg, s);
        if (row_blocks == 4) return launch_cfg<4, 1>(a, wp, nbt, dbg, s);
    }
    return -1;
}

}  // namespace fpcc
