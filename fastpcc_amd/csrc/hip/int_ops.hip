// Integer-only pipeline of the lossl_coord_int codec on gfx950: coordinate hash table, int8 sparse convolution / linear
// on v_mfma_i32_32x32x32_i8 with the fixed-point epilogue fused, stand-alone fixed-point epilogues, LUT softmax -> CDF.
//
// Everything here is exact integer arithmetic, so results are independent of summation order and are compared BIT FOR
// BIT with the oracle (oracle/int_ops.c), including the final rANS stream.
//
// The convolution is output-stationary like the float one: a wave owns 32 output rows x (up to) 128 output channels,
// walks the kernel offsets present in its rows and feeds the MFMA straight from global memory (A: 16 B of the gathered
// input row per lane, B: 16 B of a weight row per lane).  No LDS, no barriers: in this codec the whole network is
// ~0.3 TOP (worked example 2) while it issues >150 launches per frame -- it is launch- and
// dependency-bound, not math-bound, so the kernel favours few launches (bias/PReLU/requant fused, raw conv optional)
// over peak MFMA rate.
#include "common.h"
#include <cstddef>
#include <cstdlib>

namespace fpcc {
namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------------------------------
// coordinate hash table (layout and hash of the reference's GPUHashTable so that cached tables are interchangeable)
__device__ __forceinline__ uint64_t coord_hash(int a, int b, int c, int d) {
    uint64_t h = 14695981039346656037ull;
    h = (h ^ (uint32_t)a) * 1099511628211ull;
    h = (h ^ (uint32_t)b) * 1099511628211ull;
    h = (h ^ (uint32_t)c) * 1099511628211ull;
    h = (h ^ (uint32_t)d) * 1099511628211ull;
    return h;
}

__device__ __forceinline__ void table_insert(unsigned long long *keys, int32_t *vals, int cap, uint64_t key, int32_t value) {
    int slot = (int)(key % (uint64_t)cap);
    for (int probes = 0; probes < cap; ++probes) {
        const unsigned long long prev = atomicCAS(&keys[slot], 0ull, (unsigned long long)key);
        if (prev == 0ull || prev == key) {
            vals[slot] = value;
            return;
        }
        slot = slot + 1 == cap ? 0 : slot + 1;
    }
}

__device__ __forceinline__ int32_t table_find(const unsigned long long *keys, const int32_t *vals, int cap, uint64_t key) {
    int slot = (int)(key % (uint64_t)cap);
    int32_t found = 0;
    for (int probes = 0; probes < cap; ++probes) {
        const unsigned long long cur = keys[slot];
        if (cur == key) found = vals[slot];
        if (cur == 0ull) break;
        slot = slot + 1 == cap ? 0 : slot + 1;
    }
    return found;
}

// BF: rows are (batch, x, y, z) -- the layout of the models' coordinate tensors -- instead of the extension's (x, y, z, batch)
template <bool BF>
__global__ void k_hash_insert_coords(unsigned long long *keys, int32_t *vals, int cap, const int4 *__restrict__ coords, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = coords[i];
    table_insert(keys, vals, cap, BF ? coord_hash(c.y, c.z, c.w, c.x) : coord_hash(c.x, c.y, c.z, c.w), i + 1);
}

__global__ void k_hash_insert_keys(unsigned long long *keys, int32_t *vals, int cap, const int64_t *__restrict__ in, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    table_insert(keys, vals, cap, (uint64_t)in[i], i + 1);
}

__global__ void k_hash_lookup_keys(const unsigned long long *keys, const int32_t *vals, int cap,
                                   const int64_t *__restrict__ in, int n, int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = table_find(keys, vals, cap, (uint64_t)in[i]);
}

struct I3 { int v[3]; };

// one thread per (output point, kernel offset); offset order of the reference: odd kernel volume -> first axis fastest,
// even -> last axis fastest; offsets (k % ks) - (ks - 1) / 2
template <bool BF>
__global__ void k_hash_lookup_coords(const unsigned long long *keys, const int32_t *vals, int cap,
                                     const int4 *__restrict__ coords, int n, I3 ks, I3 st, int volume,
                                     int32_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t / volume;
    if (i >= n) return;
    const int k = (int)(t - i * volume);
    const int4 c = coords[i];
    const int base[3] = {BF ? c.y : c.x, BF ? c.z : c.y, BF ? c.w : c.z};
    const int batch = BF ? c.x : c.w;
    int q[3];
    int rem = k;
    if (volume & 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { q[a] = base[a] * st.v[a] + rem % ks.v[a] - (ks.v[a] - 1) / 2; rem /= ks.v[a]; }
    } else {
#pragma unroll
        for (int a = 2; a >= 0; --a) { q[a] = base[a] * st.v[a] + rem % ks.v[a] - (ks.v[a] - 1) / 2; rem /= ks.v[a]; }
    }
    out[i * volume + k] = table_find(keys, vals, cap, coord_hash(q[0], q[1], q[2], batch));
}

// ------------------------------------------------------------------------------------------------------------------
// fixed-point epilogue shared by the fused and the stand-alone kernels
__device__ __forceinline__ int64_t rha(int64_t p, int s) {   // round half away from zero, arithmetic shift
    if (s <= 0) return p;
    const int64_t half = (int64_t)1 << (s - 1);
    return p >= 0 ? (p + half) >> s : -((-p + half) >> s);
}

__device__ __forceinline__ int64_t prelu_q625(int64_t v, int32_t slope) { return v < 0 ? rha(v * (int64_t)slope, 25) : v; }

__device__ __forceinline__ int32_t requant(int64_t v, uint32_t mul, int64_t zp, int shift, int out_bits) {
    const int64_t r = rha(v * (int64_t)mul + zp, shift);
    const int64_t lo = out_bits == 8 ? -128 : (out_bits == 16 ? -32768 : (int64_t)INT32_MIN);
    const int64_t hi = out_bits == 8 ? 127 : (out_bits == 16 ? 32767 : (int64_t)INT32_MAX);
    return (int32_t)(r < lo ? lo : (r > hi ? hi : r));
}

// requant(v, ..., 8) of an int32 v with per-tensor parameters: one 32 x 32 -> 64-bit multiply-add and a branch-free rounding where
// nothing can wrap (multiplier < 2^31, |zp| < 2^61, 1 <= shift <= 60 -- uniform tests), the generic form otherwise; see conv_i8_epilogue
__device__ __forceinline__ int32_t requant8_i32(int32_t v, uint32_t mul, int64_t zp, int shift) {
    if ((int32_t)mul >= 0 && shift >= 1 && shift <= 60 && zp < ((int64_t)1 << 61) && zp > -((int64_t)1 << 61)) {
        const int64_t hm1 = ((int64_t)1 << (shift - 1)) - 1;
        const int64_t q1 = (int64_t)v * (int64_t)(int32_t)mul + (zp + hm1);
        const int64_t r = (q1 + (int64_t)(q1 >= hm1)) >> shift;
        return (int32_t)(r < -128 ? -128 : (r > 127 ? 127 : r));
    }
    return requant((int64_t)v, mul, zp, shift, 8);
}

// Additional int8 copies of an int32 (Q8.23) result, each requantised with the parameters of ONE consumer's RequantFxpToScaledInt8
// (cuda_ops.py:473-509: per-tensor multiplier, zero point, shift): written by the producer's epilogue, so the consumers'
// stand-alone requantisation launches -- one read of the [n, C] int32 matrix and one int8 write each -- disappear.
struct Also8 {                                           // scalar fields on purpose: arrays indexed in a loop end up in scratch
    int8_t *out0; int ld0; int pad0; const uint32_t *mul0; const int64_t *zp0; int shift0;     // columns [c_out, pad) are zeroed
    int8_t *out1; int ld1; int pad1; const uint32_t *mul1; const int64_t *zp1; int shift1;
    int n;
};

struct ConvI8Args {
    const int8_t *a; int lda;               // [n_in][lda] int8, lda % 16 == 0, columns >= c_in zero
    const int32_t *nbr; int n_off; int64_t nbr_ks; int64_t nbr_os; int nbr_bias;   // input row = nbr[...] - nbr_bias, < 0 absent
    const int8_t *w; int ldw;               // [n_off][c_out][ldw] int8, ldw % 16 == 0
    int k_steps;                            // ceil(c_in / 32)
    const int32_t *zp_comp;                 // [n_off][c_out] or NULL
    const int32_t *bias; const int32_t *slope; const uint32_t *mul; const int64_t *zp; int shift; int out_bits;
    void *out; int ldo; int c_out; int64_t n_out; int out_pad;   // columns [c_out, out_pad) of an int8 output are zeroed
    const int32_t *row_order;               // tile position -> output row (NULL: identity)
    const int32_t *res; int ld_res; const int32_t *slope2;   // int32 outputs only: out = clamp_i32(prelu(res + out (wrapping), slope2))
    Also8 also;                                              // int32 outputs only
};

// The descriptor is read from the kernel-argument segment where the epilogue needs it: referenced as `p.also` the compiler loads
// its 13 fields into SGPRs at kernel entry and keeps them live through the main loop (22 SGPR spills, 320 bytes of scratch per
// lane and a 5x slower k_conv_i8_tiled, measured).  Valid because ConvI8Args is the kernels' first (by-value) parameter.
__device__ __forceinline__ Also8 load_also8() {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const char __attribute__((address_space(4))) *CP;
    typedef const int32_t __attribute__((address_space(4))) *IP;
    const CP base = (CP)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ConvI8Args, also);
    Also8 e;
    int32_t *dst = reinterpret_cast<int32_t *>(&e);
#pragma unroll
    for (unsigned i = 0; i < sizeof(Also8) / 4; ++i) dst[i] = ((IP)base)[i];
    return e;
#else
    return Also8{};
#endif
}

__device__ __forceinline__ int32_t pack4_i8(int32_t a, int32_t b, int32_t c, int32_t d) {
    return (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | (int32_t)((uint32_t)(d & 0xff) << 24);
}

// The wave's finished 32 x (32 NB) tile of int32 results, read back (its own stores, L2-resident) four columns per lane and
// written as the consumers' int8 copies.  A compact loop behind the unrolled epilogue on purpose: requantising inside the
// epilogue triples its 64-element body, and the accumulators end up in scratch.
template <int NB>
__device__ __forceinline__ void also8_tail(const ConvI8Args &p, const int32_t *my_rows, int col0, int lane) {
    if (p.out_bits != 32) return;
    const Also8 e = load_also8();
    if (e.n == 0) return;
    __builtin_amdgcn_s_waitcnt(0);                           // this wave's stores of the tile have left
    __builtin_amdgcn_wave_barrier();
    constexpr int Q = 8 * NB;                                // groups of four columns per row
    const int32_t *out = static_cast<const int32_t *>(p.out);
    for (int idx = lane; idx < 32 * Q; idx += 64) {
        const int r = idx / Q, col = col0 + 4 * (idx - r * Q);
        const int64_t o = my_rows[r];
        if (o < 0 || col >= p.ldo) continue;
        i32x4 v = {0, 0, 0, 0};
        if (col < p.c_out) v = *reinterpret_cast<const i32x4 *>(out + o * p.ldo + col);     // c_out, ldo multiples of 4 (checked on the host)
        for (int i = 0; i < e.n; ++i) {
            int8_t *dst = i == 0 ? e.out0 : e.out1;
            const int ld = i == 0 ? e.ld0 : e.ld1, pad = i == 0 ? e.pad0 : e.pad1, sh = i == 0 ? e.shift0 : e.shift1;
            const uint32_t mul = (i == 0 ? e.mul0 : e.mul1)[0];
            const int64_t zp = (i == 0 ? e.zp0 : e.zp1)[0];
            if (col < p.c_out)
                *reinterpret_cast<int32_t *>(dst + o * ld + col) = pack4_i8(requant8_i32(v.x, mul, zp, sh), requant8_i32(v.y, mul, zp, sh),
                                                                            requant8_i32(v.z, mul, zp, sh), requant8_i32(v.w, mul, zp, sh));
            else if (col < pad)
                *reinterpret_cast<int32_t *>(dst + o * ld + col) = 0;
        }
    }
}

// single element form for the stand-alone epilogue kernel
__device__ __forceinline__ void store_also8(const Also8 &e, int64_t o, int col, int c_out, int32_t v) {
    if (e.n > 0) {
        if (col < c_out) e.out0[o * e.ld0 + col] = (int8_t)requant((int64_t)v, e.mul0[0], e.zp0[0], e.shift0, 8);
        else if (col < e.pad0) e.out0[o * e.ld0 + col] = 0;
    }
    if (e.n > 1) {
        if (col < c_out) e.out1[o * e.ld1 + col] = (int8_t)requant((int64_t)v, e.mul1[0], e.zp1[0], e.shift1, 8);
        else if (col < e.pad1) e.out1[o * e.ld1 + col] = 0;
    }
}

// tail of SparseResBlockIn32W8Out32.forward fused behind the convolution's own epilogue (cuda_ops.py:82-92): the int32 tensor
// add of the reference wraps, the PReLU is `prelu` of src/element_wise/prelu.cu
__device__ __forceinline__ int32_t residual_prelu(int32_t v, int32_t r, int32_t slope2) {
    const int32_t x = (int32_t)((uint32_t)v + (uint32_t)r);
    const int64_t y = prelu_q625((int64_t)x, slope2);
    return (int32_t)(y < (int64_t)INT32_MIN ? (int64_t)INT32_MIN : (y > (int64_t)INT32_MAX ? (int64_t)INT32_MAX : y));
}

// Source of every absent operand (missing neighbour, column past c_out, half k-step past the padded row): loads stay
// unconditional, which lets the compiler keep them in flight across the MFMAs of the previous stage.
__device__ __attribute__((aligned(16))) int8_t g_zero_i8[64];

constexpr int kI8MaxOffsets = 64;       // 4x4x4 kernels of the embedding convolutions

// Bias / PReLU / requantisation / residual epilogue of a wave's 32 x (32 NB) tile, shared by the two convolution kernels.  Every load
// it needs -- slope, zero point, per-column bias and multiplier, the residual operand of all its elements -- is issued BEFORE the
// first store: a load placed between two stores makes the wave wait (vmcnt(0)) for every store issued so far, one memory round trip
// per output element (profiles/r04/prologue_epilogue.md; the residual form used to do exactly that, 16 NB times per wave).
template <int NB>
__device__ __forceinline__ void conv_i8_epilogue_generic(const ConvI8Args &p, const i32x16 (&acc)[NB], const int32_t *my_rows, int col0, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int32_t slope = p.slope ? p.slope[0] : 0;
    const int64_t zp = p.zp ? p.zp[0] : 0;
    const int32_t slope2 = p.res ? p.slope2[0] : 0;
    int32_t b[NB];
    uint32_t m[NB];
    bool live[NB], pad[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = col0 + 32 * nb + li;
        live[nb] = col < p.c_out;
        pad[nb] = !live[nb] && p.out_bits == 8 && col < p.out_pad;
        b[nb] = (live[nb] && p.bias) ? p.bias[col] : 0;
        m[nb] = (live[nb] && p.mul) ? p.mul[col] : 0u;
    }
    int64_t orow[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) orow[reg] = my_rows[(reg & 3) + 8 * (reg >> 2) + 4 * lh];
    if (p.out_bits != 8 && p.res) {
        // residual form: all residual operands first (16 NB registers), then the stores
        int32_t r[NB][16];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                r[nb][reg] = (live[nb] && orow[reg] >= 0) ? p.res[orow[reg] * p.ld_res + col0 + 32 * nb + li] : 0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!live[nb]) continue;
            const int col = col0 + 32 * nb + li;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t o = orow[reg];
                if (o < 0) continue;
                int32_t v = acc[nb][reg];
                if (p.mul) {
                    int64_t t = (int64_t)acc[nb][reg] + b[nb];
                    if (p.slope) t = prelu_q625(t, slope);
                    v = requant(t, m[nb], zp, p.shift, p.out_bits);
                }
                static_cast<int32_t *>(p.out)[o * p.ldo + col] = residual_prelu(v, r[nb][reg], slope2);
            }
        }
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!live[nb] && !pad[nb]) continue;
            const int col = col0 + 32 * nb + li;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t o = orow[reg];
                if (o < 0) continue;
                int32_t v = 0;
                if (live[nb]) {
                    if (p.mul) {
                        int64_t t = (int64_t)acc[nb][reg] + b[nb];
                        if (p.slope) t = prelu_q625(t, slope);
                        v = requant(t, m[nb], zp, p.shift, p.out_bits);
                    } else {
                        v = acc[nb][reg];
                    }
                }
                if (p.out_bits == 8) static_cast<int8_t *>(p.out)[o * p.ldo + col] = (int8_t)v;
                else static_cast<int32_t *>(p.out)[o * p.ldo + col] = v;
            }
        }
    }
}

// The same epilogue for the case that occurs -- a full tile (every row and column exists) whose intermediates fit where ONE
// 32 x 32 -> 64-bit multiply-add computes each product -- in ~26 instructions per element without a branch:
//   t = acc + bias as an int32 (overflow raises `bad`);
//   PReLU of a negative t: rha(t * slope, 25) = (t * slope + 2^24 - [slope > 0]) >> 25 -- the sign of the product is known from the
//   (uniform) slope, so the rounding constant rides in the multiply-add's addend; the result must fit int32 again;
//   requantisation: p = y * mul + zp;  rha(p, s) = (p + h + (p >> 63)) >> s with h = 2^(s-1)  [for p < 0: -((-p + h) >> s) =
//   ceil((p - h) / 2^s) = floor((p + h - 1) / 2^s)], computed as q1 = y * mul + (zp + h - 1) and q1 + [q1 >= h - 1];
//   the multiplier must be below 2^31, |zp| < 2^61 and 1 <= s <= 60 (uniform), so that nothing wraps.
// The generic form does every product as a wrapping 64 x 64 multiply and both sides of every sign test: ~80 instructions per element,
// ~30 us of every workgroup's life whatever the convolution itself cost (profiles/r04/int8_conv.md).  A lane that meets a value
// outside these ranges raises `bad`; if any lane of the wave did, the wave runs the generic epilogue afterwards, which overwrites the
// tile with the reference's wrapping arithmetic (same thread, same addresses: ordered).
struct EpiFast {
    int32_t slope, slope2;
    int64_t c25, c25_2;             // 2^24 - [slope > 0]: rounding constant of the PReLU product of a negative input
    int64_t c_neg, hm1; int shift;  // zp + h - 1, h - 1
};
__device__ __forceinline__ int32_t sat_i32(int64_t r) {
    const int32_t lo = (int32_t)r, hi = (int32_t)(r >> 32);
    return hi == (lo >> 31) ? lo : ((hi >> 31) ^ 0x7fffffff);
}
// PReLU of an int32 (exact for every input): the result as int64
__device__ __forceinline__ int64_t prelu_neg(int32_t t, int32_t slope, int64_t c25) {
    return (int64_t)((uint64_t)((int64_t)t * (int64_t)slope) + (uint64_t)c25) >> 25;
}
template <int OUT_BITS>
__device__ __forceinline__ int32_t epi_fast_elem(int32_t a, int32_t b, int32_t m, const EpiFast &e, uint32_t &bad_sign, uint32_t &bad_any) {
    const int32_t t = (int32_t)((uint32_t)a + (uint32_t)b);
    bad_sign |= (uint32_t)((a ^ t) & (b ^ t));                          // sign bit: a + b left int32
    // (a layer without PReLU runs with the identity slope 2^25: (t 2^25 + 2^24 - 1) >> 25 = t, no branch in the element)
    const int64_t yy = prelu_neg(t, e.slope, e.c25);
    const int32_t ylo = (int32_t)yy, yhi = (int32_t)(yy >> 32);
    bad_any |= t < 0 ? (uint32_t)(yhi ^ (ylo >> 31)) : 0u;              // non-zero: the PReLU result left int32
    const int32_t y = t < 0 ? ylo : t;
    const int64_t q1 = (int64_t)((uint64_t)((int64_t)y * (int64_t)m) + (uint64_t)e.c_neg);
    const int64_t r = (q1 + (int64_t)(q1 >= e.hm1)) >> e.shift;
    const int32_t r32 = sat_i32(r);
    return OUT_BITS == 8 ? min(max(r32, -128), 127) : r32;
}

// The element WITHOUT its range checks, for the launches in which no value can leave its range (uniform conditions, tested once per
// wave in conv_i8_epilogue): the accumulator of int8 x int8 products over n_off x c_in terms is bounded, so acc + bias stays an int32
// when the column's bias leaves that much room; a PReLU slope in (-1, 1] (Q6.25) cannot grow a value, so its result stays an int32;
// the multiplier is below 2^31.  HI (requantisation shift >= 32): the quotient is the high word of the 64-bit sum shifted by
// shift - 32 -- an int32 by construction, no 64-bit shift and no saturation.  18 instead of 34 issue slots per element; the epilogue
// is 41 % of a workgroup's life on the finest LiDAR levels (profiles/r05/int8_stage.md).  Same integers as epi_fast_elem.
template <int OUT_BITS, bool HI>
__device__ __forceinline__ int32_t epi_lean_elem(int32_t a, int32_t b, int32_t m, const EpiFast &e) {
    const int32_t t = a + b;
    const int32_t y = t < 0 ? (int32_t)prelu_neg(t, e.slope, e.c25) : t;
    const int64_t q1 = (int64_t)((uint64_t)((int64_t)y * (int64_t)m) + (uint64_t)e.c_neg);
    const int64_t q2 = q1 + (int64_t)(q1 >= e.hm1);
    const int32_t r32 = HI ? ((int32_t)(q2 >> 32) >> (e.shift - 32)) : sat_i32(q2 >> e.shift);
    return OUT_BITS == 8 ? min(max(r32, -128), 127) : r32;
}

// EDBG (timing experiments of the int8-output form, results wrong): 32 = the values are computed but not stored, 64 = stored without
// being computed (the accumulator's low byte)
template <int NB, int EDBG = 0>
__device__ __forceinline__ void conv_i8_epilogue(const ConvI8Args &p, const i32x16 (&acc)[NB], const int32_t *my_rows, int col0, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int64_t zp = p.zp ? p.zp[0] : 0;
    // (uniform) the fast form's conditions; my_rows[31] is the tile's last row of this wave: rows are valid from the front
    const bool fast = p.mul != nullptr && p.shift >= 1 && p.shift <= 60 && zp < ((int64_t)1 << 61) && zp > -((int64_t)1 << 61) &&
                      col0 + 32 * NB <= p.c_out && my_rows[31] >= 0;
    bool need_generic = !fast;
    if (fast) {
        EpiFast e;
        e.slope = p.slope ? p.slope[0] : (1 << 25);
        e.slope2 = p.res ? p.slope2[0] : 0;
        e.c25 = ((int64_t)1 << 24) - (e.slope > 0);
        e.c25_2 = ((int64_t)1 << 24) - (e.slope2 > 0);
        e.shift = p.shift;
        e.hm1 = ((int64_t)1 << (p.shift - 1)) - 1;
        e.c_neg = zp + e.hm1;
        int32_t b[NB], m[NB];
        uint32_t bad_sign = 0u, bad_any = 0u;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = col0 + 32 * nb + li;
            b[nb] = p.bias ? p.bias[col] : 0;
            m[nb] = (int32_t)p.mul[col];
            bad_sign |= (uint32_t)m[nb];                                 // a multiplier of 2^31 or more
        }
        int64_t orow[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) orow[reg] = my_rows[(reg & 3) + 8 * (reg >> 2) + 4 * lh];
        // (uniform) may the range checks be dropped?  |acc| <= n_off x (32 k_steps) x 128 x 128 when nothing but products went into it
        // (any int8 weight, -128 included: the entry points take arbitrary int8 weights)
        const int64_t acc_max = (int64_t)p.n_off * p.k_steps * 32 * (128 * 128);
        bool cols_fit = !p.zp_comp && acc_max < ((int64_t)1 << 31);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int64_t room = (int64_t)INT32_MAX - acc_max;
            cols_fit = cols_fit && m[nb] >= 0 && (int64_t)b[nb] <= room && (int64_t)b[nb] >= -room;
        }
        const bool lean = __builtin_amdgcn_ballot_w64(!cols_fit) == 0ull && e.slope > -(1 << 25) && e.slope <= (1 << 25);
        const bool hi = p.shift >= 32;
        if (p.out_bits == 8) {
            // int8 outputs: a lane owns ONE column of 16 rows per column block -- 64 one-byte stores per lane.  (Setting the tile down in
            // LDS first and writing whole 16-byte pieces of whole rows -- NB store instructions instead of 16 NB -- was built in round 5,
            // bit-exact, and not faster: 92.6 against 89.5 us on the 113 K-row level, profiles/r05/int8_stage.md.)
            int8_t *out = static_cast<int8_t *>(p.out) + col0 + li;
            int32_t sink = 0;
#define FPCC_I8_PUT(ELEM)                                                                                          \
            _Pragma("unroll") for (int reg = 0; reg < 16; ++reg) {                                                  \
                int8_t *o = out + orow[reg] * p.ldo;                                                                \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                 \
                    if (EDBG & 64) o[32 * nb] = (int8_t)acc[nb][reg];                                               \
                    else if (EDBG & 32) sink ^= (ELEM);                                                             \
                    else o[32 * nb] = (int8_t)(ELEM);                                                               \
                }                                                                                                   \
            }
            if (lean && hi) { FPCC_I8_PUT((epi_lean_elem<8, true>(acc[nb][reg], b[nb], m[nb], e))) }
            else if (lean) { FPCC_I8_PUT((epi_lean_elem<8, false>(acc[nb][reg], b[nb], m[nb], e))) }
            else { FPCC_I8_PUT((epi_fast_elem<8>(acc[nb][reg], b[nb], m[nb], e, bad_sign, bad_any))) }
#undef FPCC_I8_PUT
            if ((EDBG & 32) && sink == 0x12345678) out[0] = (int8_t)sink;
        } else if (!p.res) {
            int32_t *out = static_cast<int32_t *>(p.out) + col0 + li;
            if (lean && !hi) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    int32_t *o = out + orow[reg] * p.ldo;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) o[32 * nb] = epi_lean_elem<32, false>(acc[nb][reg], b[nb], m[nb], e);
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    int32_t *o = out + orow[reg] * p.ldo;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) o[32 * nb] = epi_fast_elem<32>(acc[nb][reg], b[nb], m[nb], e, bad_sign, bad_any);
                }
            }
        } else {
            // residual form: all residual operands first (a load between two stores waits for every store issued so far), then the stores
            int32_t r[16][NB];
            const int32_t *res = p.res + col0 + li;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) r[reg][nb] = res[orow[reg] * p.ld_res + 32 * nb];
            int32_t *out = static_cast<int32_t *>(p.out) + col0 + li;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                int32_t *o = out + orow[reg] * p.ldo;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int32_t v = (lean && !hi) ? epi_lean_elem<32, false>(acc[nb][reg], b[nb], m[nb], e)
                                                    : epi_fast_elem<32>(acc[nb][reg], b[nb], m[nb], e, bad_sign, bad_any);
                    const int32_t x = (int32_t)((uint32_t)v + (uint32_t)r[reg][nb]);          // the reference's int32 tensor add wraps
                    o[32 * nb] = x < 0 ? sat_i32(prelu_neg(x, e.slope2, e.c25_2)) : x;        // exact for every int32 x
                }
            }
        }
        need_generic = __builtin_amdgcn_ballot_w64((int32_t)bad_sign < 0 || bad_any != 0u) != 0ull;      // (uniform)
    }
    if (need_generic) conv_i8_epilogue_generic<NB>(p, acc, my_rows, col0, lane);
    also8_tail<NB>(p, my_rows, col0, lane);
}

// One wave = 32 output rows x 32*NB output columns; the four waves of a workgroup are independent (no barrier).
// A stage = (kernel offset present in the wave's rows, k-step of 32 input channels): one 16-byte A load and NB 16-byte
// W loads per lane, NB MFMAs.  The operands of stage s+1 are requested before the MFMAs of stage s are issued.
// SPLIT: blockIdx.z selects ONE kernel offset and the raw int32 sums are added atomically to `acc_out` (integer
// addition is associative: any order gives the same bits) -- for maps too small to fill the chip with row tiles.
template <int NB, bool SPLIT>
__global__ __launch_bounds__(256) void k_conv_i8(ConvI8Args p, int32_t *acc_out, int ld_acc) {
    __shared__ int32_t s_idx[SPLIT ? 1 : 4 * kI8MaxOffsets * 32];      // wave-private slices
    __shared__ int32_t s_row[4 * 32];                                    // output row of each tile position (-1 past the end)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if (row0 >= p.n_out) return;
    const int col0 = blockIdx.y * 32 * NB;
    const int64_t my_row = row0 + li < p.n_out ? (p.row_order ? (int64_t)p.row_order[row0 + li] : row0 + li) : -1;
    int32_t *my_rows = s_row + wave * 32;
    if (lh == 0) my_rows[li] = (int32_t)my_row;
    const int k_lo = SPLIT ? blockIdx.z : 0;
    const int n_k = SPLIT ? 1 : p.n_off;
    int32_t *my_idx = s_idx + (SPLIT ? 0 : wave * kI8MaxOffsets * 32);

    unsigned long long mask = 0ull;
    int32_t idx_one = -1;
    for (int k = 0; k < n_k; ++k) {
        int32_t idx = -1;
        if (my_row >= 0)
            idx = p.nbr ? p.nbr[(int64_t)(k_lo + k) * p.nbr_ks + my_row * p.nbr_os] - p.nbr_bias : (int32_t)my_row;
        if (__ballot(idx >= 0) != 0ull) mask |= 1ull << k;
        if (SPLIT) idx_one = idx;
        else if (lh == 0) my_idx[k * 32 + li] = idx;
    }
    if (SPLIT && mask == 0ull) return;

    i32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0;

    const int8_t *zero = g_zero_i8;
    const int8_t *wcol[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = col0 + 32 * nb + li;
        wcol[nb] = col < p.c_out ? p.w + (int64_t)col * p.ldw : nullptr;
    }
    auto fetch = [&](int k, int s, i32x4 &av, i32x4 (&bv)[NB]) {
        const int32_t idx = SPLIT ? idx_one : my_idx[k * 32 + li];
        const int koff = 32 * s + 16 * lh;       // the second half of the last k-step may lie past a padded row
        const int8_t *ap = (idx >= 0 && koff < p.lda) ? p.a + (int64_t)idx * p.lda + koff : zero;
        av = *reinterpret_cast<const i32x4 *>(ap);
        const int64_t wk = (int64_t)(k_lo + k) * p.c_out * p.ldw + koff;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int8_t *bp = (wcol[nb] && koff < p.ldw) ? wcol[nb] + wk : zero;
            bv[nb] = *reinterpret_cast<const i32x4 *>(bp);
        }
    };

    const int stages = __popcll(mask) * p.k_steps;
    if (stages > 0) {
        unsigned long long rest = mask;
        int k_n = __ffsll((long long)rest) - 1, s_n = 0;
        i32x4 a_n, b_n[NB];
        fetch(k_n, 0, a_n, b_n);
        for (int st = 0; st < stages; ++st) {
            const i32x4 a_c = a_n;
            i32x4 b_c[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b_c[nb] = b_n[nb];
            const int k_c = k_n, s_c = s_n;
            if (++s_n == p.k_steps) {
                s_n = 0;
                rest &= rest - 1;
                if (rest) k_n = __ffsll((long long)rest) - 1;       // after the last stage: re-read a valid one
            }
            fetch(k_n, s_n, a_n, b_n);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_c, b_c[nb], acc[nb], 0, 0, 0);
            if (p.zp_comp && s_c == p.k_steps - 1) {
                const int32_t idx = SPLIT ? idx_one : my_idx[k_c * 32 + li];
                const unsigned rows_present = (unsigned)(__ballot(idx >= 0) & 0xffffffffull);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int col = col0 + 32 * nb + li;
                    const int32_t comp = col < p.c_out ? p.zp_comp[(int64_t)(k_lo + k_c) * p.c_out + col] : 0;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                        if ((rows_present >> rr) & 1u) acc[nb][reg] += comp;
                    }
                }
            }
        }
    }

    if (SPLIT) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = col0 + 32 * nb + li;
            if (col >= p.c_out) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t o = my_rows[(reg & 3) + 8 * (reg >> 2) + 4 * lh];
                if (o >= 0 && acc[nb][reg] != 0) atomicAdd(acc_out + o * ld_acc + col, acc[nb][reg]);
            }
        }
        return;
    }

    conv_i8_epilogue<NB>(p, acc, my_rows, col0, lane);
}

// Workgroup-tiled variant for maps that fill the chip with row tiles: 4 waves = 128 output rows x 32*NB output columns.
// The wave-per-block kernel above reads W straight from global memory, 16 bytes per lane out of every 256-byte weight
// row, once per WAVE: one eighth of every cache line fetched is used and the lines do not survive in the vector L1
// until the next k-step (measured: ~3 us per k-step, L1-refill bound).  Here a stage is (kernel offset, 128 input
// channels): the workgroup fetches the 32*NB x 128-byte W tile ONCE with whole-line loads into LDS (double buffered, one
// barrier per stage, 16-byte pieces XOR-swizzled so that the MFMA operand reads are conflict free), every wave gathers
// its 32 A rows as full 128-byte lines into registers one stage ahead, and issues up to 4*NB MFMAs per stage.  Offsets
// absent from the whole tile are skipped by the workgroup, offsets absent from a wave's rows by that wave.
// Diagnostics (profiles/r05/int8_stage.md): s_memtime stamps of wave 0 of every workgroup of k_conv_i8_tiled<4, 16> in LDS slots, copied to
// the buffer set with fpcc_conv_i8_debug_stamps when the workgroup ends.  A stamp is one LDS store by lane 0 with the exec mask narrowed
// in place (no branch: a branch in the stage loop makes hipcc drain vmcnt).
constexpr int kI8StampSlots = 48;
__device__ unsigned long long *g_i8_stamp_buf = nullptr;
__device__ long long g_i8_stamp_cap = 0;
__device__ __forceinline__ void i8_stamp(unsigned long long *slot) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    const unsigned addr = (unsigned)(uintptr_t)slot;
    asm volatile("s_mov_b64 exec, 1\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(t) : "memory");
}

// NB = 2 (round 5): 64-column tiles.  The 128-column form needs all 256 VGPRs (and spills 14) for its 64 accumulator registers and the
// 64-element epilogue, which holds a SIMD to two waves; on the levels of a LiDAR sweep where a tile executes 2-8 stages the kernel is a
// chain of dependent round trips (row order -> kernel map -> operands) with nothing to hide them behind -- 41 % of its wave cycles are
// parked at a wait (profiles/r05/int8_stage.md).  Half the columns is half the registers: four waves per SIMD, at the price of gathering
// every A row once more.
template <int NB, int DBG = 0, int MW = (NB == 4 ? 2 : 4)>   // DBG (timing experiments, results wrong): 1 no MFMA, 2 no gather (A from the zero row), 4 no W fetch
__global__ __launch_bounds__(256, MW) void k_conv_i8_tiled(ConvI8Args p) {
    constexpr int COLS = 32 * NB;
    constexpr int W_PIECES = COLS * 8 / 256;                     // 16-byte pieces of one W tile per thread
    __shared__ i32x4 sB[2][8 * COLS];
    extern __shared__ int32_t s_idx[];                           // [n_off][128] (dynamic: 13.5 KB for 27 offsets, 32 KB for 64)
    __shared__ int32_t s_row[128];
    __shared__ unsigned long long s_mask[4];
    __shared__ unsigned long long s_i8_stamp[(DBG & 16) ? kI8StampSlots : 1];
#define FPCC_I8_STAMP(i) do { if ((DBG & 16) && wave == 0) i8_stamp(&s_i8_stamp[(i)]); } while (0)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (DBG & 16) {
        if (tid < kI8StampSlots) s_i8_stamp[tid] = 0;
        __syncthreads();
    }
    FPCC_I8_STAMP(0);
    const int li = lane & 31, lh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int col0 = blockIdx.y * COLS;

    // The tile's slice of the kernel map -> LDS.  Row-major table (nbr_ks == 1, the layout of the integer operators): thread t owns tile
    // row t & 127 and every second 16-byte piece of its row, pieces of parity t >> 7 -- a row's n_off entries are n_off / 4 requests of
    // 16 bytes into its one or two cache lines, all issued before the first is used.  (Until round 5 every ENTRY was a 4-byte request:
    // 64 lanes x 4 bytes of 64 different rows per load instruction, i.e. 64 line requests for 256 bytes -- 3 584 requests per workgroup
    // for a 13.8 KB slice, more than its operand gathers and stores together.)  Rows are 4 n_off bytes apart, so the pieces are only
    // dword-aligned.  Any other layout: one request per entry, eight in flight.
    {
        const int r = tid & 127;
        const int64_t row = row0 + r < p.n_out ? (p.row_order ? (int64_t)p.row_order[row0 + r] : row0 + r) : -1;
        if (tid < 128) s_row[tid] = (int32_t)row;
        const int32_t *src = p.nbr ? p.nbr + (row < 0 ? 0 : row) * p.nbr_os : nullptr;
        if (p.nbr && p.nbr_ks == 1 && p.n_off >= 4) {
            typedef int32_t i32x4_dw __attribute__((ext_vector_type(4), aligned(4)));
            const int n_pieces = (p.n_off + 3) >> 2;
            for (int q0 = tid >> 7; q0 < n_pieces; q0 += 8) {
                i32x4_dw v[4];
                int first[4];                                            // first entry of the piece; the last piece of a row whose length
#pragma unroll                                                           // is not a multiple of four starts early instead of reading past the row
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + 2 * u;
                    first[u] = min(4 * (q < n_pieces ? q : q0), p.n_off - 4);
                    v[u] = *reinterpret_cast<const i32x4_dw *>(src + first[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (q0 + 2 * u >= n_pieces) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) s_idx[(first[u] + e) * 128 + r] = row < 0 ? -1 : v[u][e] - p.nbr_bias;
                }
            }
        } else {
            for (int k0 = tid >> 7; k0 < p.n_off; k0 += 16) {
                int32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 2 * u;
                    v[u] = (p.nbr && row >= 0 && k < p.n_off) ? src[(int64_t)k * p.nbr_ks] : 0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 2 * u;
                    if (k < p.n_off) s_idx[k * 128 + r] = row < 0 ? -1 : (p.nbr ? v[u] - p.nbr_bias : (int32_t)row);
                }
            }
        }
    }
    __syncthreads();
    FPCC_I8_STAMP(1);                                            // the tile's slice of the kernel map is in LDS
    const int32_t *my_idx = s_idx + wave * 32 + li;
    // which offsets any of this wave's 32 rows has.  Lane half h looks at the offsets of parity h, eight LDS reads in flight at a time
    // (one read, one ballot per offset in turn -- the earlier form -- was 27 dependent LDS round trips: 4 000 cycles of every
    // workgroup's life whatever the level, profiles/r05/int8_stage.md)
    unsigned long long wmask = 0ull;
    for (int k0 = 0; k0 < p.n_off; k0 += 16) {
        int32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 2 * u + lh;
            v[u] = k < p.n_off ? my_idx[k * 128] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long b = __ballot(v[u] >= 0);
            if (b & 0xffffffffull) wmask |= 1ull << (k0 + 2 * u);
            if (b >> 32) wmask |= 1ull << (k0 + 2 * u + 1);
        }
    }
    if (lane == 0) s_mask[wave] = wmask;
    __syncthreads();
    const unsigned long long tmask = s_mask[0] | s_mask[1] | s_mask[2] | s_mask[3];
    const int n_chunks = (p.k_steps + 3) / 4;
    const int n_stages = __popcll(tmask) * n_chunks;
    FPCC_I8_STAMP(2);                                            // offsets present in the tile known

    i32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0;

    const int8_t *zero = g_zero_i8;
    auto fetch_a = [&](int k, int c, i32x4 (&ra)[4]) {
        const int32_t idx = (DBG & 2) ? -1 : my_idx[k * 128];
        const int8_t *arow = p.a + (int64_t)(idx >= 0 ? idx : 0) * p.lda;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int koff = 128 * c + 32 * s4 + 16 * lh;
            ra[s4] = *reinterpret_cast<const i32x4 *>((idx >= 0 && koff < p.lda) ? arow + koff : zero);
        }
    };
    auto fetch_w = [&](int k, int c, i32x4 (&rw)[W_PIECES]) {
#pragma unroll
        for (int j = 0; j < W_PIECES; ++j) {
            const int piece = tid + 256 * j;
            const int col = col0 + (piece >> 3), koff = 128 * c + 16 * (piece & 7);
            rw[j] = *reinterpret_cast<const i32x4 *>((col < p.c_out && koff < p.ldw && !(DBG & 4))
                                                         ? p.w + ((int64_t)k * p.c_out + col) * p.ldw + koff : zero);
        }
    };
    auto stash_w = [&](int buf, const i32x4 (&rw)[W_PIECES]) {
#pragma unroll
        for (int j = 0; j < W_PIECES; ++j) {
            const int piece = tid + 256 * j;
            const int col = piece >> 3, part = piece & 7;
            sB[buf][part * COLS + (col ^ part)] = rw[j];
        }
    };

    if (n_stages > 0) {
        unsigned long long rest = tmask;
        int k_cur = __ffsll((long long)rest) - 1, c_cur = 0;
        int k_next = k_cur, c_next = 0;
        i32x4 ra_cur[4], ra_nxt[4], rw[W_PIECES];
        fetch_a(k_cur, 0, ra_nxt);
        fetch_w(k_cur, 0, rw);
        stash_w(0, rw);
        __syncthreads();
        FPCC_I8_STAMP(3);                                        // the first W tile is in LDS (one full round trip)
        for (int st = 0; st < n_stages; ++st) {
            FPCC_I8_STAMP(4 + (st < 32 ? st : 32));
            if (++c_next == n_chunks) {
                c_next = 0;
                rest &= rest - 1;
                k_next = rest ? __ffsll((long long)rest) - 1 : k_cur;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) ra_cur[s4] = ra_nxt[s4];
            fetch_a(k_next, c_next, ra_nxt);
            fetch_w(k_next, c_next, rw);
            if ((wmask >> k_cur) & 1ull) {
                const i32x4 *cB = sB[st & 1];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    if (4 * c_cur + s4 < p.k_steps) {
                        const int part = 2 * s4 + lh;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            if (DBG & 1) acc[nb][0] += ra_cur[s4][0] + cB[part * COLS + ((32 * nb + li) ^ part)][0];
                            else
                            acc[nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ra_cur[s4], cB[part * COLS + ((32 * nb + li) ^ part)],
                                                                           acc[nb], 0, 0, 0);
                    }
                }
                if (p.zp_comp && c_cur == n_chunks - 1) {
                    const unsigned rows_present = (unsigned)(__ballot(my_idx[k_cur * 128] >= 0) & 0xffffffffull);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int col = col0 + 32 * nb + li;
                        const int32_t comp = col < p.c_out ? p.zp_comp[(int64_t)k_cur * p.c_out + col] : 0;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            if ((rows_present >> ((reg & 3) + 8 * (reg >> 2) + 4 * lh)) & 1u) acc[nb][reg] += comp;
                    }
                }
            }
            stash_w((st + 1) & 1, rw);
            __syncthreads();
            k_cur = k_next;
            c_cur = c_next;
        }
    }

    FPCC_I8_STAMP(38);                                           // last stage done
    if (DBG & 8) {                                   // no epilogue
        int32_t x = 0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) x ^= acc[nb][r];
        if (x == 0x12345678) static_cast<int32_t *>(p.out)[lane] = x;
        return;
    }
    conv_i8_epilogue<NB, (DBG & (32 | 64))>(p, acc, s_row + wave * 32, col0, lane);
    if (DBG & 16) {
        __builtin_amdgcn_s_waitcnt(0);                           // this wave's stores have been issued and acknowledged
        FPCC_I8_STAMP(39);
        if (wave == 0) {
            if (lane == 0) {
                s_i8_stamp[44] = (unsigned long long)n_stages;
                s_i8_stamp[45] = ((unsigned long long)blockIdx.x << 8) | blockIdx.y;
                s_i8_stamp[46] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
            }
            __builtin_amdgcn_wave_barrier();
            const long long wg = (long long)blockIdx.y * gridDim.x + blockIdx.x;
            if (g_i8_stamp_buf && (wg + 1) * kI8StampSlots <= g_i8_stamp_cap && lane < kI8StampSlots)
                g_i8_stamp_buf[wg * kI8StampSlots + lane] = s_i8_stamp[lane];
        }
    }
#undef FPCC_I8_STAMP
}

// stand-alone epilogue on an int32 matrix: out = clamp(rha((prelu(in + bias)) * mul + zp, shift))
__global__ void k_epilogue_i32(const int32_t *__restrict__ in, int ldi, const int32_t *bias, const int32_t *slope,
                               const uint32_t *mul, int mul_stride, const int64_t *zp, int shift, int out_bits,
                               void *out, int ldo, int64_t n, int ch, int out_pad, const int32_t *row_group,
                               const int32_t *res = nullptr, int ld_res = 0, const int32_t *slope2 = nullptr, Also8 also = Also8{}) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int width = out_pad > ch ? out_pad : ch;
    if (also.n > 0 && also.pad0 > width) width = also.pad0;
    if (also.n > 1 && also.pad1 > width) width = also.pad1;
    if (e >= n * width) return;
    const int64_t r = e / width;
    const int c = (int)(e - r * width);
    int32_t v = 0;
    if (c < ch) {
        const int64_t pc = (row_group ? (int64_t)row_group[r] * ch : 0) + c;      // parameter column of this element
        int64_t t = (int64_t)in[r * ldi + c] + (bias ? (int64_t)bias[pc] : 0);
        if (slope) t = prelu_q625(t, slope[0]);
        v = requant(t, mul[pc * mul_stride], zp ? zp[0] : 0, shift, out_bits);
    }
    if (out_bits == 8) {
        if (c < (out_pad > ch ? out_pad : ch)) static_cast<int8_t *>(out)[r * ldo + c] = (int8_t)v;
    } else {
        const int32_t vf = (c < ch && res) ? residual_prelu(v, res[r * ld_res + c], slope2[0]) : v;
        if (c < ch) static_cast<int32_t *>(out)[r * ldo + c] = vf;
        if (also.n) store_also8(also, r, c, ch, vf);
    }
}

// The same epilogue, four columns per thread: one 16-byte read of the int32 sums, one 4-byte (int8) or 16-byte (int32) write, the int8
// copies packed four to a store.  For ch, the row strides and the paddings multiples of 4 and 16-byte-aligned rows (checked by
// epilogue_v4_ok on the host); the element-wise kernel above wrote single bytes.
__global__ void k_epilogue_i32_v4(const int32_t *__restrict__ in, int ldi, const int32_t *bias, const int32_t *slope,
                                  const uint32_t *mul, int mul_stride, const int64_t *zp, int shift, int out_bits,
                                  void *out, int ldo, int64_t n, int ch, int out_pad, const int32_t *row_group,
                                  const int32_t *res, int ld_res, const int32_t *slope2, Also8 also) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int own = out_pad > ch ? out_pad : ch;
    int width = own;
    if (also.n > 0 && also.pad0 > width) width = also.pad0;
    if (also.n > 1 && also.pad1 > width) width = also.pad1;
    const int w4 = width >> 2;
    if (e >= n * w4) return;
    const int64_t r = e / w4;
    const int c = 4 * (int)(e - r * w4);
    int32_t v[4] = {0, 0, 0, 0};
    if (c < ch) {
        const i32x4 x = *reinterpret_cast<const i32x4 *>(in + r * ldi + c);
        const int32_t xs[4] = {x.x, x.y, x.z, x.w};
        const int64_t pc = (row_group ? (int64_t)row_group[r] * ch : 0) + c;
        const int64_t z = zp ? zp[0] : 0;
        // the lean form of conv_i8_epilogue where its ranges hold (uniform: shift, zero point; per element: the rest), else the generic one
        const bool lean = shift >= 1 && shift <= 60 && z < ((int64_t)1 << 61) && z > -((int64_t)1 << 61);
        EpiFast ef;
        ef.slope = slope ? slope[0] : (1 << 25);
        ef.slope2 = 0;
        ef.c25 = ((int64_t)1 << 24) - (ef.slope > 0);
        ef.c25_2 = 0;
        ef.shift = shift;
        ef.hm1 = lean ? ((int64_t)1 << (shift - 1)) - 1 : 0;
        ef.c_neg = z + ef.hm1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t bj = bias ? bias[pc + j] : 0;
            const uint32_t mj = mul[(pc + j) * mul_stride];
            uint32_t bad_sign = lean ? mj : 0x80000000u, bad_any = 0u;
            const int32_t fast = out_bits == 8 ? epi_fast_elem<8>(xs[j], bj, (int32_t)mj, ef, bad_sign, bad_any)
                                               : epi_fast_elem<32>(xs[j], bj, (int32_t)mj, ef, bad_sign, bad_any);
            if ((int32_t)bad_sign < 0 || bad_any != 0u) {          // (rare) outside the lean form's ranges: the reference's wrapping arithmetic
                int64_t t = (int64_t)xs[j] + (int64_t)bj;
                if (slope) t = prelu_q625(t, slope[0]);
                v[j] = requant(t, mj, z, shift, out_bits);
            } else {
                v[j] = fast;
            }
        }
    }
    if (out_bits == 8) {
        if (c < own) *reinterpret_cast<int32_t *>(static_cast<int8_t *>(out) + r * ldo + c) = pack4_i8(v[0], v[1], v[2], v[3]);
        return;
    }
    if (c < ch && res) {
        const i32x4 q = *reinterpret_cast<const i32x4 *>(res + r * ld_res + c);
        const int32_t qs[4] = {q.x, q.y, q.z, q.w};
        const int32_t s2 = slope2[0];
        const int64_t c25_2 = ((int64_t)1 << 24) - (s2 > 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t x = (int32_t)((uint32_t)v[j] + (uint32_t)qs[j]);                  // the reference's int32 tensor add wraps
            v[j] = x < 0 ? sat_i32(prelu_neg(x, s2, c25_2)) : x;                            // = residual_prelu, exact for every int32
        }
    }
    if (c < ch) {
        const i32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<i32x4 *>(static_cast<int32_t *>(out) + r * ldo + c) = o;
    }
    for (int i = 0; i < also.n; ++i) {
        int8_t *dst = i == 0 ? also.out0 : also.out1;
        const int ld = i == 0 ? also.ld0 : also.ld1, pad = i == 0 ? also.pad0 : also.pad1, sh = i == 0 ? also.shift0 : also.shift1;
        if (c < ch) {
            const uint32_t m8 = (i == 0 ? also.mul0 : also.mul1)[0];
            const int64_t z8 = (i == 0 ? also.zp0 : also.zp1)[0];
            *reinterpret_cast<int32_t *>(dst + r * ld + c) = pack4_i8(requant8_i32(v[0], m8, z8, sh), requant8_i32(v[1], m8, z8, sh),
                                                                      requant8_i32(v[2], m8, z8, sh), requant8_i32(v[3], m8, z8, sh));
        } else if (c < pad) {
            *reinterpret_cast<int32_t *>(dst + r * ld + c) = 0;
        }
    }
}

__host__ inline bool epilogue_v4_ok(const void *in, int ldi, const void *out, int ldo, int out_bits, int ch, int out_pad,
                                    const void *res, int ld_res, const Also8 &a) {
    auto al = [](const void *p, unsigned m) { return (reinterpret_cast<uintptr_t>(p) & (m - 1)) == 0; };
    if (ch % 4 || ldi % 4 || !al(in, 16)) return false;
    if (out_bits == 8) { if (ldo % 4 || out_pad % 4 || !al(out, 4)) return false; }
    else if (ldo % 4 || !al(out, 16)) return false;
    if (res && (ld_res % 4 || !al(res, 16))) return false;
    if (a.n > 0 && (a.ld0 % 4 || a.pad0 % 4 || !al(a.out0, 4))) return false;
    if (a.n > 1 && (a.ld1 % 4 || a.pad1 % 4 || !al(a.out1, 4))) return false;
    return true;
}

// out = clamp_i32(prelu_q625(a (+ b)))
__global__ void k_prelu_i32(const int32_t *__restrict__ a, const int32_t *__restrict__ b, const int32_t *slope, int64_t n,
                            int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t x = a[i];
    if (b) x = (int32_t)((uint32_t)x + (uint32_t)b[i]);    // the reference's int32 tensor add wraps
    const int64_t v = prelu_q625((int64_t)x, slope[0]);
    out[i] = (int32_t)(v < (int64_t)INT32_MIN ? (int64_t)INT32_MIN : (v > (int64_t)INT32_MAX ? (int64_t)INT32_MAX : v));
}

// ------------------------------------------------------------------------------------------------------------------
// LUT softmax.  lut[k] = llround(65536 * exp(-k / 512)), k = 0 .. 6144 (filled once by fpcc_int_init on the host)
constexpr int kLutSize = 12 * 512 + 1;
__device__ int32_t g_exp_lut[kLutSize];

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per row, rows of up to 256 entries (4 per lane).  MODE 0: Q0.32 probabilities (softmax_int32);
// MODE 1: uint16 CDF rows of batch_quantize_pmf; MODE 2: (start, freq - 1) of one symbol per row.
template <int MODE>
__global__ __launch_bounds__(256) void k_softmax_rows(const int32_t *__restrict__ in, int64_t n, int c, int pre_shift,
                                                      uint32_t *__restrict__ prob_out, uint16_t *__restrict__ cdf_out,
                                                      const int16_t *__restrict__ sym, uint16_t *__restrict__ start_out,
                                                      uint16_t *__restrict__ freqm1_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int32_t *x = in + row * c;
    int32_t q[4];
    int m = INT32_MIN;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = 4 * lane + j;
        q[j] = e < c ? (x[e] >> pre_shift) : INT32_MIN;
        m = max(m, q[j]);
    }
    const int top = wave_max(m) + 64;
    int32_t ex[4];
    int local = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = 4 * lane + j;
        int id = (int)(((int64_t)top - (int64_t)q[j]) >> 7);
        id = id > kLutSize - 1 ? kLutSize - 1 : id;
        ex[j] = e < c ? g_exp_lut[id] : 0;
        local += ex[j];
    }
    const int sum = wave_sum(local);
    const uint64_t inv = sum > 0 ? (((uint64_t)1 << 32) + (uint64_t)(sum >> 1)) / (uint64_t)sum : ((uint64_t)1 << 32) / (uint64_t)c;
    uint32_t pr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t t = (uint64_t)(uint32_t)ex[j] * inv;
        pr[j] = t > 0xffffffffull ? 0xffffffffu : (uint32_t)t;
    }
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * lane + j < c) prob_out[row * c + 4 * lane + j] = pr[j];
        return;
    }
    // frequencies and their inclusive prefix sum over the row
    uint32_t f[4];
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[j] = 4 * lane + j < c ? (uint32_t)(((uint64_t)pr[j] * (uint64_t)(65536 - c)) >> 32) + 1u : 0u;
        run += f[j];
        f[j] = run;                                  // inclusive within the lane
    }
    uint32_t scan = run;                             // inclusive scan of lane totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(scan, o);
        if (lane >= o) scan += up;
    }
    const uint32_t before = scan - run;
    uint32_t edge[4];                                // upper edge of entry e; the last entry is forced to 65535
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = 4 * lane + j;
        edge[j] = e == c - 1 ? 65535u : before + f[j];
    }
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * lane + j < c) cdf_out[row * c + 4 * lane + j] = (uint16_t)edge[j];
        return;
    }
    // MODE 2: the range of symbol s: [edge[s-1], edge[s]) with edge[-1] = 0 and the last upper edge 65536
    const int s = (int)(uint16_t)sym[row];
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = 4 * lane + j;
        if (e == s - 1) lo = edge[j];
        if (e == s) hi = e == c - 1 ? 65536u : edge[j];
    }
    lo = (uint32_t)wave_sum((int)lo);                // exactly one lane holds a non-zero contribution
    hi = (uint32_t)wave_sum((int)hi);
    if (lane == 0) {
        start_out[row] = (uint16_t)lo;
        freqm1_out[row] = (uint16_t)(hi - lo - 1u);
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace fpcc

using namespace fpcc;

static bool g_i8_stamps_on = false;
extern "C" int fpcc_conv_i8_debug_stamps(unsigned long long *buf, int64_t n_u64) {
    if (n_u64 < 0 || (n_u64 > 0 && !buf)) return fail_arg("conv_i8_debug_stamps: null buffer");
    const long long cap = buf ? n_u64 : 0;
    FPCC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_i8_stamp_buf), &buf, sizeof(buf)));
    FPCC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_i8_stamp_cap), &cap, sizeof(cap)));
    g_i8_stamps_on = buf != nullptr;
    return FPCC_OK;
}

extern "C" int fpcc_int_init(void) {
    static bool done = false;
    if (done) return FPCC_OK;
    static int32_t host_lut[kLutSize];
    for (int k = 0; k < kLutSize; ++k) host_lut[k] = (int32_t)llround(exp(-(double)k / 512.0) * 65536.0);
    FPCC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_exp_lut), host_lut, sizeof host_lut));
    done = true;
    return FPCC_OK;
}

extern "C" int fpcc_hash_insert_coords(int64_t *table_keys, int32_t *table_vals, int64_t capacity, const int32_t *coords,
                                       int64_t n, void *stream) {
    if (capacity < 1 || capacity > INT32_MAX || n < 0 || n > capacity) return fail_arg("hash_insert_coords: bad capacity / n");
    if (n == 0) return FPCC_OK;
    if (!table_keys || !table_vals || !coords || !aligned16(coords)) return fail_arg("hash_insert_coords: null or unaligned pointer");
    hipLaunchKernelGGL(k_hash_insert_coords<false>, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<unsigned long long *>(table_keys), table_vals, (int)capacity,
                       reinterpret_cast<const int4 *>(coords), (int)n);
    FPCC_LAUNCHED(k_hash_insert_coords);
    return FPCC_OK;
}

extern "C" int fpcc_hash_lookup_coords(const int64_t *table_keys, const int32_t *table_vals, int64_t capacity,
                                       const int32_t *coords, int64_t n, const int32_t *kernel_sizes_host,
                                       const int32_t *strides_host, int32_t *out, void *stream) {
    if (capacity < 1 || capacity > INT32_MAX || n < 0 || !kernel_sizes_host || !strides_host)
        return fail_arg("hash_lookup_coords: bad arguments");
    I3 ks, st;
    int volume = 1;
    for (int a = 0; a < 3; ++a) {
        ks.v[a] = kernel_sizes_host[a];
        st.v[a] = strides_host[a];
        if (ks.v[a] < 1 || st.v[a] < 1) return fail_arg("hash_lookup_coords: kernel sizes and strides must be positive");
        volume *= ks.v[a];
    }
    if (n == 0) return FPCC_OK;
    if (!table_keys || !table_vals || !coords || !out || !aligned16(coords)) return fail_arg("hash_lookup_coords: null or unaligned pointer");
    hipLaunchKernelGGL(k_hash_lookup_coords<false>, dim3(blocks_for(n * volume, kThreads)), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<const unsigned long long *>(table_keys), table_vals, (int)capacity,
                       reinterpret_cast<const int4 *>(coords), (int)n, ks, st, volume, out);
    FPCC_LAUNCHED(k_hash_lookup_coords);
    return FPCC_OK;
}

extern "C" int fpcc_hash_insert_coords_bxyz(int64_t *table_keys, int32_t *table_vals, int64_t capacity, const int32_t *coords,
                                       int64_t n, void *stream) {
    if (capacity < 1 || capacity > INT32_MAX || n < 0 || n > capacity) return fail_arg("hash_insert_coords: bad capacity / n");
    if (n == 0) return FPCC_OK;
    if (!table_keys || !table_vals || !coords || !aligned16(coords)) return fail_arg("hash_insert_coords: null or unaligned pointer");
    hipLaunchKernelGGL(k_hash_insert_coords<true>, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<unsigned long long *>(table_keys), table_vals, (int)capacity,
                       reinterpret_cast<const int4 *>(coords), (int)n);
    FPCC_LAUNCHED(k_hash_insert_coords);
    return FPCC_OK;
}

extern "C" int fpcc_hash_lookup_coords_bxyz(const int64_t *table_keys, const int32_t *table_vals, int64_t capacity,
                                       const int32_t *coords, int64_t n, const int32_t *kernel_sizes_host,
                                       const int32_t *strides_host, int32_t *out, void *stream) {
    if (capacity < 1 || capacity > INT32_MAX || n < 0 || !kernel_sizes_host || !strides_host)
        return fail_arg("hash_lookup_coords: bad arguments");
    I3 ks, st;
    int volume = 1;
    for (int a = 0; a < 3; ++a) {
        ks.v[a] = kernel_sizes_host[a];
        st.v[a] = strides_host[a];
        if (ks.v[a] < 1 || st.v[a] < 1) return fail_arg("hash_lookup_coords: kernel sizes and strides must be positive");
        volume *= ks.v[a];
    }
    if (n == 0) return FPCC_OK;
    if (!table_keys || !table_vals || !coords || !out || !aligned16(coords)) return fail_arg("hash_lookup_coords: null or unaligned pointer");
    hipLaunchKernelGGL(k_hash_lookup_coords<true>, dim3(blocks_for(n * volume, kThreads)), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<const unsigned long long *>(table_keys), table_vals, (int)capacity,
                       reinterpret_cast<const int4 *>(coords), (int)n, ks, st, volume, out);
    FPCC_LAUNCHED(k_hash_lookup_coords);
    return FPCC_OK;
}

extern "C" int fpcc_hash_insert_keys(int64_t *table_keys, int32_t *table_vals, int64_t capacity, const int64_t *keys,
                                     int64_t n, void *stream) {
    if (capacity < 1 || capacity > INT32_MAX || n < 0 || n > capacity) return fail_arg("hash_insert_keys: bad capacity / n");
    if (n == 0) return FPCC_OK;
    if (!table_keys || !table_vals || !keys) return fail_arg("hash_insert_keys: null pointer");
    hipLaunchKernelGGL(k_hash_insert_keys, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<unsigned long long *>(table_keys), table_vals, (int)capacity, keys, (int)n);
    FPCC_LAUNCHED(k_hash_insert_keys);
    return FPCC_OK;
}

extern "C" int fpcc_hash_lookup_keys(const int64_t *table_keys, const int32_t *table_vals, int64_t capacity,
                                     const int64_t *keys, int64_t n, int32_t *out, void *stream) {
    if (capacity < 1 || capacity > INT32_MAX || n < 0) return fail_arg("hash_lookup_keys: bad capacity / n");
    if (n == 0) return FPCC_OK;
    if (!table_keys || !table_vals || !keys || !out) return fail_arg("hash_lookup_keys: null pointer");
    hipLaunchKernelGGL(k_hash_lookup_keys, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<const unsigned long long *>(table_keys), table_vals, (int)capacity, keys, (int)n, out);
    FPCC_LAUNCHED(k_hash_lookup_keys);
    return FPCC_OK;
}

// rows up to which multi-offset int8 convolutions are evaluated one workgroup per (row tile, kernel offset)
constexpr int64_t kI8SplitMaxRows = 8192;

static bool i8_split(const int32_t *nbr, int n_offsets, int64_t n_out) {
    return nbr && n_offsets >= 8 && n_out <= kI8SplitMaxRows;
}

extern "C" int64_t fpcc_conv_i8_ws_bytes(int has_nbr, int n_offsets, int has_requant, int c_out, int64_t n_out) {
    return (has_nbr && n_offsets >= 8 && n_out > 0 && n_out <= kI8SplitMaxRows && has_requant) ? n_out * (int64_t)c_out * 4 : 0;
}

extern "C" int fpcc_conv_i8(const int8_t *a, int c_in, int lda, const int32_t *nbr, int n_offsets, int64_t nbr_ks,
                            int64_t nbr_os, int nbr_bias, const int8_t *w, int ldw, const int32_t *zp_comp,
                            const int32_t *bias, const int32_t *slope, const uint32_t *requant_mul, const int64_t *zero_point,
                            int shift, int out_bits, void *out, int ldo, int out_pad, int c_out, int64_t n_out,
                            const int32_t *row_order, void *ws, int64_t ws_bytes, void *stream) {
    return fpcc_conv_i8_res(a, c_in, lda, nbr, n_offsets, nbr_ks, nbr_os, nbr_bias, w, ldw, zp_comp, bias, slope, requant_mul,
                            zero_point, shift, out_bits, out, ldo, out_pad, c_out, n_out, row_order, nullptr, 0, nullptr, ws,
                            ws_bytes, stream);
}

static int make_also(const fpcc_requant8 *also, int n_also, int out_bits, bool has_requant, int c_out, int covered_cols, Also8 &e) {
    e = Also8{};
    if (n_also < 0 || n_also > 2) return fail_arg("at most two additional int8 outputs");
    if (n_also == 0) return FPCC_OK;
    if (!also || out_bits != 32 || !has_requant) return fail_arg("additional int8 outputs need a requantised int32 primary output");
    if (c_out % 4) return fail_arg("additional int8 outputs need a channel count that is a multiple of 4");
    for (int i = 0; i < n_also; ++i) {
        const fpcc_requant8 &q = also[i];
        if (!q.out || !q.requant_mul || !q.zero_point || q.shift < 0 || q.ld < c_out || q.pad > q.ld || q.pad > covered_cols || q.ld % 4 ||
            q.pad % 4 || (reinterpret_cast<uintptr_t>(q.out) & 3))
            return fail_arg("additional int8 output: null / unaligned pointer, negative shift or row stride / padding out of range");
        const int pad = q.pad > c_out ? q.pad : c_out;
        if (i == 0) { e.out0 = q.out; e.ld0 = q.ld; e.pad0 = pad; e.mul0 = q.requant_mul; e.zp0 = q.zero_point; e.shift0 = q.shift; }
        else { e.out1 = q.out; e.ld1 = q.ld; e.pad1 = pad; e.mul1 = q.requant_mul; e.zp1 = q.zero_point; e.shift1 = q.shift; }
    }
    e.n = n_also;
    return FPCC_OK;
}

extern "C" int fpcc_conv_i8_res(const int8_t *a, int c_in, int lda, const int32_t *nbr, int n_offsets, int64_t nbr_ks,
                                int64_t nbr_os, int nbr_bias, const int8_t *w, int ldw, const int32_t *zp_comp,
                                const int32_t *bias, const int32_t *slope, const uint32_t *requant_mul,
                                const int64_t *zero_point, int shift, int out_bits, void *out, int ldo, int out_pad, int c_out,
                                int64_t n_out, const int32_t *row_order, const int32_t *residual, int ld_res,
                                const int32_t *slope2, void *ws, int64_t ws_bytes, void *stream) {
    return fpcc_conv_i8_also(a, c_in, lda, nbr, n_offsets, nbr_ks, nbr_os, nbr_bias, w, ldw, zp_comp, bias, slope, requant_mul, zero_point,
                             shift, out_bits, out, ldo, out_pad, c_out, n_out, row_order, residual, ld_res, slope2, nullptr, 0, ws, ws_bytes,
                             stream);
}

extern "C" int fpcc_conv_i8_also(const int8_t *a, int c_in, int lda, const int32_t *nbr, int n_offsets, int64_t nbr_ks,
                                 int64_t nbr_os, int nbr_bias, const int8_t *w, int ldw, const int32_t *zp_comp,
                                 const int32_t *bias, const int32_t *slope, const uint32_t *requant_mul,
                                 const int64_t *zero_point, int shift, int out_bits, void *out, int ldo, int out_pad, int c_out,
                                 int64_t n_out, const int32_t *row_order, const int32_t *residual, int ld_res,
                                 const int32_t *slope2, const fpcc_requant8 *also, int n_also, void *ws, int64_t ws_bytes,
                                 void *stream) {
    return fpcc::conv_i8_run(a, c_in, lda, nbr, n_offsets, nbr_ks, nbr_os, nbr_bias, w, ldw, zp_comp, bias, slope, requant_mul, zero_point,
                             shift, out_bits, out, ldo, out_pad, c_out, n_out, row_order, residual, ld_res, slope2, also, n_also, ws, ws_bytes,
                             false, stream);
}

// ws_zeroed: the offset-split form's accumulator (ws) already holds zeros -- a caller that runs several such layers clears all their
// accumulators with one memset (fpcc_int_level_*)
int fpcc::conv_i8_run(const int8_t *a, int c_in, int lda, const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os, int nbr_bias,
                      const int8_t *w, int ldw, const int32_t *zp_comp, const int32_t *bias, const int32_t *slope,
                      const uint32_t *requant_mul, const int64_t *zero_point, int shift, int out_bits, void *out, int ldo, int out_pad,
                      int c_out, int64_t n_out, const int32_t *row_order, const int32_t *residual, int ld_res, const int32_t *slope2,
                      const fpcc_requant8 *also, int n_also, void *ws, int64_t ws_bytes, bool ws_zeroed, void *stream) {
    if (residual && (!slope2 || out_bits != 32 || !requant_mul || ld_res < c_out))
        return fail_arg("conv_i8: a fused residual needs its PReLU slope, a requantised int32 output and ld_res >= c_out");
    if (n_out < 0 || c_in < 1 || c_out < 1 || n_offsets < 1 || n_offsets > kI8MaxOffsets)
        return fail_arg("conv_i8: sizes out of range (n_offsets must be 1..64)");
    if (n_out == 0) return FPCC_OK;
    if (!a || !w || !out) return fail_arg("conv_i8: null pointer");
    if (!nbr && n_offsets != 1) return fail_arg("conv_i8: identity map needs n_offsets == 1");
    if (lda % 16 || ldw % 16 || lda < c_in || ldw < c_in || !aligned16(a) || !aligned16(w))
        return fail_arg("conv_i8: rows of A and W must be 16-byte aligned with strides that are multiples of 16 and >= c_in");
    if (out_bits != 8 && out_bits != 32) return fail_arg("conv_i8: out_bits must be 8 or 32");
    if (requant_mul && shift < 0) return fail_arg("conv_i8: negative requant shift");
    if (!requant_mul && out_bits != 32) return fail_arg("conv_i8: raw accumulators are int32");
    if (ldo < c_out || out_pad > ldo) return fail_arg("conv_i8: output row stride too small");
    ConvI8Args p{a, lda, nbr, n_offsets, nbr_ks, nbr_os, nbr_bias, w, ldw, (c_in + 31) / 32, zp_comp,
                 bias, slope, requant_mul, zero_point, shift, out_bits, out, ldo, c_out, n_out, out_pad, row_order,
                 residual, ld_res, slope2, Also8{}};
    const int width = out_pad > c_out ? out_pad : c_out;
    if (int rc = make_also(also, n_also, out_bits, requant_mul != nullptr, c_out, width, p.also)) return rc;
    const unsigned gx = (unsigned)((n_out + 127) / 128);
    hipStream_t s = as_stream(stream);
    if (i8_split(nbr, n_offsets, n_out)) {
        // raw sums are accumulated atomically: into the caller's workspace when an epilogue follows, else into `out`
        int32_t *acc = static_cast<int32_t *>(out);
        int ld_acc = ldo;
        if (requant_mul) {
            if (!ws || ws_bytes < n_out * (int64_t)c_out * 4) return fail_arg("conv_i8: this shape needs fpcc_conv_i8_ws_bytes() bytes of workspace");
            acc = static_cast<int32_t *>(ws);
            ld_acc = c_out;
        }
        if (!(ws_zeroed && requant_mul))
            if (int rc = check_hip(hipMemsetAsync(acc, 0, (size_t)n_out * ld_acc * 4, s), "conv_i8: memset")) return rc;
        const dim3 grid(gx, (c_out + 127) / 128, n_offsets);
        if (c_out > 64) hipLaunchKernelGGL((k_conv_i8<4, true>), grid, dim3(256), 0, s, p, acc, ld_acc);
        else if (c_out > 32) hipLaunchKernelGGL((k_conv_i8<2, true>), grid, dim3(256), 0, s, p, acc, ld_acc);
        else hipLaunchKernelGGL((k_conv_i8<1, true>), grid, dim3(256), 0, s, p, acc, ld_acc);
        FPCC_LAUNCHED(k_conv_i8_split);
        if (requant_mul) {
            const int pad8 = out_bits == 8 ? out_pad : 0;
            if (epilogue_v4_ok(acc, c_out, out, ldo, out_bits, c_out, pad8, residual, ld_res, p.also))
                hipLaunchKernelGGL(k_epilogue_i32_v4, dim3(blocks_for(n_out * ((width + 3) / 4), kThreads)), dim3(kThreads), 0, s, acc, c_out,
                                   bias, slope, requant_mul, 1, zero_point, shift, out_bits, out, ldo, n_out, c_out, pad8, nullptr,
                                   residual, ld_res, slope2, p.also);
            else
                hipLaunchKernelGGL(k_epilogue_i32, dim3(blocks_for(n_out * width, kThreads)), dim3(kThreads), 0, s, acc, c_out, bias,
                                   slope, requant_mul, 1, zero_point, shift, out_bits, out, ldo, n_out, c_out, pad8, nullptr,
                                   residual, ld_res, slope2, p.also);
            FPCC_LAUNCHED(k_epilogue_i32);
        }
        return FPCC_OK;
    }
    static const int tiled = [] { const char *e = getenv("FPCC_I8_TILED"); return e ? atoi(e) : 1; }();
    // linear layers (identity map): the wave-per-block kernel reads 16 bytes of every 256-byte weight row per lane and spends ~3 us per
    // k-step on L1 refills whatever the row count (26-29 us for C -> 255 on a 300-row level); the tiled kernel stages W through LDS
    static const int linear_tiled_min = [] { const char *e = getenv("FPCC_I8_LINEAR_TILED_MIN"); return e ? atoi(e) : 1; }();
    static const int tiled_dbg_env = [] { const char *e = getenv("FPCC_I8_DBG"); return e ? atoi(e) : 0; }();
    const int tiled_dbg = g_i8_stamps_on ? 16 : tiled_dbg_env;
    const unsigned idx_bytes = (unsigned)n_offsets * 128u * 4u;             // the tiled kernel's slice of the kernel map in LDS
    if (width > 64 && tiled && n_out >= 2048 && tiled_dbg) {
        const dim3 grid(gx, (width + 127) / 128);
        // the ablations profiles/r04/int8_conv.md cites: 1 no MFMA, 7 no MFMA / gather / W fetch, 8 no epilogue, 15 all of them
        if (tiled_dbg == 1) hipLaunchKernelGGL((k_conv_i8_tiled<4, 1>), grid, dim3(256), idx_bytes, s, p);
        else if (tiled_dbg == 8) hipLaunchKernelGGL((k_conv_i8_tiled<4, 8>), grid, dim3(256), idx_bytes, s, p);
        else if (tiled_dbg == 15) hipLaunchKernelGGL((k_conv_i8_tiled<4, 15>), grid, dim3(256), idx_bytes, s, p);
        else if (tiled_dbg == 16) hipLaunchKernelGGL((k_conv_i8_tiled<4, 16>), grid, dim3(256), idx_bytes, s, p);   // stamps, results exact
        else if (tiled_dbg == 32) hipLaunchKernelGGL((k_conv_i8_tiled<4, 32>), grid, dim3(256), idx_bytes, s, p);   // epilogue computes, stores nothing
        else if (tiled_dbg == 64) hipLaunchKernelGGL((k_conv_i8_tiled<4, 64>), grid, dim3(256), idx_bytes, s, p);   // epilogue stores, computes nothing
        else hipLaunchKernelGGL((k_conv_i8_tiled<4, 7>), grid, dim3(256), idx_bytes, s, p);
    } else if (width > 64 && tiled && n_out >= (nbr ? 2048 : linear_tiled_min)) {
        static const int tile_nb = [] { const char *e = getenv("FPCC_I8_NB"); return e ? atoi(e) : 4; }();
        static const int tile_mw = [] { const char *e = getenv("FPCC_I8_MW"); return e ? atoi(e) : 4; }();
        if (tile_nb == 2 && tile_mw == 3) hipLaunchKernelGGL((k_conv_i8_tiled<2, 0, 3>), dim3(gx, (width + 63) / 64), dim3(256), idx_bytes, s, p);
        else if (tile_nb == 2) hipLaunchKernelGGL((k_conv_i8_tiled<2>), dim3(gx, (width + 63) / 64), dim3(256), idx_bytes, s, p);
        else hipLaunchKernelGGL((k_conv_i8_tiled<4>), dim3(gx, (width + 127) / 128), dim3(256), idx_bytes, s, p);
    } else if (width > 64) {
        hipLaunchKernelGGL((k_conv_i8<4, false>), dim3(gx, (width + 127) / 128), dim3(256), 0, s, p, nullptr, 0);
    } else if (width > 32) {
        hipLaunchKernelGGL((k_conv_i8<2, false>), dim3(gx, 1), dim3(256), 0, s, p, nullptr, 0);
    } else {
        hipLaunchKernelGGL((k_conv_i8<1, false>), dim3(gx, 1), dim3(256), 0, s, p, nullptr, 0);
    }
    FPCC_LAUNCHED(k_conv_i8);
    return FPCC_OK;
}

extern "C" int fpcc_epilogue_i32(const int32_t *in, int ldi, const int32_t *bias, const int32_t *slope,
                                 const uint32_t *requant_mul, int mul_per_channel, const int64_t *zero_point, int shift,
                                 int out_bits, void *out, int ldo, int out_pad, int64_t n, int ch, const int32_t *row_group,
                                 void *stream) {
    return fpcc_epilogue_i32_also(in, ldi, bias, slope, requant_mul, mul_per_channel, zero_point, shift, out_bits, out, ldo, out_pad, n, ch,
                                  row_group, nullptr, 0, stream);
}

extern "C" int fpcc_epilogue_i32_also(const int32_t *in, int ldi, const int32_t *bias, const int32_t *slope,
                                      const uint32_t *requant_mul, int mul_per_channel, const int64_t *zero_point, int shift,
                                      int out_bits, void *out, int ldo, int out_pad, int64_t n, int ch, const int32_t *row_group,
                                      const fpcc_requant8 *also, int n_also, void *stream) {
    if (n < 0 || ch < 1 || shift < 0) return fail_arg("epilogue_i32: bad sizes or negative shift");
    if (n == 0) return FPCC_OK;
    if (!in || !requant_mul || !out) return fail_arg("epilogue_i32: null pointer");
    if (out_bits != 8 && out_bits != 16 && out_bits != 32) return fail_arg("epilogue_i32: out_bits must be 8, 16 or 32");
    if (out_bits == 16) return fail_arg("epilogue_i32: int16 outputs are not used by any in-scope model");
    int width = out_pad > ch ? out_pad : ch;
    if (ldo < width || ldi < ch) return fail_arg("epilogue_i32: row stride too small");
    Also8 e;
    if (int rc = make_also(also, n_also, out_bits, true, ch, 1 << 30, e)) return rc;
    if (e.n > 0 && e.pad0 > width) width = e.pad0;
    if (e.n > 1 && e.pad1 > width) width = e.pad1;
    const int pad8 = out_bits == 8 ? out_pad : 0;
    if (epilogue_v4_ok(in, ldi, out, ldo, out_bits, ch, pad8, nullptr, 0, e))
        hipLaunchKernelGGL(k_epilogue_i32_v4, dim3(blocks_for(n * ((width + 3) / 4), kThreads)), dim3(kThreads), 0, as_stream(stream), in, ldi,
                           bias, slope, requant_mul, mul_per_channel ? 1 : 0, zero_point, shift, out_bits, out, ldo, n, ch, pad8, row_group,
                           nullptr, 0, nullptr, e);
    else
        hipLaunchKernelGGL(k_epilogue_i32, dim3(blocks_for(n * width, kThreads)), dim3(kThreads), 0, as_stream(stream), in, ldi,
                           bias, slope, requant_mul, mul_per_channel ? 1 : 0, zero_point, shift, out_bits, out, ldo, n, ch,
                           pad8, row_group, nullptr, 0, nullptr, e);
    FPCC_LAUNCHED(k_epilogue_i32);
    return FPCC_OK;
}

namespace fpcc {
namespace {
// occupancy bits as requantised int8 features behind the columns of an int8 activation matrix: out[r][col0 + k] = q(bit ? one : 0)
// with the consumer's RequantFxpToScaledInt8 parameters, columns [col0 + 8, pad) zeroed -- the int8 image of
// requant(cat(R, bits << shift)) without building the int32 concatenation
__global__ void k_fill_bits_i8(const uint8_t *__restrict__ bits, int64_t n, int32_t fxp_one, const uint32_t *mul, const int64_t *zp, int shift,
                               int8_t *__restrict__ out, int ld, int col0, int pad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = pad - col0;
    if (e >= n * w) return;
    const int64_t r = e / w;
    const int k = (int)(e - r * w);
    int8_t v = 0;
    if (k < 8) v = (int8_t)requant(bits[8 * r + k] ? (int64_t)fxp_one : 0, mul[0], zp[0], shift, 8);
    out[r * ld + col0 + k] = v;
}
}  // namespace
}  // namespace fpcc

extern "C" int fpcc_fill_bits_i8(const uint8_t *bits, int64_t n, int32_t fxp_one, const uint32_t *requant_mul, const int64_t *zero_point,
                                 int shift, int8_t *out, int ld, int col0, int pad, void *stream) {
    if (n < 0 || shift < 0 || col0 < 0 || pad < col0 + 8 || pad > ld) return fail_arg("fill_bits_i8: bad sizes");
    if (n == 0) return FPCC_OK;
    if (!bits || !requant_mul || !zero_point || !out) return fail_arg("fill_bits_i8: null pointer");
    hipLaunchKernelGGL(k_fill_bits_i8, dim3(blocks_for(n * (pad - col0), kThreads)), dim3(kThreads), 0, as_stream(stream), bits, n, fxp_one,
                       requant_mul, zero_point, shift, out, ld, col0, pad);
    FPCC_LAUNCHED(k_fill_bits_i8);
    return FPCC_OK;
}

extern "C" int fpcc_prelu_i32(const int32_t *a, const int32_t *b, const int32_t *slope, int64_t n, int32_t *out, void *stream) {
    if (n < 0) return fail_arg("prelu_i32: n < 0");
    if (n == 0) return FPCC_OK;
    if (!a || !slope || !out) return fail_arg("prelu_i32: null pointer");
    hipLaunchKernelGGL(k_prelu_i32, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), a, b, slope, n, out);
    FPCC_LAUNCHED(k_prelu_i32);
    return FPCC_OK;
}

static int softmax_common(int mode, const int32_t *in, int64_t n, int c, int pre_shift, uint32_t *prob, uint16_t *cdf,
                          const int16_t *sym, uint16_t *start, uint16_t *freqm1, void *stream) {
    if (n < 0 || c < 2 || c > 256 || pre_shift < 0 || pre_shift > 31) return fail_arg("softmax: rows of 2..256 entries, shift 0..31");
    if (n == 0) return FPCC_OK;
    if (!in) return fail_arg("softmax: null pointer");
    if (int rc = fpcc_int_init()) return rc;
    const dim3 grid(blocks_for(n, 4)), block(256);
    hipStream_t s = as_stream(stream);
    if (mode == 0) hipLaunchKernelGGL((k_softmax_rows<0>), grid, block, 0, s, in, n, c, pre_shift, prob, cdf, sym, start, freqm1);
    else if (mode == 1) hipLaunchKernelGGL((k_softmax_rows<1>), grid, block, 0, s, in, n, c, pre_shift, prob, cdf, sym, start, freqm1);
    else hipLaunchKernelGGL((k_softmax_rows<2>), grid, block, 0, s, in, n, c, pre_shift, prob, cdf, sym, start, freqm1);
    FPCC_LAUNCHED(k_softmax_rows);
    return FPCC_OK;
}

extern "C" int fpcc_softmax_i32(const int32_t *in, int64_t n, int c, uint32_t *out, void *stream) {
    if (n > 0 && !out) return fail_arg("softmax_i32: null pointer");
    return softmax_common(0, in, n, c, 0, out, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int fpcc_logits_to_cdf16(const int32_t *logits, int64_t n, int c, int pre_shift, uint16_t *cdf_out, void *stream) {
    if (n > 0 && !cdf_out) return fail_arg("logits_to_cdf16: null pointer");
    return softmax_common(1, logits, n, c, pre_shift, nullptr, cdf_out, nullptr, nullptr, nullptr, stream);
}

extern "C" int fpcc_logits_to_ranges(const int32_t *logits, int64_t n, int c, int pre_shift, const int16_t *symbols,
                                     uint16_t *start_out, uint16_t *freq_minus_1_out, void *stream) {
    if (n > 0 && (!symbols || !start_out || !freq_minus_1_out)) return fail_arg("logits_to_ranges: null pointer");
    return softmax_common(2, logits, n, c, pre_shift, nullptr, nullptr, symbols, start_out, freq_minus_1_out, stream);
}
