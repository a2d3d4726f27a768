/*
 * TEST ORACLE -- NOT PRODUCT CODE.
 *
 * Plain-C restatement of the integer-only pipeline of the reference's `int_sparse_conv_ext` (CUDA + CUTLASS, no CPU
 * path exists in the reference; every entry point TORCH_CHECKs a CUDA device).  The arithmetic is fully specified by
 * scalar device functions in the reference, which this file follows:
 *
 *   orc_requant*            /root/reference/lib/int_sparse_conv/src/element_wise/requant.cu:7-26
 *   orc_bias_requant*       /root/reference/lib/int_sparse_conv/src/element_wise/bias_requant.cu:6-28
 *   orc_prelu_requant*      /root/reference/lib/int_sparse_conv/src/element_wise/prelu_requant.cu:6-35
 *   orc_bias_prelu_requant* /root/reference/lib/int_sparse_conv/src/element_wise/bias_prelu_requant.cu:6-37
 *   orc_prelu_i32           /root/reference/lib/int_sparse_conv/src/element_wise/prelu.cu:6-21
 *   orc_softmax_i32         /root/reference/lib/int_sparse_conv/src/softmax.cu:41-106 (LUT: build_lut :108-117)
 *   orc_gather_gemm_i8      /root/reference/lib/int_sparse_conv/src/gather_gemm_scatter.cu:11-144 and gemm.cu:11-127
 *                           (int8 x int8 -> int32 dot products, saturating accumulate), driven as in
 *                           /root/reference/lib/int_sparse_conv/cuda_ops.py:153-166
 *
 * Parity: UNPINNED against the reference binary (needs nvcc + CUTLASS + an NVIDIA GPU).  Pinned by hand-derived
 * known-answer vectors in tests/test_oracle_int.py (rounding at +-0.5 LSB, saturation, shift 0, row_sum fallback) and by
 * a checksum of the exponent table against the table text in the reference (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* round-half-away-from-zero arithmetic right shift; identity for s <= 0 */
static inline int64_t rha(int64_t p, int s) {
    if (s <= 0) return p;
    int64_t half = (int64_t)1 << (s - 1);
    return p >= 0 ? (p + half) >> s : -((-p + half) >> s);
}

static inline int64_t clamp64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline int64_t prelu_q625(int64_t v, int32_t slope) { return v < 0 ? rha(v * (int64_t)slope, 25) : v; }

static inline void type_range(int out_bits, int64_t *lo, int64_t *hi) {
    if (out_bits == 8) { *lo = -128; *hi = 127; }
    else if (out_bits == 16) { *lo = -32768; *hi = 32767; }
    else { *lo = INT32_MIN; *hi = INT32_MAX; }
}

/*
 * One routine covers the four fused epilogues: bias and slope are optional (NULL).
 * in [n, ch] int32; bias [ch] int32; slope [1] int32 (Q6.25); mul [ch] uint32; zp int64; out written as int32 values
 * already clamped to the range of an `out_bits`-bit signed integer.
 */
void orc_epilogue_i32(const int32_t *in, const int32_t *bias, const int32_t *slope, const uint32_t *mul, int64_t zp,
                      int shift, int out_bits, int64_t n, int64_t ch, int32_t *out) {
    int64_t lo, hi;
    type_range(out_bits, &lo, &hi);
    for (int64_t i = 0; i < n * ch; ++i) {
        int64_t c = i % ch;
        int64_t v = (int64_t)in[i] + (bias ? (int64_t)bias[c] : 0);
        if (slope) v = prelu_q625(v, slope[0]);
        int64_t p = v * (int64_t)mul[c] + zp;
        out[i] = (int32_t)clamp64(rha(p, shift), lo, hi);
    }
}

void orc_prelu_i32(const int32_t *in, int32_t slope, int64_t n, int32_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (int32_t)clamp64(prelu_q625((int64_t)in[i], slope), INT32_MIN, INT32_MAX);
}

#define ORC_LUT_N (12 * 512 + 1)

void orc_exp_lut(int32_t *lut) {
    for (int k = 0; k < ORC_LUT_N; ++k) lut[k] = (int32_t)llround(exp(-(double)k / 512.0) * 65536.0);
}

/* in [n, c] int32 Q15.16 -> out [n, c] uint32 Q0.32 */
void orc_softmax_i32(const int32_t *in, int64_t n, int64_t c, uint32_t *out) {
    static int32_t lut[ORC_LUT_N];
    static int ready = 0;
    if (!ready) { orc_exp_lut(lut); ready = 1; }
    for (int64_t r = 0; r < n; ++r) {
        const int32_t *row = in + r * c;
        int32_t m = INT32_MIN;
        for (int64_t j = 0; j < c; ++j) if (row[j] > m) m = row[j];
        const int32_t top = m + 64;
        int32_t sum = 0;
        for (int64_t j = 0; j < c; ++j) {
            int32_t id = (top - row[j]) >> 7;
            if (id > ORC_LUT_N - 1) id = ORC_LUT_N - 1;
            sum += lut[id];
        }
        uint64_t inv = sum > 0 ? (((uint64_t)1 << 32) + (uint64_t)(sum >> 1)) / (uint64_t)sum
                               : ((uint64_t)1 << 32) / (uint64_t)c;
        for (int64_t j = 0; j < c; ++j) {
            int32_t id = (top - row[j]) >> 7;
            if (id > ORC_LUT_N - 1) id = ORC_LUT_N - 1;
            uint64_t p = (uint64_t)lut[id] * inv;
            out[r * c + j] = p > 0xffffffffull ? 0xffffffffu : (uint32_t)p;
        }
    }
}

static inline int32_t sat_add32(int64_t a) { return (int32_t)clamp64(a, INT32_MIN, INT32_MAX); }

/*
 * D[scatter[l], :] = D[scatter[l], :] + A[gather[l], :] . B^T      (B is [c_out, c_in] int8; D int32, accumulated in place)
 * gather/scatter NULL means identity over n_pairs rows.
 */
void orc_gather_gemm_i8(const int8_t *a, int64_t c_in, const int8_t *b, int64_t c_out,
                        const int32_t *gather, const int32_t *scatter, int64_t n_pairs, int32_t *d) {
    for (int64_t l = 0; l < n_pairs; ++l) {
        const int8_t *ar = a + (gather ? gather[l] : l) * c_in;
        int32_t *dr = d + (scatter ? scatter[l] : l) * c_out;
        for (int64_t j = 0; j < c_out; ++j) {
            const int8_t *br = b + j * c_in;
            int64_t acc = dr[j];
            for (int64_t c = 0; c < c_in; ++c) acc += (int64_t)ar[c] * (int64_t)br[c];
            dr[j] = sat_add32(acc);
        }
    }
}
