/*
 * TEST ORACLE -- NOT PRODUCT CODE.
 *
 * Plain-C restatement of the float sparse convolution the reference obtains from MinkowskiEngine
 * (un-vendored third-party dependency, "MinkowskiEngine ~= 0.5.4", /root/reference/README.md:49-50; call sites
 * /root/reference/lib/minkowski_sparse_conv_layers.py:67-91).  Published algorithm restated:
 *
 *      out[o, :] = sum_k  X[in_k(o), :] @ W[k]  + bias        (fp32; W laid out [K, C_in, C_out])
 *
 * for every kernel offset k that has an input row in_k(o) for output row o (the "kernel map").  How the map is
 * built is in oracle/coords.py; this file only does the arithmetic.
 *
 * Parity: UNPINNED against MinkowskiEngine (it cannot be built or run in this image).  It is
 * pinned against oracle/sparse_conv.py's torch formulation (index_select -> mm -> index_add_, the structure of ME's
 * CPU backend) in tests/test_oracle_float.py::test_conv_mm_equals_chain.
 *
 * fp32 addition is not associative and the reference leaves the summation order to cuBLAS.  This restatement fixes
 * one: a single fused-multiply-add chain per output element, kernel offsets ascending, input channels in the order
 * selected by `order`:
 *      order 0: natural 0,1,2,...
 *      order 1: inside every aligned group of 8 channels: 0,4,1,5,2,6,3,7  (the order in which gfx950's
 *               v_mfma_f32_32x32x2_f32 consumes a 16-byte-per-lane A fragment; a trailing partial group keeps
 *               its relative order, as if zero padded)
 *      order 2: per kernel offset its own chain (channels as in order 1) starting from zero; the offsets' partial
 *               sums are then added in ascending offset order -- the association of a per-offset
 *               gather-GEMM-scatter-add evaluation (and of MinkowskiEngine's own loop over kernel offsets)
 *      order 3: the kernel offsets are cut into four fixed contiguous groups [ceil(g K / 4), ceil((g + 1) K / 4)); each group
 *               is its own order-1 chain from zero over the offsets present, and the four partial sums are added as
 *               ((g0 + g1) + g2) + g3 (a group without a present offset contributes its zero) -- the association of four
 *               waves that share the offsets of one output block
 * so that a device kernel documenting the same order can be compared bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_ACT_NONE = 0, ORC_ACT_PRELU = 1, ORC_ACT_RELU = 2 };

static inline int64_t chan_at(int64_t pos, int order) {
    if (order == 0) return pos;   /* orders 1, 2 and 3 share the channel permutation */
    static const int p8[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    return (pos & ~(int64_t)7) + p8[pos & 7];
}

/*
 * x1 [n_in, c1] (row stride ld1 floats), optional x2 [n_in, c2] (row stride ld2): the input row is their concatenation.
 * nbr [K, n_out] int32 (input row or -1); NULL means K == 1 and in(o) == o.
 * w [K, c1+c2, c_out]; bias [c_out] or NULL.
 * out_map NULL (row o -> o) or [n_out] (row o is written to out_map[o], skipped when negative).
 * out row stride = ldo floats.
 * act: 0 none, 1 PReLU(slope), 2 ReLU.  clip > 0 clamps to [-clip, clip] afterwards.
 */
void orc_gather_conv_f32(const float *x1, int64_t c1, int64_t ld1, const float *x2, int64_t c2, int64_t ld2,
                         const int32_t *nbr, int64_t K, int64_t n_out,
                         const float *w, const float *bias, int64_t c_out,
                         const int32_t *out_map, float *out, int64_t ldo,
                         int act, float slope, float clip, int order) {
    const int64_t c_in = c1 + c2;
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)c_out);
        float *part = (float *)malloc(sizeof(float) * (size_t)c_out * 4);
#pragma omp for schedule(dynamic, 64)
        for (int64_t o = 0; o < n_out; ++o) {
            int64_t dst = out_map ? out_map[o] : o;
            if (dst < 0) continue;
            for (int64_t j = 0; j < c_out; ++j) acc[j] = 0.0f;
            if (order == 3) for (int64_t j = 0; j < 4 * c_out; ++j) part[j] = 0.0f;
            for (int64_t k = 0; k < K; ++k) {
                int64_t r = nbr ? nbr[k * n_out + o] : o;
                if (r < 0) continue;
                const float *wk = w + k * c_in * c_out;
                int grp = 0;
                if (order == 3) while (grp < 3 && ((grp + 1) * K + 3) / 4 <= k) ++grp;     /* group whose range holds offset k */
                float *tgt = order == 2 ? part : order == 3 ? part + grp * c_out : acc;
                if (order == 2) for (int64_t j = 0; j < c_out; ++j) part[j] = 0.0f;
                /* orders 1 and 2 walk whole groups of 8: channels past c_in (zero padding on the device) are skipped */
                const int64_t span = order == 0 ? c_in : ((c_in + 7) & ~(int64_t)7);
                for (int64_t pos = 0; pos < span; ++pos) {
                    int64_t c = chan_at(pos, order);
                    if (c >= c_in) continue;
                    float xv = c < c1 ? x1[r * ld1 + c] : x2[r * ld2 + (c - c1)];
                    const float *wr = wk + c * c_out;
                    for (int64_t j = 0; j < c_out; ++j) tgt[j] = fmaf(xv, wr[j], tgt[j]);
                }
                if (order == 2) for (int64_t j = 0; j < c_out; ++j) acc[j] = acc[j] + part[j];
            }
            if (order == 3)
                for (int64_t j = 0; j < c_out; ++j)
                    acc[j] = ((part[j] + part[c_out + j]) + part[2 * c_out + j]) + part[3 * c_out + j];
            float *orow = out + dst * ldo;
            for (int64_t j = 0; j < c_out; ++j) {
                float v = acc[j];
                if (bias) v = v + bias[j];
                if (act == ORC_ACT_PRELU) v = v < 0.0f ? v * slope : v;
                else if (act == ORC_ACT_RELU) v = v < 0.0f ? 0.0f : v;
                if (clip > 0.0f) v = v < -clip ? -clip : (v > clip ? clip : v);
                orow[j] = v;
            }
        }
        free(acc);
        free(part);
    }
}


/* The logistic function in front of the 16-bit occupancy probability, as the HIP path SPECIFIES it from numerics version 3 on
 * (include/fpcc_hip.h, fpcc_logit_to_prob16; fastpcc_amd/csrc/hip/entropy.hip:sigmoid_spec): exp(-x) by Cody-Waite reduction and a
 * degree-5 polynomial, every operation an IEEE-754 binary32 fused multiply-add, multiplication, addition, round-to-nearest-even or
 * correctly rounded division -- so that ANY conforming host or device produces the same bits and a stream written on one decodes on
 * the other.  (The reference calls torch.sigmoid, geo_lossl_em.py:96-99, whose last bit depends on the libm / vector library at hand;
 * this function agrees with it to within 2 ulp, i.e. the 16-bit probability to within one step.) */
static inline float sigmoid_spec(float x) {
    float t = -x;
    t = fminf(fmaxf(t, -87.0f), 87.0f);
    const float n = rintf(t * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, t);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float pr = p * r;
    float e = fmaf(pr, r, r);
    e = e + 1.0f;
    union { uint32_t u; float f; } scale;
    scale.u = (uint32_t)((int32_t)n + 127) << 23;
    e = e * scale.f;
    const float d = 1.0f + e;
    return 1.0f / d;
}

void orc_sigmoid_spec_f32(const float *x, int64_t n, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = sigmoid_spec(x[i]);
}
