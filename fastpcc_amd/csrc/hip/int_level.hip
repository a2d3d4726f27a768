// libfpcc_hip -- the integer codec's traversal, one C call per half level (fpcc_int_level_trunk / fpcc_int_level_expand).
//
// A OneScalePredictor level of /root/reference/models/convolutional/lossl_coord_int/model.py:95-213 is a fixed sequence of nine int8
// layers with their fixed-point epilogues, one octree step and a few requantisations.  Issued layer by layer from the host language it
// costs a module call, a cache lookup, three allocations and a 33-argument foreign call per layer -- 370-520 us per level where the
// launches themselves take 150-200 us on the ten coarse levels of a LiDAR sweep (profiles/r05/int_traversal.md).  Here the block's
// layers are a descriptor table filled once (fpcc_int_onescale) and a level is two calls that issue the whole sequence; every layer
// goes through the same entry points as before (fpcc_conv_i8_also, fpcc_epilogue_i32_also, fpcc_octree_children, fpcc_fill_bits_i8),
// so the integers are the same by construction.
#include "common.h"

using namespace fpcc;

namespace {

struct Carver {                                   // hands out 256-byte aligned pieces of the caller's workspace
    char *base;
    int64_t used = 0;
    explicit Carver(void *ws) : base(static_cast<char *>(ws)) {}
    template <typename T> T *take(int64_t count) {
        T *p = base ? reinterpret_cast<T *>(base + used) : nullptr;
        used += align_up(count * (int64_t)sizeof(T), 256);
        return p;
    }
};

inline int ceil16(int v) { return (v + 15) / 16 * 16; }

inline fpcc_requant8 also_of(const fpcc_i8_requant &q, int8_t *out, int ld, int pad) {
    return fpcc_requant8{out, ld, pad, q.requant_mul, q.zero_point, q.shift};
}

// one layer through the body of fpcc_conv_i8_also; `split_ws`: the accumulator of the offset-split form (maps of <= 8192 rows), one per
// layer and all of them cleared by ONE memset at the head of the level call
int run_layer(const fpcc_i8_layer &L, const int8_t *a, int lda, int64_t n_out, const int32_t *nbr, const int32_t *row_order, void *out,
              const int32_t *residual, const int32_t *slope2, const fpcc_requant8 *also, int n_also, void *split_ws, int64_t split_bytes,
              void *stream) {
    return conv_i8_run(a, L.c_in, lda, nbr, L.n_offsets, 1, L.n_offsets, 1, L.w, L.ldw, L.zp_comp, L.bias, L.slope, L.requant_mul,
                       L.zero_point, L.shift, L.out_bits, out, L.c_out, 0, L.c_out, n_out, row_order, residual, residual ? L.c_out : 0, slope2,
                       also, n_also, split_ws, split_bytes, split_ws != nullptr, stream);
}

bool layer_ok(const fpcc_i8_layer &L, int c_in, int c_out, int n_offsets, int out_bits) {
    return L.w && L.requant_mul && L.zero_point && L.c_in == c_in && L.c_out == c_out && L.n_offsets == n_offsets && L.out_bits == out_bits &&
           L.ldw % 16 == 0 && L.ldw >= c_in;
}

}  // namespace

extern "C" int64_t fpcc_int_level_trunk(const fpcc_int_onescale *blk, int64_t n, const int32_t *feat, const int8_t *feat_q8,
                                        const int32_t *nbr27, const int32_t *row_order, int32_t *res_out, int8_t *q_pred, int8_t *q_up,
                                        int ld_up, int32_t *logits, void *ws, int64_t ws_bytes, void *stream) {
    if (!blk || n < 0) return fail_arg("int_level_trunk: null block or n < 0");
    const int C = blk->channels;
    if (C < 16 || C % 16) return fail_arg("int_level_trunk: the channel count must be a positive multiple of 16");
    if (!layer_ok(blk->dec_conv1, C, C, 27, 8) || !layer_ok(blk->dec_conv2, C, C, 27, 32) || !layer_ok(blk->pred_conv, C, C, 27, 8) ||
        !layer_ok(blk->pred_linear, C, blk->pred_linear.c_out, 1, 32) || !blk->dec_slope)
        return fail_arg("int_level_trunk: the block's layer table does not describe a OneScalePredictor of this width");
    Carver cv(ws);
    int8_t *q_in = feat_q8 ? nullptr : cv.take<int8_t>(n * C);
    int8_t *t1 = cv.take<int8_t>(n * C);
    int8_t *t2 = cv.take<int8_t>(n * C);
    const int64_t split_bytes = align_up(fpcc_conv_i8_ws_bytes(1, 27, 1, C, n), 256);
    char *split = split_bytes ? cv.take<char>(3 * split_bytes) : nullptr;      // one accumulator per 3x3x3 layer
    if (!ws) return cv.used > 0 ? cv.used : 16;
    if (ws_bytes < cv.used) { set_error("int_level_trunk: workspace %lld < %lld", (long long)ws_bytes, (long long)cv.used); return FPCC_E_WORKSPACE; }
    if (n == 0) return FPCC_OK;
    if (!feat || !nbr27 || !res_out || !q_pred || !logits) return fail_arg("int_level_trunk: null pointer");
    if (split) FPCC_HIP(hipMemsetAsync(split, 0, (size_t)(3 * split_bytes), as_stream(stream)));
    if (q_up && (ld_up < C || ld_up % 4)) return fail_arg("int_level_trunk: row stride of the second int8 copy");
    if (!feat_q8) {                                                     // dec.input_requant, when the producer of `feat` did not write it
        const fpcc_i8_requant &r = blk->dec_in;
        if (int rc = fpcc_epilogue_i32(feat, C, nullptr, nullptr, r.requant_mul, 0, r.zero_point, r.shift, 8, q_in, C, 0, n, C, nullptr, stream))
            return rc;
        feat_q8 = q_in;
    }
    // dec: conv_prelu, then conv2 with prelu(feat + .) and the int8 copies for `pred` and `upsample` in its epilogue
    if (int rc = run_layer(blk->dec_conv1, feat_q8, C, n, nbr27, row_order, t1, nullptr, nullptr, nullptr, 0, split, split_bytes, stream)) return rc;
    fpcc_requant8 also[2] = {also_of(blk->pred_in, q_pred, C, C), also_of(blk->up_in, q_up, ld_up, C)};
    if (int rc = run_layer(blk->dec_conv2, t1, C, n, nbr27, row_order, res_out, feat, blk->dec_slope, also, q_up ? 2 : 1,
                           split ? split + split_bytes : nullptr, split_bytes, stream))
        return rc;
    // pred: 3x3x3 convolution, linear layer to the 255 logits
    if (int rc = run_layer(blk->pred_conv, q_pred, C, n, nbr27, row_order, t2, nullptr, nullptr, nullptr, 0,
                           split ? split + 2 * split_bytes : nullptr, split_bytes, stream))
        return rc;
    return run_layer(blk->pred_linear, t2, C, n, nullptr, nullptr, logits, nullptr, nullptr, nullptr, 0, nullptr, 0, stream);
}

extern "C" int64_t fpcc_int_level_expand(const fpcc_int_onescale *blk, int64_t n, int64_t m, const int32_t *res, int8_t *q_up, int ld_up,
                                         const int16_t *symbols, const int32_t *coords, const int32_t *nbr27, const int32_t *row_order,
                                         int32_t *child_coords, int32_t *feat_out, const fpcc_i8_requant *next_in, int8_t *feat_q8_out,
                                         void *ws, int64_t ws_bytes, void *stream) {
    if (!blk || n < 0 || m < 0 || m > 8 * n) return fail_arg("int_level_expand: null block or sizes out of range");
    const int C = blk->channels;
    if (C < 16 || C % 16 || !blk->has_upsample) return fail_arg("int_level_expand: the block has no upsampling stage of a supported width");
    if (!layer_ok(blk->up_linear, C + 8, C, 1, 32) || !layer_ok(blk->up_conv1, C, C, 27, 8) || !layer_ok(blk->up_conv2, C, C, 27, 32) ||
        !layer_ok(blk->up_out, C, 8 * C, 1, 32) || !blk->up_slope || !blk->up_linear.slope)
        return fail_arg("int_level_expand: the block's layer table does not describe a OneScalePredictor of this width");
    Carver cv(ws);
    uint8_t *bits = cv.take<uint8_t>(n * 8);
    int32_t *parent_row = cv.take<int32_t>(m);
    int32_t *octant = cv.take<int32_t>(m);
    const int64_t table_rows = (m + 127) / 128 * 128;
    int32_t *table = cv.take<int32_t>(table_rows * 8);
    const int64_t oct_bytes = fpcc_octree_children(nullptr, nullptr, nullptr, n, m, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                                   nullptr, 0, nullptr);
    if (oct_bytes < 0) return oct_bytes;
    void *oct_ws = cv.take<char>(oct_bytes);
    int32_t *u = cv.take<int32_t>(n * C);                                 // upsample[1]: linear + PReLU, Q8.23
    int8_t *u_q = cv.take<int8_t>(n * C);                                 // ... requantised for the residual block
    int8_t *t1 = cv.take<int8_t>(n * C);
    int32_t *v = cv.take<int32_t>(n * C);                                 // the residual block's output
    int8_t *v_q = cv.take<int8_t>(n * C);                                 // ... requantised for the last linear layer
    int32_t *raw = cv.take<int32_t>(m * C);                               // that layer's raw sums, occupied (row, octant) pairs only
    const int64_t split_bytes = align_up(fpcc_conv_i8_ws_bytes(1, 27, 1, C, n), 256);
    char *split = split_bytes ? cv.take<char>(2 * split_bytes) : nullptr;
    if (!ws) return cv.used > 0 ? cv.used : 16;
    if (ws_bytes < cv.used) { set_error("int_level_expand: workspace %lld < %lld", (long long)ws_bytes, (long long)cv.used); return FPCC_E_WORKSPACE; }
    if (n == 0 || m == 0) return FPCC_OK;
    if (!res || !q_up || !symbols || !nbr27 || !feat_out) return fail_arg("int_level_expand: null pointer");
    if (ld_up < ceil16(C + 8) || ld_up % 16) return fail_arg("int_level_expand: the int8 copy needs room for the eight occupancy columns");
    if (feat_q8_out && !next_in) return fail_arg("int_level_expand: an int8 copy needs its requantiser");
    if (split) FPCC_HIP(hipMemsetAsync(split, 0, (size_t)(2 * split_bytes), as_stream(stream)));
    // the level's octree step: occupied (row, octant) pairs, their gather table and -- for the decoder -- the children's coordinates
    if (int64_t rc = fpcc_octree_children(symbols, nullptr, coords, n, m, 1 << 23, coords ? child_coords : nullptr, parent_row, octant, table,
                                          table_rows, bits, nullptr, oct_ws, oct_bytes, stream))
        return rc;
    // requant(cat(R, bits << 23)): the left part came out of the trunk's last epilogue, the eight bit columns are filled here
    if (int rc = fpcc_fill_bits_i8(bits, n, 1 << 23, blk->up_in.requant_mul, blk->up_in.zero_point, blk->up_in.shift, q_up, ld_up, C, ld_up, stream))
        return rc;
    fpcc_requant8 a1 = also_of(blk->up_res_in, u_q, C, C);
    if (int rc = run_layer(blk->up_linear, q_up, ld_up, n, nullptr, nullptr, u, nullptr, nullptr, &a1, 1, nullptr, 0, stream)) return rc;
    if (int rc = run_layer(blk->up_conv1, u_q, C, n, nbr27, row_order, t1, nullptr, nullptr, nullptr, 0, split, split_bytes, stream)) return rc;
    fpcc_requant8 a2 = also_of(blk->up_out_in, v_q, C, C);
    if (int rc = run_layer(blk->up_conv2, t1, C, n, nbr27, row_order, v, u, blk->up_slope, &a2, 1, split ? split + split_bytes : nullptr, split_bytes,
                           stream))
        return rc;
    // the last linear layer C -> 8 C for the occupied octants only: an 8-"offset" gather with one entry per output row, then its epilogue with
    // the octant's bias / multiplier columns (and the next level's first requantiser)
    const fpcc_i8_layer &L = blk->up_out;
    if (int rc = fpcc_conv_i8_also(v_q, C, C, table, 8, 1, 8, 1, L.w, L.ldw, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 32, raw, C, 0, C, m,
                                   nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, stream))
        return rc;
    fpcc_requant8 a3{};
    if (feat_q8_out) a3 = also_of(*next_in, feat_q8_out, C, C);
    return fpcc_epilogue_i32_also(raw, C, L.bias, nullptr, L.requant_mul, 1, L.zero_point, L.shift, 32, feat_out, C, 0, m, C, octant,
                                  feat_q8_out ? &a3 : nullptr, feat_q8_out ? 1 : 0, stream);
}
