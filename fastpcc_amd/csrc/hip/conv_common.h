// Declarations shared by the fp32 sparse-convolution translation units (conv.hip: wave-autonomous / workgroup-tiled / VALU
// kernels and the C ABI; conv_lds.hip: the LDS-operand kernel of the large maps).
#pragma once
#include "common.h"

namespace fpcc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float *x1; int c1; int ld1;
    const float *x2; int c2; int ld2;
    const int32_t *nbr; int n_off; int64_t nbr_ks; int64_t nbr_os;
    const float *w; const float *bias; int c_out; int groups;
    const int32_t *out_map; int64_t om_os; int64_t om_gs; float *out; int ldo; int64_t n_out;
    int act; const float *slope; float clip;
    const int32_t *row_order;  // tile position -> output row (NULL: identity); see fpcc_conv_row_keys
};

static __device__ float g_zero_row[64];   // 256 bytes of zeros: the source of every absent neighbour (one copy per translation unit)

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

constexpr int kMaxOffsets = 27;   // the MFMA kernels keep the tile's neighbour indices in LDS: [27][rows]

// Grouped evaluation (summation order 3): the K kernel offsets form four fixed contiguous groups [begin(g), begin(g + 1)).
__host__ __device__ __forceinline__ int offset_group_begin(int g, int n_off) { return (g * n_off + 3) / 4; }
__host__ __device__ __forceinline__ int offset_group_of(int k, int n_off) {
    return (k >= offset_group_begin(1, n_off)) + (k >= offset_group_begin(2, n_off)) + (k >= offset_group_begin(3, n_off));
}

// conv_lds.hip: order-3 (folded) evaluation of a multi-offset layer with both operands staged through LDS; `rows_log` selects the
// tile (2, 3 or 4 row blocks of 32 in lockstep).  Returns FPCC_OK, an error, or -1 when the shape is not covered.
int launch_conv_lds(const ConvArgs &a, const float *wp, int row_blocks, int dbg, hipStream_t s);

}  // namespace fpcc
