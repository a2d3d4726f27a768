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

typedef int i32x4 __attribute__((ext_vector_type(4)));

// A ROW-MAJOR neighbour table (nbr_ks == 1: the n_off entries of an output row lie side by side, e.g. child_row [m][8] or the
// transposed 3x3x3 table [n][32] of fpcc_transpose_table_i32) lets a lane fetch its row's entries as 16-byte pieces of ONE cache line.
// The offset-major table [27][n] costs 27 requests of 4 bytes per row, each to another line as soon as the rows of a block are not
// consecutive (neighbour-pattern row order) -- on a loaded chip that prologue took a wave ~70 K cycles (profiles/r04/prologue_epilogue.md).
// Together with a row order the row-major table is indexed by TILE POSITION: its row p holds the neighbours of output row row_order[p]
// (the caller gathers the table's rows once per coordinate map), so that the entries of a block of consecutive positions are
// consecutive in memory and no load of the prologue depends on another.
__device__ __forceinline__ bool table_is_row_major(const ConvArgs &a) {
    return a.nbr && a.nbr_ks == 1 && (a.nbr_os & 3) == 0 && a.nbr_os >= ((a.n_off + 3) & ~3) && (reinterpret_cast<uintptr_t>(a.nbr) & 15) == 0;
}

// Grouped evaluation (summation order 3): the K kernel offsets form four fixed contiguous groups [begin(g), begin(g + 1)).
__host__ __device__ __forceinline__ int offset_group_begin(int g, int n_off) { return (g * n_off + 3) / 4; }
__host__ __device__ __forceinline__ int offset_group_of(int k, int n_off) {
    return (k >= offset_group_begin(1, n_off)) + (k >= offset_group_begin(2, n_off)) + (k >= offset_group_begin(3, n_off));
}

// Diagnostics: s_memtime stamps of a wave's life in LDS slots, copied to a global buffer when the wave ends (fpcc_conv_debug_stamps).
// A stamp is one LDS store by lane 0 with the exec mask narrowed in place (no branch: a branch in a stage loop makes hipcc drain vmcnt).
constexpr int kStampSlots = 48;
static __device__ unsigned long long *g_stamp_buf = nullptr;      // one copy per translation unit, both set by fpcc_conv_debug_stamps
static __device__ long long g_stamp_cap = 0;
__device__ __forceinline__ void stamp_lds(unsigned long long *slot) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    const unsigned addr = (unsigned)(uintptr_t)slot;
    asm volatile("s_mov_b64 exec, 1\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(t) : "memory");
}
__device__ __forceinline__ void stamp_lds_realtime(unsigned long long *slot) {      // the constant 100 MHz counter: clock = d(memtime) / d(realtime)
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    const unsigned addr = (unsigned)(uintptr_t)slot;
    asm volatile("s_mov_b64 exec, 1\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(t) : "memory");
}
int set_lds_stamp_buffer(unsigned long long *buf, long long cap);   // conv_lds.hip's copy of the two symbols

// conv_lds.hip: order-3 (folded) evaluation of a multi-offset layer with both operands staged through LDS; `rows_log` selects the
// tile (2, 3 or 4 row blocks of 32 in lockstep).  Returns FPCC_OK, an error, or -1 when the shape is not covered.
int launch_conv_lds(const ConvArgs &a, const float *wp, int row_blocks, int dbg, hipStream_t s);

}  // namespace fpcc
