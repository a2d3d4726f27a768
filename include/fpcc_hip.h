/*
 * fpcc_hip.h -- C ABI of libfpcc_hip.so: hand-written HIP kernels for gfx950 (MI355X) behind the FastPCC
 * encode/decode hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the caller (e.g. PyTorch's caching allocator)
 *     owns all memory, nothing is allocated or freed inside;
 *   - `stream` is a hipStream_t passed as void*; every function only enqueues work on it (no synchronisation), so the
 *     calls can be captured into a hipGraph;
 *   - return value: 0 on success, a negative fpcc_status otherwise; fpcc_last_error() gives the message of the last
 *     failure on the calling thread;
 *   - functions that need scratch memory take (ws, ws_bytes); called with ws == NULL they return the number of bytes
 *     they need for that problem size (a positive value) and enqueue nothing;
 *   - variable-size results are written into caller buffers sized for the worst case, and their length is left in a
 *     device counter the caller reads back when it needs it (one D2H copy may serve several calls).
 *
 * Which reference interface each entry point replaces is cited per function (paths relative to /root/reference).
 * MinkowskiEngine itself is an un-vendored dependency of the reference; for the float path the citations are the
 * reference's call sites into it.
 */
#ifndef FPCC_HIP_H_
#define FPCC_HIP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FPCC_OK = 0,
    FPCC_E_ARG = -1,       /* invalid argument / unsupported shape */
    FPCC_E_HIP = -2,       /* a HIP runtime call failed */
    FPCC_E_WORKSPACE = -3  /* workspace too small */
} fpcc_status;

const char *fpcc_last_error(void);
/* Number of visible HIP devices (0 without a GPU); never initialises a context beyond hipGetDeviceCount. */
int fpcc_device_count(void);
/* Diagnostic, no reference counterpart: one wave spins for spin_us (1 .. 10000) microseconds of the constant 100 MHz counter and
 * writes out2[0] = shader-clock cycles that went by, out2[1] = 100 MHz ticks; cycles / ticks * 100 = the shader clock in MHz that
 * the power management grants at that point of the stream (tools/gap_probe.py). */
int fpcc_clock_probe(int64_t *out2, int spin_us, void *stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Coordinates.  A coordinate set at pyramid level l is a sorted array of unique 64-bit keys                      */
/*      key = batch << (3*bits) | morton3(x >> l, y >> l, z >> l)       (x is Morton bit 0)                        */
/* so that the parent of a key is key >> 3 and its octant (kernel index of a 2x2x2 stride-2 kernel, x fastest) is */
/* key & 7.  `bits` is the number of bits per axis AT THAT LEVEL.                                                 */
/* ------------------------------------------------------------------------------------------------------------ */

/* Morton key of int32 coordinates; replaces space_filling_curves_ext.morton3d_encode_magicbits
 * (lib/space_filling_curves/src/morton3d.cu:19-72, binding.cu:17-23).  coords: [n, row_stride] int32, the three axes at
 * columns col_bit0, col_bit1, col_bit2 (the column whose bits land on Morton bit 0, 1, 2: axis order 'xyz' of a [n,3]
 * array is 0,1,2; 'zyx' is 2,1,0).  Each coordinate is taken modulo 2^21 as in the reference. */
int fpcc_morton3d_encode(const int32_t *coords, int64_t n, int64_t row_stride, int col_bit0, int col_bit1, int col_bit2,
                         int64_t *keys_out, void *stream);

/* Hilbert key of int32 coordinates; replaces space_filling_curves_ext.hilbert3d_encode_lut
 * (lib/space_filling_curves/src/hilbert3d.cu:28-60, binding.cu:19-23): `bits` in [1, 21] bits per axis, most significant
 * first; col_x / col_y / col_z are the columns that play x, y, z (axis_order 'xyz' of a [n,3] array is 0,1,2).  The
 * 12-state machine is generated from the curve's geometry and uploaded on first use (synchronous, once per process). */
int fpcc_hilbert3d_encode(const int32_t *coords, int64_t n, int64_t row_stride, int col_x, int col_y, int col_z, int bits,
                          int64_t *keys_out, void *stream);

/* Level-l keys of batched coordinates [n,4] = (batch, x, y, z): what ME.SparseTensor(coordinates=...) hashes
 * (models/convolutional/lossy_coord_v2/model.py:147-153). */
int fpcc_keys_from_coords(const int32_t *coords, int64_t n, int level, int bits, int64_t *keys_out, void *stream);
/* Inverse: coordinates (batch, x, y, z) in absolute units (multiples of 1 << level), optionally + offset_xyz[3] (device). */
int fpcc_coords_from_keys(const int64_t *keys, int64_t n, int level, int bits, const int32_t *offset_xyz,
                          int32_t *coords_out, void *stream);

/* Stable ascending radix sort of (key, original row) pairs; replaces torch.argsort(morton) (lossy_coord_v2/model.py:142).
 * perm_out[i] = original row of the i-th smallest key.  end_bit: number of significant key bits. */
int64_t fpcc_sort_keys(const int64_t *keys_in, int64_t n, int end_bit, int64_t *keys_out, int32_t *perm_out,
                       void *ws, int64_t ws_bytes, void *stream);

/* keys sorted -> unique keys (first occurrence kept).  first_out[u] = index of the first row of unique key u;
 * count_out (device int32[1]) = number of unique keys.  Replaces the de-duplication ME.SparseTensor performs. */
int64_t fpcc_unique_keys(const int64_t *keys, int64_t n, int64_t *ukeys_out, int32_t *first_out, int32_t *count_out,
                         void *ws, int64_t ws_bytes, void *stream);

/* One pyramid step (stride-2 coordinate map, what cm.stride(key, 2) / a stride-2 MinkowskiConvolution creates,
 * lib/minkowski_sparse_conv_layers.py:114-127): from sorted unique keys[n] produce
 *   parent_of[n]          parent row of every row,
 *   pkeys[<=n]            sorted unique parent keys,
 *   child_row[<=n][8]     row of child (parent, octant) or -1,
 *   count_out             device int32[1], number of parents. */
int64_t fpcc_coarsen(const int64_t *keys, int64_t n, int32_t *parent_of, int64_t *pkeys, int32_t *child_row,
                     int32_t *count_out, void *ws, int64_t ws_bytes, void *stream);
/* One level of the integer codec's ENCODER-side octree analysis (replaces get_bin: out_coords = unique_consecutive(coords >> 1) and the
 * (2, 2, 2) / stride-2 "fold" convolution over a hash-table kernel map, models/convolutional/lossl_coord_int/model.py:262-295 -- about
 * 21 device operations and one synchronisation per level there): from the sorted, duplicate-free keys[n] of a level -- Morton code
 * with z on bit 0, so that key & 7 = 4 dx + 2 dy + dz is the child's kernel offset, the sample index above the code -- the next coarser
 * level with m rows (known from fpcc_level_histogram[_clouds]): pkeys[m] = keys >> 3 (sample index from bit `batch_shift` up),
 * coords int32 [m][4] = (sample, x, y, z), bits int32 [m][8] (0 | 1 per child: the fold convolution's output), table int32
 * [table_rows >= m][8] (child row + 1, 0 = none, padding rows zeroed: the kernel map fpcc_conv_i8 reads for that convolution),
 * symbols int16 [m] (occupancy byte - 1, bit 7 - k = child k, model.py:60).  Any output may be NULL.  ws == NULL returns the
 * workspace size. */
int64_t fpcc_octree_level(const int64_t *keys, int64_t n, int64_t m, int batch_shift, int64_t *pkeys, int32_t *coords, int32_t *bits,
                          int32_t *table, int64_t table_rows, int16_t *symbols, void *ws, int64_t ws_bytes, void *stream);
/* Row counts of the `levels` (1..21) next coarser levels of a sorted, duplicate-free key array in ONE pass, so that a whole pyramid
 * costs one count read-back instead of one per level: hist[t], t = 0..levels (int32, zeroed here), = number of neighbouring key pairs
 * whose highest differing bit is in [3 t, 3 t + 3) (t clamped to `levels`); the map `l` levels coarser has
 * 1 + sum_{t >= l} hist[t] rows (n > 0).  No reference counterpart (MinkowskiEngine builds strided maps one at a time). */
int64_t fpcc_level_histogram(const int64_t *keys, int64_t n, int levels, int32_t *hist, void *stream);
/* The same pass for a batch of clouds coded in ONE network traversal (the reference's lists of independent clouds:
 * models/convolutional/lossy_coord_v2/model.py:247-256,277-288 compress_partitions / decompress_partitions): keys carry the cloud
 * index above bit `cloud_shift`, rows are cloud-major.  hist[c][t] (int32 [n_clouds][levels + 2], zeroed here): for t <= levels the
 * neighbouring key pairs INSIDE cloud c whose highest differing bit is in [3 t, 3 t + 3) (clamped as above), for t = levels + 1 the
 * keys of cloud c; cloud c has (hist[c][levels + 1] > 0) + sum_{t >= l} hist[c][t] rows on the map `l` levels coarser. */
int64_t fpcc_level_histogram_clouds(const int64_t *keys, int64_t n, int levels, int cloud_shift, int n_clouds, int32_t *hist,
                                    void *stream);

/* Occupancy-driven refinement (decoder side; replaces coords = pred.C[mask]; cm.insert_and_map(...),
 * models/convolutional/lossy_coord_lossy_color/geo_lossl_em.py:278-283, and MinkowskiPruning on a generated set):
 * candidates are the 8 children of each of m parents in (parent, octant) order; mask[8m] selects the kept ones.
 *   keys_out[<=8m], parent_of[<=8m], child_row[m][8], count_out device int32[1]. */
int64_t fpcc_refine(const int64_t *pkeys, int64_t m, const uint8_t *mask, int64_t *keys_out, int32_t *parent_of,
                    int32_t *child_row, int32_t *count_out, void *ws, int64_t ws_bytes, void *stream);

/* 3x3x3 kernel map (what cm.kernel_map(key, key, kernel_size=3) answers; offsets enumerated x fastest, centred):
 * nbr[d][i] = row of the voxel at offset d from voxel i, or -1.  nbr is [27][n] int32.
 *   _search      : binary search in the sorted keys (any level, no parent needed);
 *   _from_parent : derived from the parent level's table and the child_row table (two dependent loads per entry, no
 *                  hashing, no key arithmetic); parent_nbr is [27][m].
 *   child_row == NULL means "all 8 children exist, row = 8*parent + octant" (a generated set). */
int fpcc_nbr27_search(const int64_t *keys, int64_t n, int bits, int32_t *nbr, void *stream);
int fpcc_nbr27_from_parent(const int64_t *keys, const int32_t *parent_of, int64_t n, const int32_t *parent_nbr,
                           int64_t m, const int32_t *child_row, int32_t *nbr, void *stream);
/* The same pass with up to two more results from the entries it has in hand (round 6; each may be NULL): rows_out int32 [n][32],
 * the table once more ROW-MAJOR (entries 27 .. 31 = -1; 16-byte aligned) -- the layout the MFMA kernels' prologue reads, which
 * rounds 2-5 made by a transposition pass over the finished table --, and masks_out uint32 [n], bit d set where neighbour d of the
 * row exists -- what fpcc_conv_row_keys_masks and fpcc_conv_tile_keys need. */
int fpcc_nbr27_from_parent_ex(const int64_t *keys, const int32_t *parent_of, int64_t n, const int32_t *parent_nbr, int64_t m,
                              const int32_t *child_row, int32_t *nbr, int32_t *rows_out, uint32_t *masks_out, void *stream);
/* the same derivation, keeping only WHICH of the 27 neighbours exist: masks_out[i] bit d = neighbour d of row i exists */
int fpcc_mask27_from_parent(const int64_t *keys, const int32_t *parent_of, int64_t n, const int32_t *parent_nbr, int64_t m,
                            const int32_t *child_row, uint32_t *masks_out, void *stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Sparse convolution, fp32, output-stationary gather -> MFMA GEMM (no scatter, no atomics, bitwise reproducible) */
/* ------------------------------------------------------------------------------------------------------------ */
enum { FPCC_ACT_NONE = 0, FPCC_ACT_PRELU = 1, FPCC_ACT_RELU = 2 };

/* Numerics version.  A decoder must recompute the encoder's fp32 activations BIT FOR BIT (they pass through round() and
 * 16-bit probability quantisation before entropy coding), so the summation order of every layer is part of the stream
 * format.  Since version 2 the order of a layer is a function of its SHAPE alone (channel counts, kernel offsets, groups) -- plus one
 * row threshold for the zero-padded shapes -- never of tile shapes, row order, kernel choice or a tuning knob:
 *     order 3 (grouped)                      multi-offset layers (8 <= K <= 27, one group) with C_in, c1 multiples of 32 and C_out
 *                                            in {32,64,128}: the K offsets form four fixed contiguous groups
 *                                            [ceil(g K / 4), ceil((g + 1) K / 4)); each group is an order-1 chain from zero over
 *                                            the offsets present, the partial sums are added as ((g0 + g1) + g2) + g3, then bias
 *     order 1 (MFMA chain)                   every other MFMA shape: one chain, offsets ascending, 0,4,1,5,2,6,3,7 inside
 *                                            aligned groups of 8 channels
 *     order 0 (natural chain)                shapes outside the MFMA path
 *     zero-padding to an MFMA shape          per-point / 3x3x3 layers on maps of at least FPCC_PAD_MIN_ROWS rows
 *     two-phase conv3 -> 1 channel (order 2) every 3x3x3 layer with one output channel and C_in % 16 == 0: per offset its own
 *                                            chain, offsets' sums added in ascending order
 *     logistic function (version 3)          the sigmoid in front of every 16-bit occupancy probability (fpcc_logit_to_prob16) is a
 *                                            SPECIFIED sequence of IEEE-754 binary32 operations (Cody-Waite reduction, degree-5
 *                                            polynomial, correctly rounded division; entropy.hip:sigmoid_spec), not the device's expf:
 *                                            any conforming host or device reproduces its bits, so a stream written on one decodes on
 *                                            the other; within one step of the 16-bit probability torch.sigmoid gives the reference
 * History: version 1 evaluated multi-offset maps of <= 8192 rows offset-split (order 2) and larger ones in order 1; version 2
 * introduced order 3 but took its probabilities from the device's expf.  A decoder decodes streams of ITS OWN version only: streams
 * of other versions are refused where the version byte is present (`numerics_version_in_header` of the model configs, which
 * fastpcc_amd/run_test.py switches on; the library default keeps the reference's byte layout) and decode to garbage where it is not
 * ("Stream compatibility").  tests/golden/v2_stream.json holds a stream of this version that every later build must
 * decode; v2_stream_numerics1.json / v2_stream_numerics2.json hold streams of the earlier versions that it must refuse. */
#define FPCC_NUMERICS_VERSION 3
#define FPCC_PAD_MIN_ROWS 8192
int fpcc_numerics_version(void);

/*
 * out[dst(o,g), :] = act( sum_k  X[nbr[k][o], :] @ W[g][k]  + bias )        o < n_out, g < groups
 *
 * Replaces MinkowskiConvolution / MinkowskiConvolutionTranspose / MinkowskiGenerativeConvolutionTranspose /
 * MinkowskiLinear (+ the bias add, MinkowskiPReLU/ReLU and ME.cat that surround them) as used through
 * lib/minkowski_sparse_conv_layers.py:31-159 and models/convolutional/lossy_coord_v2/layers.py:263-271,306-315.
 *
 *   X        rows are the channel-concatenation of x1 [*, c1] (row stride ld1 floats) and, if x2 != NULL, x2 [*, c2];
 *   nbr      int32 input row per (offset k, output row o) at nbr[k*nbr_ks + o*nbr_os], -1 = absent; NULL = identity
 *            (n_offsets == 1).  A [27][n] table has (ks, os) = (n, 1); a child_row table [m][8] read as a stride-2
 *            2x2x2 convolution has (1, 8);
 *   w        [groups][n_offsets][c1 + c2][c_out] fp32 (MinkowskiEngine kernel layout [K, C_in, C_out] per group);
 *   bias     [c_out] or NULL;  slope: device float[1] for PReLU;  clip > 0 clamps the result to [-clip, clip];
 *   dst      out_map == NULL: row o*groups + g;  else out_map[o*om_os + g*om_gs] (negative: skipped);
 *   out      row stride ldo floats.
 * groups > 1 expresses a stride-2 transposed / generative convolution: group g = octant g of parent row o, with
 * out_map = child_row [m][8] ((om_os, om_gs) = (8, 1)) or NULL for the full generated set.
 *
 * Summation order (fixed, documented for bit-exact checking): see "Numerics version" above; fpcc_conv_f32_order_ex() reports
 * the order of a shape (0 natural chain, 1 MFMA chain, 3 grouped).  A grouped shape runs on the wave kernel and needs packed
 * weights: from the caller (fpcc_conv_f32_pk), or packed per call into a workspace of fpcc_conv_f32_ws_bytes() bytes.
 */
int fpcc_conv_f32(const float *x1, int c1, int ld1, const float *x2, int c2, int ld2,
                  const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os,
                  const float *w, const float *bias, int c_out, int groups,
                  const int32_t *out_map, int64_t om_os, int64_t om_gs, float *out, int ldo, int64_t n_out,
                  int act, const float *slope, float clip, const int32_t *row_order, void *ws, int64_t ws_bytes,
                  void *stream);
/* The same operator with an additional PACKED copy of the weights (NULL: exactly fpcc_conv_f32).  Shapes for which
 * fpcc_conv_packed_floats() != 0 (C_in, c1 multiples of 32; C_out in {32, 64, 128}) then run on the wave-autonomous MFMA
 * kernel: every wave owns 32 (or 64) output rows x 1, 2 or 4 column blocks and reads its B operands straight from the packed
 * weights (L2-resident) as coalesced 16-byte loads -- no LDS staging; multi-offset shapes additionally split their offsets over
 * the four waves of a workgroup (order 3).  Same results bit for bit with and without the packed copy.
 * fpcc_conv_pack_weights_f32: w [n_mats][c_in][c_out] (n_mats = groups * n_offsets) ->
 *     w_packed[m][cc][g8][nb][h][i][j] = w[m][32 cc + 8 g8 + 4 h + j][32 nb + i]      (same number of floats);
 * the caller caches it next to the weights (fastpcc_amd/hipops.py keeps one per weight tensor). */
int64_t fpcc_conv_packed_floats(int c1, int c2, int c_out, int n_offsets, int groups);
/* Tuning knobs of the wave kernel (process-wide; initial values from FPCC_CONV_WAVE / FPCC_WAVE_NBW / FPCC_WAVE_SB).  None
 * changes a result -- the tests run every setting against the same oracle output.  Returns the previous value.
 *   0  use the wave kernel when packed weights are given (1) or the workgroup-tiled kernel (0)
 *   1  column blocks per wave: 0 = by map size, else 1 | 2 | 4
 *   2  1 = the next stage's address arithmetic may be scheduled between the MFMAs, 0 = nothing crosses the load points
 *   3  timing experiments only (results are WRONG): 1 = no gather traffic, 2 = weights from one chunk, 3 = both; 0 = off;
 *      16 = stage stamps of the grouped kernel (results exact, see fpcc_conv_debug_stamps)
 *   5  row tile of the workgroup-tiled kernel: 0 = by map size, 1 | 2 | 3 = 128 | 64 | 32 rows
 *   6  rows from which per-point layers (one offset, identity map) run on the persistent kernel that keeps the weights in
 *      registers (FPCC_POINTWISE_MIN_ROWS, default 32768); 0 = never
 *   8  column blocks per workgroup of the grouped evaluation: 0 = by map size, else 1 | 2 (FPCC_GROUPED_NBW)
 *   9  rows from which order-1 multi-offset layers use 64 x 64 wave tiles (FPCC_WAVE22_MIN_ROWS; 0 = never, the default)
 *   7  (NOT result-neutral, refused unless FPCC_EXPERIMENT=1) 1 = evaluate grouped shapes in order 1 instead: A/B experiments
 *  10  rows from which order-3 layers with 64 or 128 output channels take BOTH MFMA operands through LDS (k_conv_lds, conv_lds.hip:
 *      LDS-DMA staging, 2-4 row blocks of a workgroup in lockstep; FPCC_LDS_MIN_ROWS; 0 = never).  Same order 3, same bits.
 *  11  row blocks per workgroup of that kernel: 2 | 3 | 4 (FPCC_LDS_ROW_BLOCKS, default 2)
 *  12  persistent form of the grouped / folded kernels: workgroups per CU that walk the units of a launch with a stride and fetch the
 *      next unit's table rows while computing (row-major tables only; FPCC_CONV_PERSIST; default 0 = one unit per workgroup / wave -- measured no faster,
 *      profiles/r04/persistent.md;
 *      values above 16 = that many workgroups in all)
 *   4  rows from which the grouped evaluation runs FOLDED -- one wave per unit adds up the four offset groups itself -- instead of on
 *      four waves per unit (FPCC_GROUPED_FOLD_ROWS, default 102400; 0 = never).  Same order 3, same bits. */
int fpcc_conv_set_tuning(int which, int value);
/* Diagnostics (profiles/r04/small_level_stage.md): with knob 3 set to 16 every wave of the grouped kernel leaves 48 64-bit words in
 * `buf` (device memory, n_u64 words; NULL / 0 detaches): s_memtime at kernel entry [0], after the neighbour-table read [1], after the
 * first operands were requested [2], at the top of stage s [3 + min(s, 36)], after the last stage [40], after the partial-sum
 * exchange [41], after the stores [42]; [44] stages, [45] workgroup << 8 | wave, [46] HW_ID, [47] XCC_ID.  Results stay exact. */
int fpcc_conv_debug_stamps(unsigned long long *buf, int64_t n_u64);
/* Timing a launch from a process whose launches are paced by several host threads: the NEXT call of fpcc_conv_f32 / fpcc_conv_f32_pk /
 * fpcc_mlp_chain_f32 / fpcc_pointwise_head_f32 by the calling thread records `start_event` (a hipEvent_t) on its stream before its
 * first launch and `end_event` after its last one, inside that one call -- recorded from the host language around the call, anything the
 * host does between the record and the launch (an interpreter switching threads) is timed as part of the kernel.  NULL, NULL cancels. */
int fpcc_time_next_launch(void *start_event, void *end_event);
/* The same diagnostics for the int8 tiled convolution (profiles/r05/int8_stage.md): while a buffer is attached, fpcc_conv_i8* launches
 * of the 128-column tiled kernel run its stamped build (results exact) and wave 0 of every workgroup leaves 48 64-bit words:
 * s_memtime at entry [0], kernel-map slice in LDS [1], offsets of the tile known [2], first W tile in LDS [3], top of stage s
 * [4 + min(s, 32)], after the last stage [38], after the epilogue's stores [39]; [44] stages, [45] blockIdx.x << 8 | blockIdx.y,
 * [46] HW_ID.  NULL / 0 detaches.  Blocks (hipMemcpyToSymbol). */
int fpcc_conv_i8_debug_stamps(unsigned long long *buf, int64_t n_u64);

/* 3x3x3 convolution of the constant-one one-channel input the codec starts from (model.py:132-136: features = 1 for every voxel):
 * out[o][j] = act(sum over existing neighbours k of w[k][j] + bias[j]), read from the rows' 27-bit presence masks
 * (fpcc_mask27_from_parent / fpcc_conv_row_keys) -- the chain of fpcc_conv_f32 with x = 1, bit for bit, without the 27-entry
 * neighbour rows.  4 <= c_out <= 32, multiple of 4. */
int fpcc_conv_ones_k3_f32(const uint32_t *masks, int64_t n, const float *w, const float *bias, int c_out, int act,
                          const float *slope, float clip, float *out, int ldo, void *stream);
int fpcc_conv_pack_weights_f32(const float *w, int64_t n_mats, int c_in, int c_out, float *w_packed, void *stream);
int fpcc_conv_f32_pk(const float *x1, int c1, int ld1, const float *x2, int c2, int ld2,
                     const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os,
                     const float *w, const float *w_packed, const float *bias, int c_out, int groups,
                     const int32_t *out_map, int64_t om_os, int64_t om_gs, float *out, int ldo, int64_t n_out,
                     int act, const float *slope, float clip, const int32_t *row_order, void *ws, int64_t ws_bytes,
                     void *stream);
/* A chain of per-point layers as ONE launch: MinkowskiLinear (+ bias + PReLU / ReLU + clamp) stacks with at most one channel
 * concatenation in the middle -- SubDecoderGeoLossl / SubDecoderGeoLossl2 of the lossless coder
 * (models/convolutional/lossy_coord_v2/layers.py:294-331: residual_decoder = MLP(1 -> C/2), MLP(C/2 -> C); decoder =
 * MLP(cat(., y) -> C), MLP(C -> C)), which the reference evaluates as 4 x (linear, bias, activation) MinkowskiEngine calls.
 *     h_0 = x [n, cx]                                  cx = 1, or a multiple of 32 (<= 256)
 *     h_{l+1} = act_l(cat_l(h_l) @ W_l + bias_l)       cat_l(h) = [h, y] for l == cat_layer (>= 1; -1: none), else h
 *     out = h_{n_layers} [n, c_out of the last layer]
 * Layer widths c_out in {32, 64, 128}; y has cy in {32, 64, 96, 128} channels.  Weights: layer l's [c_in][c_out] matrix
 * (MinkowskiLinear's weight transposed), plain in `w` (required for a one-channel first layer) and packed by
 * fpcc_conv_pack_weights_f32(w, 1, c_in, c_out) in `w_packed` (required for every other layer).  Intermediate activations
 * stay in LDS.  Every output element is the FMA chain of the corresponding fpcc_conv_f32 launches (summation order 1), so a
 * fused chain and the layer-by-layer evaluation give the same bits. */
#define FPCC_MLP_CHAIN_MAX_LAYERS 4
typedef struct {
    const float *w;          /* [c_in][c_out] */
    const float *w_packed;   /* fpcc_conv_pack_weights_f32 layout, or NULL for a one-channel first layer */
    const float *bias;       /* [c_out] or NULL */
    const float *slope;      /* device float[1], PReLU only */
    int c_out;
    int act;                 /* FPCC_ACT_* */
    float clip;              /* > 0: clamp to [-clip, clip] */
} fpcc_mlp_layer;
typedef struct {
    const float *x; int cx; int ldx;
    const float *y; int cy; int ldy;       /* concatenated after the activations entering layer `cat_layer` */
    int cat_layer;                          /* -1: no concatenation */
    int n_layers;
    fpcc_mlp_layer layers[FPCC_MLP_CHAIN_MAX_LAYERS];
    float *out; int ldo;
    int64_t n;
} fpcc_mlp_chain;
int fpcc_mlp_chain_f32(const fpcc_mlp_chain *chain, void *stream);
/* Tuning (result-neutral): 1 = the chain shapes of the codecs (1->64->128 ++128 ->128->128, 128->128->128, 64->64->64) run in the
 * workgroup form that keeps the weights in registers (default), 0 = every chain in the wave form; negative = query.  Returns
 * the previous setting. */
int fpcc_mlp_chain_set_form(int form);
/* Narrow per-point head C0 -> C1 -> 1 (the decoder's classify block, models/convolutional/lossy_coord_v2/layers.py:105-110:
 * ConvBlock(16, 8, 1, 1) + ConvBlock(8, 1, 1, 1) on the 8 N candidates) as one launch, one thread per row:
 *     out[o] = act2(sum_j act1(sum_c x[o][c] w1[c][j] + b1[j]) w2[j] + b2), clamped to [-clip, clip] when clip > 0.
 * Shapes (C0, C1) in {(16, 8), (8, 4)}.  order1 = summation order the SEPARATE evaluation of the hidden layer would use for
 * this row count (1: zero-padded MFMA chain on maps of >= FPCC_PAD_MIN_ROWS rows, 0: natural chain below); the output layer is a
 * natural chain.  Same bits as the two fpcc_conv_f32 launches. */
int fpcc_pointwise_head_f32(const float *x, int c0, int ldx, const float *w1, const float *b1, int c1, int act1,
                            const float *slope1, int order1, const float *w2, const float *b2, int act2, const float *slope2,
                            float clip, float *out, int64_t n, void *stream);
/* row_order (MFMA path only, NULL = natural): a permutation of [0, n_out); tile position p computes output row
 * row_order[p].  It changes which rows share a 32-row MFMA block -- and with it how many (block, offset) products are
 * executed -- never a result.  fpcc_conv_row_keys writes, per row, a sort key (window of 2^window_log2 consecutive rows
 * in the high word, Gray rank of the row's neighbour-presence pattern in the low word); sorting them with
 * fpcc_sort_keys (end_bit 63) yields a row_order with like patterns adjacent: on voxelised surfaces ~15 instead of ~24
 * of the 27 offsets per block.  The reference has no counterpart (MinkowskiEngine scatters per offset).
 * Table layout beside a row order: an offset-major table (nbr_ks != 1) is indexed by OUTPUT ROW as always; a ROW-MAJOR table
 * (nbr_ks == 1, fpcc_transpose_table_i32) is indexed by TILE POSITION -- its row p holds the neighbours of output row row_order[p]
 * (gather the table's rows by row_order once per coordinate map) -- so that the entries a block of positions needs are consecutive
 * in memory and no load of a kernel's prologue depends on another. */
int fpcc_conv_row_keys(const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os, int64_t n, int window_log2,
                       int64_t *keys_out, uint32_t *masks_out, void *stream);
/* masks_out (may be NULL): per row, bit k set iff the row has kernel offset k. */
/* The same keys from presence masks that exist already (fpcc_nbr27_from_parent_ex / fpcc_mask27_from_parent write them):
 * 4 bytes read per row instead of the row's 27 table entries. */
int fpcc_conv_row_keys_masks(const uint32_t *masks, int64_t n, int window_log2, int64_t *keys_out, void *stream);
/* out[p][0..ld) = rows[order[p]][0..ld): the rows of a row-major table (ld a multiple of 4, 16-byte aligned) put into position
 * order as whole 16-byte pieces -- the copy the paragraph above asks for, once per coordinate map. */
int fpcc_gather_table_rows_i32(const int32_t *rows, int ld, const int32_t *order, int64_t n, int32_t *out, void *stream);
/* Heaviest tiles first.  fpcc_conv_tile_keys writes, per group of `group` (<= 64) consecutive positions of row_order
 * (NULL = identity), from the row masks of fpcc_conv_row_keys, the key (kernel offsets the group's rows lack) << 32 | group index; sorting the keys (fpcc_sort_keys) gives a
 * group permutation with the groups that execute most offsets first, and fpcc_conv_regroup_rows applies it:
 *     out[g * group + r] = row_order[group_perm[g] * group + r]      (the ragged tail of n % group rows stays last).
 * fpcc_conv_f32 takes the tiles of a row order in dispatch order, so with `group` = the tile height (64 rows from 32 Ki rows
 * up, 32 below) the launch ends with its lightest tiles instead of whichever came last. */
int fpcc_conv_tile_keys(const uint32_t *row_masks, int n_offsets, const int32_t *row_order, int64_t n, int group,
                        int64_t *keys_out, void *stream);
/* perm_out[i] = the group with the i-th smallest key of fpcc_conv_tile_keys (n_groups <= 16384): a stable counting sort by the
 * keys' upper halves in one launch -- same permutation as fpcc_sort_keys on these keys. */
int fpcc_conv_group_order(const int64_t *group_keys, int64_t n_groups, int32_t *perm_out, void *stream);
int fpcc_conv_regroup_rows(const int32_t *row_order, const int32_t *group_perm, int64_t n, int group, int32_t *row_order_out,
                           void *stream);
/* Workspace the shape needs (0 for most).  Multi-offset convolutions (8 <= n_offsets <= 27, groups == 1, C_out in
 * {32, 64, 128}) on maps of at most 8192 rows are evaluated offset-split: one workgroup per (row tile, kernel offset)
 * writes raw partial sums to ws[n_offsets][n_out][c_out], a second kernel reduces them (order 2 below).  Small pyramid
 * levels otherwise run as a few workgroups walking a long serial chain of stages.  The caller owns ws (16-byte aligned). */
int64_t fpcc_conv_f32_ws_bytes(int c1, int c2, int c_out, int n_offsets, int groups, int64_t n_out);
int fpcc_conv_f32_order(int c1, int c2, int c_out);
/* Same, including the launch-dependent choice: the offset-split shapes of fpcc_conv_f32_ws_bytes have order 2
 * (per-offset FMA chains from zero in the MFMA channel order, partial sums added in ascending offset order, then the
 * bias).
 * INVARIANT for callers that batch independent clouds in one launch: since numerics version 2 the order is a function of the
 * layer's SHAPE alone (n_out is accepted and ignored); the one rule above this interface that does look at a row count
 * (FPCC_PAD_MIN_ROWS, zero-padding of narrow shapes) must be applied to the rows of the cloud a row belongs to, not of the launch:
 * an encoder and a decoder may batch different sets of clouds together and must still agree bit for bit. */
int fpcc_conv_f32_order_ex(int c1, int c2, int c_out, int n_offsets, int groups, int64_t n_out);

/* Weight gradient of the same operator (training; MinkowskiEngine's autograd backward, reached from
 * lib/minkowski_sparse_conv_layers.py:85-91 under train.py:262-270):
 *     dw[g][k][ci][co] (+)= sum_o x[in(k,o)][ci] * dy[dst(o,g)][co]
 * with in(k,o) = nbr[k*nbr_ks + o*nbr_os] (NULL: o) and dst(o,g) = out_map[o*om_os + g*om_gs] (NULL: o*groups + g), rows
 * with a negative entry skipped; o < n.  x is ONE matrix [*, c_in] (concatenate two sources first, or call twice on
 * the two halves of dw).  ws: fpcc_conv_wgrad_ws_bytes() bytes of partial sums per row split, reduced in ascending split
 * order (the association is fixed by the shape alone, so results are reproducible).  accumulate != 0 adds to dw.
 * The input gradient is fpcc_conv_f32 itself on the mirrored row maps with transposed weights (fastpcc_amd/autograd.py). */
int64_t fpcc_conv_wgrad_ws_bytes(int c_in, int c_out, int n_offsets, int groups, int64_t n);
int fpcc_conv_wgrad_f32(const float *x, int c_in, int ldx, const float *dy, int c_out, int ldy,
                        const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os,
                        const int32_t *out_map, int64_t om_os, int64_t om_gs, int groups, int64_t n,
                        const int32_t *row_order, float *dw, int accumulate, void *ws, int64_t ws_bytes, void *stream);
/* Kernel of the input-gradient convolution: wt[k][co][ci] = w[flip ? n_offsets-1-k : k][ci][co] (fastpcc_amd/autograd.py). */
int fpcc_transpose_weights_f32(const float *w, int n_offsets, int c_in, int c_out, int flip, float *wt, void *stream);
/* row_order (NULL = none): the neighbour-pattern permutation of fpcc_conv_row_keys for this nbr table.  Multi-offset maps are
 * then reduced in that order, in blocks of 32 rows, and a block none of whose rows has the offset is skipped (on surfaces
 * half of the (row, offset) pairs do not exist); row splits are interleaved.  Same sums in another -- still fixed -- order. */

/* Backward of fpcc_conv_f32's fused epilogue y = act(pre + bias) from the layer OUTPUT y (act: none | ReLU | PReLU with one
 * slope > 0, for which pre < 0 <=> y < 0 and pre = y / slope there):
 *     g = dy * act'(pre),   dbias[c] = sum_rows g[., c] (NULL: skipped),   dslope[0] = sum dy * pre over pre < 0 (NULL: skipped).
 * g feeds the input / weight gradients above.  Replaces the bias-add and MinkowskiPReLU autograd nodes of
 * lib/minkowski_sparse_conv_layers.py:85-91.  ws: fpcc_epilogue_bwd_ws_bytes() bytes of per-row-block partial sums,
 * reduced in ascending block order (reproducible).  g may alias dy. */
int64_t fpcc_epilogue_bwd_ws_bytes(int64_t n, int c);
int fpcc_epilogue_bwd_f32(const float *y, int ldy, const float *dy, int lddy, int64_t n, int c, int act, const float *slope,
                          float *g, int ldg, float *dbias, float *dslope, void *ws, int64_t ws_bytes, void *stream);

/* out[o] = act( sum_k y[nbr[k*nbr_ks + o*nbr_os]][k] + bias[0] ): the gather half of a 3x3x3 convolution with ONE output
 * channel, whose per-offset dot products y = X @ [w_0 | ... | w_26] (padded to 32 columns) were computed per input row
 * by fpcc_conv_f32.  Together they replace MinkowskiConvolution(C, 1, kernel_size=3) (occupancy / residual heads,
 * models/convolutional/lossy_coord_v2/layers.py:241-242,258).  Summation order 2: per-offset FMA chains from zero, partial
 * sums added in ascending offset order, then the bias. */
int fpcc_gather_sum_f32(const float *y, int ldy, const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os,
                        int64_t n, const float *bias, int act, const float *slope, float clip, float *out, void *stream);
/* The same on the GENERATED set of a level (row = 8 parent + octant, all 8 m candidates; fpcc_nbr27_from_parent with keys == NULL)
 * without building that set's table: parent_nbr is the PARENT level's offset-major table [27][m]; the 27 neighbour rows of every
 * candidate are derived in registers.  y has 8 m rows; out float [8 m].  Same sum, same bits as fpcc_gather_sum_f32 on the built table. */
int fpcc_gather_sum_generated_f32(const float *y, int ldy, const int32_t *parent_nbr, int64_t m, const float *bias, int act,
                                  const float *slope, float clip, float *out, void *stream);

/* out[i, :] = x[index[i], :] -- features re-ordered into the canonical (Morton) row order of a coordinate map, the
 * permutation ME.SparseTensor applies to its features (models/convolutional/lossy_coord_v2/model.py:147-153). */
int fpcc_gather_rows_f32(const float *x, int c, int ld, const int32_t *index, int64_t n, float *out, int ldo, void *stream);
/* Offset-major neighbour table [n_offsets][n] (fpcc_nbr27_from_parent) -> row-major rows_out [n][ld] (entries n_offsets .. ld - 1 of a
 * row are -1; ld % 4 == 0).  fpcc_conv_f32 takes either layout through (nbr_ks, nbr_os) = (n, 1) / (1, ld); from the row-major one
 * (16-byte aligned) the MFMA kernels read a row's entries as 16-byte pieces of ONE cache line instead of n_offsets 4-byte requests to
 * n_offsets lines -- what their prologue costs once the rows of a block are not consecutive (row_order). */
int fpcc_transpose_table_i32(const int32_t *table, int n_offsets, int64_t n, int32_t *rows_out, int ld, void *stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Entropy-model glue that has to sit next to the features                                                         */
/* ------------------------------------------------------------------------------------------------------------ */

/* p = clip(round(float64(sigmoid_fp32(logit)) * 65536), 1, 65535) as uint16: GeoLosslessEntropyModel.init_prob
 * (models/convolutional/lossy_coord_lossy_color/geo_lossl_em.py:95-99). */
int fpcc_logit_to_prob16(const float *logit, int64_t n, uint16_t *prob_out, void *stream);

/* x = round_half_even(x * scale) in place, symbols_out = int32(x), then x /= scale
 * (geo_lossl_em.py:168-172, 202-206).  symbols_out may be NULL. */
int fpcc_quantize_symbols(float *x, int64_t n, float scale, int32_t *symbols_out, void *stream);

/* occupancy of candidate children: mask[8*m] = child_row[p][k] >= 0  (get_coord_mask, geo_lossl_em.py:306-317) */
int fpcc_child_mask(const int32_t *child_row, int64_t m, uint8_t *mask_out, void *stream);

/* Adaptive top-k pruning of Decoder.get_keep (models/convolutional/lossy_coord_v2/layers.py:151-180) for one sample:
 * logit[8*m] in (parent, octant) order; keep = logit > kth_smallest(non-local-max logits, 8m - target) OR local max.
 * A voxel is a local max when it equals the maximum over the 8 candidates of its parent.  keep_out uint8[8m].
 * The threshold is found by radix SELECT (three histogram passes over the logits, ~12 bytes read per candidate; rounds 1-5 sorted
 * them all): an order statistic does not depend on how it is found, so the mask is unchanged.  ws (16-byte aligned): the size the
 * call returns with ws == NULL. */
int64_t fpcc_topk_keep(const float *logit, int64_t m, int64_t target, uint8_t *keep_out,
                       void *ws, int64_t ws_bytes, void *stream);

/* Final step of Decoder.test_forward (models/convolutional/lossy_coord_v2/layers.py:148 `fea.C[:, 1:][keep]`) fused with
 * the `+ coord_offset` of PCC.decompress (lossy_coord_v2/model.py:274): the kept children of parents pkeys[m]
 * (level+1 keys) -> xyz_out int32 [<=8m, 3] in Morton order at `level`; count_out device int32[1]. */
int64_t fpcc_compact_coords(const int64_t *pkeys, int64_t m, const uint8_t *mask, int level, int bits,
                            const int32_t *offset_xyz, int32_t *xyz_out, int32_t *count_out,
                            void *ws, int64_t ws_bytes, void *stream);

/* Same rule with coarser cells, as Decoder.get_keep needs when a stage sits more than one level below the coarsest
 * decoder level (models/convolutional/lossy_coord_lossy_color/layers.py:183-190: pooling stride = 2^stages / stride):
 * candidate group p (8 children of one voxel) belongs to cell cell_of_group[p] in [0, n_cells); a candidate is a local
 * maximum when it equals the maximum over its whole cell. */
int64_t fpcc_topk_keep_cells(const float *logit, int64_t m, const int32_t *cell_of_group, int64_t n_cells, int64_t target,
                             uint8_t *keep_out, void *ws, int64_t ws_bytes, void *stream);

/* Rate term of the noisy deep-factorised bottleneck (training): for y [n][c] (row stride ldy) and the per-channel parameters of a
 * 1-3-3-3-3-1 logit network -- weights[i] [c][f_out][f_in], biases[i] [c][f_out], factors[i] [c][f_out], raw as stored
 * (softplus / tanh applied inside) --
 *     logp = log(cdf(y + half_width) - cdf(y - half_width))       (survival functions right of the median)
 * out [c][59]: per channel the gradients of sum(logp) w.r.t. the 33 weights, 13 biases, 12 factors (layer by layer, row-major)
 * and, last, sum(logp) itself;  dy (may be NULL) [n][c]: d sum(logp) / dy.  One kernel instead of the ~60 tensor ops (and
 * their autograd twins) of lib/entropy_models/distributions/deep_factorized.py:24-40 + uniform_noise.py:30-63 behind the
 * bits loss of continuous_batched.py:62-69.  ws: fpcc_deep_factorized_ws_bytes() bytes; reduction order fixed. */
int64_t fpcc_deep_factorized_ws_bytes(int64_t n, int c);
int fpcc_deep_factorized_bits_f32(const float *y, int64_t n, int c, int ldy, const float *const *weights,
                                  const float *const *biases, const float *const *factors, float half_width,
                                  float *dy, int lddy, float *out, void *ws, int64_t ws_bytes, void *stream);

/* Rate term of the scale-indexed noisy normal (Gaussian-conditional bottleneck, training), element-wise over n values:
 *     s = exp(log_scale_offset + log_scale_factor * index),   lp = log(Phi((y + h) / s) - Phi((y - h) / s))
 * out[0] = sum lp (fixed reduction order);  dy / dindex (either may be NULL): its derivatives.  log Phi in the three segments
 * of lib/entropy_models/distributions/special_math.py:138-258, the survival side right of the median as in
 * distributions/uniform_noise.py:36-63; behind the bits loss of continuous_indexed.py:145-159 with the parameter functions of
 * :265-273 (`ScaleNoisyNormalEntropyModel`, hyperprior/noisy_deep_factorized/basic.py:158-203). */
int64_t fpcc_noisy_normal_ws_bytes(int64_t n);
int fpcc_noisy_normal_bits_f32(const float *y, const float *index, int64_t n, float log_scale_offset, float log_scale_factor,
                               float half_width, float *dy, float *dindex, float *out, void *ws, int64_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Geometry distortion (D1, point-to-point)                                                                       */
/* ------------------------------------------------------------------------------------------------------------ */
/* dist2_out[i] = squared distance from query i = (batch, x, y, z) to its nearest voxel of the same batch in the set given
 * by sorted unique keys (fpcc_keys_from_coords at level 0, `bits` bits per axis); -1 when the batch is empty in the
 * set.  nn_row_out (may be NULL) = row of that voxel (ties: the first in key order).  Exact, integer.
 * Replaces the PLY round trip through the external `pc_error` binary (lib/evaluators.py:94-112,
 * lib/metrics/pc_error_wrapper.py:40-107) and the brute-force KNN of lib/knn3d/src/knn3d.cu:74-130. */
int fpcc_nn_dist2(const int64_t *keys, int64_t m, int bits, const int32_t *query, int64_t n, int64_t *dist2_out,
                  int32_t *nn_row_out, void *stream);
/* K nearest neighbours (1 <= K <= 16) of every float point p1[n1][3] among p2[n2][3]: idx_out int64 [n1][K] (-1 where n2 < K),
 * dist2_out float [n1][K] squared distances, ascending per row (the reference leaves the order unspecified); brute force.
 * Replaces knn3d_ext.knn3d(p1, p2, K, version) (lib/knn3d/src/binding.cpp:5-7, knn3d.cu:74-130; the `version` argument picks
 * among equivalent kernels there and has no counterpart). */
int fpcc_knn3d(const float *p1, int64_t n1, const float *p2, int64_t n2, int k, int64_t *idx_out, float *dist2_out, void *stream);
/* *sum_out (device) = sum of the non-negative entries; integer, so independent of the reduction order. */
int fpcc_sum_i64(const int64_t *values, int64_t n, uint64_t *sum_out, void *stream);

/* Point-to-plane (D2) and Hausdorff distortion -- what the reference obtains from `pc_error -a A -b B -n normals --hausdorff=1`
 * (lib/metrics/pc_error_wrapper.py:40-76; the normals of A come from its PLY file or from Open3D's estimate_normals, :57-72).
 * All clouds are SORTED unique Morton key sets (fpcc_keys_from_coords + fpcc_sort_keys); rows index those sorted sets; queries are
 * int32 (batch, x, y, z) rows, 16-byte aligned.
 *
 * fpcc_knn_voxels: the k (1..32) nearest voxels of every query in the total order (squared distance, row) -- idx_out int32 [n][k]
 *   (-1 where m < k), dist2_out int64 [n][k].  The search starts at block edge 2^start_level (a hint; any value gives the same result).
 * fpcc_pca_normals: one unit normal per point (double [n][3]) = the eigenvector of the smallest eigenvalue of the covariance of its
 *   k neighbours `nbr` int32 [n][k] (rows of `keys`, negative = absent), closed-form symmetric 3x3 eigen-decomposition; sign chosen so
 *   that n . (1, sqrt 2, sqrt 5) > 0; (0, 0, 1) where fewer than 3 neighbours or a vanishing covariance (Open3D's rule).
 * fpcc_nn_plane_dist2: per query, over ALL voxels j at the minimum distance (pc_error's neighbours "at the same distance"):
 *   plane_out[i] = mean_j ((q_i - p_j) . normals[j])^2, dist2_out[i] = that minimum (int64, optional), nn_row_out[i] = the first such
 *   row (optional).  normals: double [m][3] in row order (NULL: plane_out = 0); rows whose normal is NaN are skipped.
 * fpcc_transfer_normals: pc_error's normals for the second cloud B from those of A: every point of A adds its normal to its nearest
 *   voxel of B (first row among ties); a voxel of B that received c > 0 normals takes their mean, every other voxel the mean normal
 *   of its own nearest voxels of A.  coords_a / coords_b: the voxels of keys_a / keys_b in row order.  Sums in 2^-40 fixed point: independent of the order of the atomics.  ws: caller-owned,
 *   16-byte aligned, fpcc_transfer_normals_ws_bytes(n_a, n_b) bytes.
 * fpcc_sum_max_f64: sum_max_out[0] = sum, [1] = maximum of `values` in a fixed order (slices of 4096, fixed tree, slices added in
 *   ascending order): a function of the data alone.  ws: 16 bytes per 4096 values. */
int fpcc_knn_voxels(const int64_t *keys, int64_t m, int bits, const int32_t *query, int64_t n, int k, int start_level, int32_t *idx_out,
                    int64_t *dist2_out, void *stream);
int fpcc_pca_normals(const int64_t *keys, int64_t m, int bits, const int32_t *nbr, int64_t n, int k, double *normals_out, void *stream);
int fpcc_nn_plane_dist2(const int64_t *keys, int64_t m, int bits, const double *normals, const int32_t *query, int64_t n,
                        int64_t *dist2_out, int32_t *nn_row_out, double *plane_out, void *stream);
int64_t fpcc_transfer_normals_ws_bytes(int64_t n_a, int64_t n_b);
int fpcc_transfer_normals(const int64_t *keys_a, int64_t n_a, const int32_t *coords_a, const double *normals_a, const int64_t *keys_b,
                          int64_t n_b, const int32_t *coords_b, int bits, double *normals_b_out, void *ws, int64_t ws_bytes, void *stream);
int fpcc_sum_max_f64(const double *values, int64_t n, double *sum_max_out, void *ws, int64_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Integer-only pipeline (lossl_coord_int).  Replaces the pybind module `int_sparse_conv_ext`                      */
/* (lib/int_sparse_conv/src/binding.cu:114-145).  All arithmetic is exact: results are order independent.          */
/* ------------------------------------------------------------------------------------------------------------ */

/* One-time upload of the exponent table of the LUT softmax (lib/int_sparse_conv/src/softmax.cu:13-22,108-117).  Called
 * lazily by the softmax entry points; call it explicitly before capturing a graph. */
int fpcc_int_init(void);

/* GPUHashTable(keys int64[cap], vals int32[cap]) -- caller-owned, zero-initialised storage, same slot layout and hash
 * as the reference (lib/int_sparse_conv/src/hashmap/hashmap_cuda.cuh:59-73,146-191), so cached tables interchange.
 *   insert_coords : coords int32 [n][4] = (x, y, z, batch), value = row + 1            (hashmap_cuda.cuh:171-191)
 *   lookup_coords : out int32 [n][volume], entry = row + 1 of the voxel at coord*stride + offset_k, 0 when absent;
 *                   kernel_sizes / strides are HOST int32[3] (the reference passes device tensors only to read them in
 *                   the kernel); offset order as in hashmap_cuda.cuh:239-258 (odd volume: x fastest, even: z fastest)
 *   insert_keys / lookup_keys : the generic-key variants (hashmap_cuda.cuh:146-168,195-218). */
int fpcc_hash_insert_coords(int64_t *table_keys, int32_t *table_vals, int64_t capacity, const int32_t *coords, int64_t n,
                            void *stream);
int fpcc_hash_lookup_coords(const int64_t *table_keys, const int32_t *table_vals, int64_t capacity, const int32_t *coords,
                            int64_t n, const int32_t *kernel_sizes_host, const int32_t *strides_host, int32_t *out,
                            void *stream);
/* the same two operations on rows laid out (batch, x, y, z) -- the layout of the models' coordinate tensors; the keys and
 * therefore the tables are identical (no re-ordering pass over the coordinates) */
int fpcc_hash_insert_coords_bxyz(int64_t *table_keys, int32_t *table_vals, int64_t capacity, const int32_t *coords, int64_t n,
                                 void *stream);
int fpcc_hash_lookup_coords_bxyz(const int64_t *table_keys, const int32_t *table_vals, int64_t capacity, const int32_t *coords,
                                 int64_t n, const int32_t *kernel_sizes_host, const int32_t *strides_host, int32_t *out,
                                 void *stream);
int fpcc_hash_insert_keys(int64_t *table_keys, int32_t *table_vals, int64_t capacity, const int64_t *keys, int64_t n,
                          void *stream);
int fpcc_hash_lookup_keys(const int64_t *table_keys, const int32_t *table_vals, int64_t capacity, const int64_t *keys,
                          int64_t n, int32_t *out, void *stream);

/* int8 x int8 -> int32 sparse convolution / linear with the fixed-point epilogue fused:
 *      acc[o][j] = sum_k  A[in_k(o), :] . W[k][j, :]  (+ zp_comp[k][j] for every present offset k)
 *      out[o][j] = requant_mul == NULL ? acc
 *                : clamp_T( rha( prelu_q6.25(acc + bias[j]) * requant_mul[j] + zero_point, shift ) ),  T = int8 | int32
 * Replaces cutlass_gather_gemm_scatter_int8 + cutlass_gemm_int8 (gather_gemm_scatter.cu:216-268, gemm.cu:197-247) as
 * driven by sparse_conv_in8w8out32 (lib/int_sparse_conv/cuda_ops.py:95-169), and the epilogue kernels
 * {bias_,prelu_,bias_prelu_,}requant_to_int{8,32} (src/element_wise/*.cu) that follow them (cuda_ops.py:383-403,609-635).
 *   A        int8 [n_in][lda], lda a multiple of 16, columns c_in..lda zero;
 *   nbr      input row of (offset k, output row o) = nbr[k*nbr_ks + o*nbr_os] - nbr_bias, negative = absent
 *            (a lookup_coords result is used directly with (ks, os, bias) = (1, volume, 1)); NULL = identity;
 *   W        int8 [n_offsets][c_out][ldw] (the reference's [K, C_out, C_in] with rows padded to ldw, multiple of 16);
 *   slope    device int32[1] (Q6.25) or NULL; zero_point device int64[1] or NULL;
 *   out      int8 or int32 [n_out][ldo]; for int8 outputs the columns c_out..out_pad are written as zeros so that the
 *            tensor can feed the next layer as A;
 *   row_order  NULL, or a permutation of [0, n_out): tile position p computes output row row_order[p] (see fpcc_conv_row_keys;
 *            on LiDAR sweeps a 32-row block in Morton order executes 2-4x the offsets its rows have, in pattern order 1.2-1.3x);
 *   ws       maps of at most 8192 rows with >= 8 kernel offsets are evaluated one workgroup per (row tile, offset), the raw
 *            sums added atomically (integer addition: any order gives the same bits); with an epilogue this needs
 *            fpcc_conv_i8_ws_bytes() bytes of device scratch, otherwise ws may be NULL. */
int fpcc_conv_i8(const int8_t *a, int c_in, int lda, const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os,
                 int nbr_bias, const int8_t *w, int ldw, const int32_t *zp_comp, const int32_t *bias, const int32_t *slope,
                 const uint32_t *requant_mul, const int64_t *zero_point, int shift, int out_bits, void *out, int ldo,
                 int out_pad, int c_out, int64_t n_out, const int32_t *row_order, void *ws, int64_t ws_bytes, void *stream);
int64_t fpcc_conv_i8_ws_bytes(int has_nbr, int n_offsets, int has_requant, int c_out, int64_t n_out);
/* The same with the tail of SparseResBlockIn32W8Out32.forward (cuda_ops.py:82-92) fused behind the epilogue of its second
 * convolution: out = clamp_i32(prelu_q6.25(residual + requant(...) (int32, wrapping), slope2)) -- `prelu` of
 * src/element_wise/prelu.cu applied to the reference's wrapping int32 tensor add.  residual int32 [n_out][ld_res] (NULL:
 * exactly fpcc_conv_i8); needs out_bits == 32 and a requantising epilogue. */
int fpcc_conv_i8_res(const int8_t *a, int c_in, int lda, const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os,
                     int nbr_bias, const int8_t *w, int ldw, const int32_t *zp_comp, const int32_t *bias, const int32_t *slope,
                     const uint32_t *requant_mul, const int64_t *zero_point, int shift, int out_bits, void *out, int ldo,
                     int out_pad, int c_out, int64_t n_out, const int32_t *row_order, const int32_t *residual, int ld_res,
                     const int32_t *slope2, void *ws, int64_t ws_bytes, void *stream);

/* Stand-alone epilogue on an int32 matrix [n][ch] (row stride ldi): requant_to_int{8,32}, bias_requant_*, prelu_requant_*,
 * bias_prelu_requant_* (src/element_wise/*.cu).  mul_per_channel == 0 broadcasts requant_mul[0]
 * (RequantFxpToScaledInt8, cuda_ops.py:505-509). */
int fpcc_epilogue_i32(const int32_t *in, int ldi, const int32_t *bias, const int32_t *slope, const uint32_t *requant_mul,
                      int mul_per_channel, const int64_t *zero_point, int shift, int out_bits, void *out, int ldo,
                      int out_pad, int64_t n, int ch, const int32_t *row_group, void *stream);
/* row_group (NULL = none): bias / requant_mul hold one set of `ch` values per group and row r uses set row_group[r] --
 * the epilogue of a linear layer C -> 8*C of which only the occupied octants are evaluated (model.py:66-74: the
 * reference computes all 8*C columns and indexes the occupied ones afterwards). */

/* out = clamp_i32(prelu_q6.25(a (+ b, wrapping))): `prelu` (src/element_wise/prelu.cu) with the residual add of
 * SparseResBlockIn32W8Out32.forward (cuda_ops.py:90) fused; b may be NULL. */
int fpcc_prelu_i32(const int32_t *a, const int32_t *b, const int32_t *slope, int64_t n, int32_t *out, void *stream);

/* softmax_int32 (src/softmax.cu:119-144): in int32 [n][c] Q15.16 -> out uint32 [n][c] Q0.32, 2 <= c <= 256. */
int fpcc_softmax_i32(const int32_t *in, int64_t n, int c, uint32_t *out, void *stream);
/* Model.batch_quantize_pmf_torch (models/convolutional/lossl_coord_int/model.py:345-353) in one pass: logits >> pre_shift,
 * LUT softmax, ((p * (65536 - c)) >> 32) + 1, cumulative sum, last entry 65535 -> uint16 CDF rows [n][c]. */
int fpcc_logits_to_cdf16(const int32_t *logits, int64_t n, int c, int pre_shift, uint16_t *cdf_out, void *stream);
/* Encoder side of the same: only the (start, freq - 1) of the coded symbol of every row leaves the device (4 B per
 * symbol instead of 2*c), ready for fpcc_simple_enc_push_ranges. */
int fpcc_logits_to_ranges(const int32_t *logits, int64_t n, int c, int pre_shift, const int16_t *symbols,
                          uint16_t *start_out, uint16_t *freq_minus_1_out, void *stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Integer codec: the traversal of one octree level in two calls                                                  */
/* ------------------------------------------------------------------------------------------------------------ */
/* One int8 convolution / linear layer with its fixed-point epilogue -- the state of a SparseConv*In8W8* / Linear*In8W8* module
 * (lib/int_sparse_conv/cuda_ops.py:194-206,516-528): w int8 [n_offsets][c_out][ldw] (rows zero-padded to ldw, a multiple of 16);
 * the other fields as fpcc_conv_i8 takes them. */
typedef struct {
    const int8_t *w; int ldw; int c_in; int c_out; int n_offsets;
    const int32_t *zp_comp; const int32_t *bias; const int32_t *slope;
    const uint32_t *requant_mul; const int64_t *zero_point; int shift; int out_bits;
} fpcc_i8_layer;
/* RequantFxpToScaledInt8 (cuda_ops.py:473-509): Q8.23 -> scaled int8; shift already includes the 23 fraction bits */
typedef struct { const uint32_t *requant_mul; const int64_t *zero_point; int shift; } fpcc_i8_requant;
/* The layers of a OneScalePredictor (models/convolutional/lossl_coord_int/model.py:95-213) of `channels` = C (a multiple of 16):
 *   dec      SparseResBlockIn32W8Out32: dec_in (input_requant), dec_conv1 (C->C, 3x3x3, PReLU, int8), dec_conv2 (C->C, 3x3x3, Q8.23),
 *            dec_slope (the block's PReLU, Q6.25)
 *   pred     pred_in, pred_conv (C->C, 3x3x3, PReLU, int8), pred_linear (C->255, Q8.23 logits)
 *   upsample (has_upsample) up_in, up_linear (C+8 -> C, PReLU, Q8.23), a residual block (up_res_in, up_conv1, up_conv2, up_slope),
 *            up_out_in, up_out (C -> 8 C, Q8.23). */
typedef struct {
    int channels; int has_upsample;
    fpcc_i8_requant dec_in; fpcc_i8_layer dec_conv1, dec_conv2; const int32_t *dec_slope;
    fpcc_i8_requant pred_in; fpcc_i8_layer pred_conv, pred_linear;
    fpcc_i8_requant up_in; fpcc_i8_layer up_linear;
    fpcc_i8_requant up_res_in; fpcc_i8_layer up_conv1, up_conv2; const int32_t *up_slope;
    fpcc_i8_requant up_out_in; fpcc_i8_layer up_out;
} fpcc_int_onescale;
/* First half of a level (OneScalePredictor._trunk, model.py:125-137,150-152): res_out = dec(feat), logits = pred(res_out), every layer
 * through fpcc_conv_i8_also as the layer-by-layer path issues it (same integers).  feat int32 [n][C] (Q8.23); feat_q8 int8 [n][C] =
 * feat requantised by dec_in when its producer wrote it (NULL: done here); nbr27 / row_order: the kernel map of the level's 3x3x3
 * convolutions as fpcc_conv_i8 reads it (row + 1 | 0, [ceil128(n)][27]; row_order may be NULL).  Outputs: res_out int32 [n][C], q_pred
 * int8 [n][C] (res_out requantised by pred_in), q_up int8 [n][ld_up] (by up_in, columns [C, ld_up) left to fpcc_int_level_expand; NULL
 * for a block without upsampling), logits int32 [n][255].  ws == NULL returns the workspace size. */
int64_t fpcc_int_level_trunk(const fpcc_int_onescale *blk, int64_t n, const int32_t *feat, const int8_t *feat_q8, const int32_t *nbr27,
                             const int32_t *row_order, int32_t *res_out, int8_t *q_pred, int8_t *q_up, int ld_up, int32_t *logits,
                             void *ws, int64_t ws_bytes, void *stream);
/* Second half (OneScalePredictor._expand, model.py:139-148,169-175): the features of the m occupied children of the level's n voxels,
 * feat_out int32 [m][C] = upsample(cat(res, occupancy bits)) evaluated for the occupied (row, octant) pairs only (row-major), and
 * feat_q8_out int8 [m][C] = feat_out requantised by *next_in (the next level's dec_in; both may be NULL).  symbols int16 [n] (symbol + 1
 * = the occupancy byte); coords int32 [n][4] + child_coords int32 [m][4]: the decoder's coordinate step (both NULL on the encoder side,
 * which has the coordinates from its analysis); q_up as fpcc_int_level_trunk left it, ld_up >= ceil16(C + 8). */
int64_t fpcc_int_level_expand(const fpcc_int_onescale *blk, int64_t n, int64_t m, const int32_t *res, int8_t *q_up, int ld_up,
                              const int16_t *symbols, const int32_t *coords, const int32_t *nbr27, const int32_t *row_order,
                              int32_t *child_coords, int32_t *feat_out, const fpcc_i8_requant *next_in, int8_t *feat_q8_out, void *ws,
                              int64_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* rANS decoders on the device (one wave per stream; the streams of libfpcc_host, same arithmetic)                  */
/* ------------------------------------------------------------------------------------------------------------ */
/* Binary decoder = BinaryRansCoder.decode of rans_ext_cpp (lib/entropy_models/rans_coder/rans_wrapper.cpp:385-428) with
 * stream, probabilities and result on the device: bits_out[i] in {0, 1}, ones_out[0] = number of ones, status[0] = 0 or
 * -2 (initial state below 2^23: not a stream).  Decoding is serial by the format (one state per stream); the wave's other
 * lanes fetch probabilities 64 at a time and hold the byte window. */
int fpcc_rans_binary_decode_dev(const uint8_t *stream, int64_t stream_len, const uint16_t *prob1, int64_t n,
                                uint8_t *bits_out, int32_t *ones_out, int32_t *status, void *hip_stream);
/* Additional int8 copies of a Q8.23 int32 result, requantised for its consumers (RequantFxpToScaledInt8, cuda_ops.py:473-509) inside
 * the producer's epilogue: out8[r][c] = clamp8(rha(v * requant_mul[0] + zero_point[0], shift)), columns [c_out, pad) zeroed.
 * fpcc_conv_i8_also / fpcc_epilogue_i32_also = fpcc_conv_i8_res / fpcc_epilogue_i32 with up to two such outputs (int32 primary
 * output with requantisation only); the consumers' stand-alone requantisation launches then have nothing left to do.
 * fpcc_fill_bits_i8 writes the occupancy bits of a level behind such a matrix: out[r][col0 + k] = q(bits[r][k] ? fxp_one : 0),
 * columns [col0 + 8, pad) zero -- the int8 image of requant(cat(R, bits << 23)) without the int32 concatenation. */
typedef struct {
    int8_t *out; int ld; int pad;
    const uint32_t *requant_mul; const int64_t *zero_point; int shift;
} fpcc_requant8;
int fpcc_conv_i8_also(const int8_t *a, int c_in, int lda, const int32_t *nbr, int n_offsets, int64_t nbr_ks, int64_t nbr_os,
                      int nbr_bias, const int8_t *w, int ldw, const int32_t *zp_comp, const int32_t *bias, const int32_t *slope,
                      const uint32_t *requant_mul, const int64_t *zero_point, int shift, int out_bits, void *out, int ldo,
                      int out_pad, int c_out, int64_t n_out, const int32_t *row_order, const int32_t *residual, int ld_res,
                      const int32_t *slope2, const fpcc_requant8 *also, int n_also, void *ws, int64_t ws_bytes, void *stream);
int fpcc_epilogue_i32_also(const int32_t *in, int ldi, const int32_t *bias, const int32_t *slope, const uint32_t *requant_mul,
                           int mul_per_channel, const int64_t *zero_point, int shift, int out_bits, void *out, int ldo, int out_pad,
                           int64_t n, int ch, const int32_t *row_group, const fpcc_requant8 *also, int n_also, void *stream);
int fpcc_fill_bits_i8(const uint8_t *bits, int64_t n, int32_t fxp_one, const uint32_t *requant_mul, const int64_t *zero_point,
                      int shift, int8_t *out, int ld, int col0, int pad, void *stream);
/* One octree step of the integer codec's traversal: from a level's child occupancy -- symbols int16 [n] (symbol + 1 = the 8 bits,
 * bit (7 - k) = child k = 4 dx + 2 dy + dz, model.py:60) or bits uint8 [n][8] -- and the parents' coordinates int32 [n][4]
 * (b, x, y, z), everything the reference derives with nonzero / index_select / shift / add / cat / scatter tensor operators
 * (models/convolutional/lossl_coord_int/model.py:60-74,169-175,262-295,430-470): per occupied child j (row-major over parents
 * and octants, m of them = the popcount the caller knows on the host) child_coords[j] = (b, 2x+dx, 2y+dy, 2z+dz),
 * parent_row[j], octant[j], table[j][8] (parent_row + 1 in column octant, else 0: the one-entry-per-row gather table of an
 * "occupied outputs only" linear layer; rows [m, table_rows) are zeroed); per parent bits_out [n][8] (0 | 1) and bits_fxp
 * [n][8] (0 | fxp_one: the occupancy bits as Q8.23 features).  Any output may be NULL.  ws == NULL returns the workspace size. */
int64_t fpcc_octree_children(const int16_t *symbols, const uint8_t *bits, const int32_t *coords, int64_t n, int64_t m,
                             int32_t fxp_one, int32_t *child_coords, int32_t *parent_row, int32_t *octant, int32_t *table,
                             int64_t table_rows, uint8_t *bits_out, int32_t *bits_fxp, void *ws, int64_t ws_bytes, void *stream);
/* RansDecoder.decode of simple_rans_ext_cpp (models/convolutional/lossy_coord_v3/rans_coder/simple_rans_wrapper.cpp:206-239)
 * on the device: rows uint16 [n | 1][width <= 256] as fpcc_logits_to_cdf16 writes them, state int32[4] = {x, position low,
 * position high, status} carried from launch to launch (initialise from fpcc_simple_dec_tell of libfpcc_host or from the
 * stream's first four bytes with position 4, status 0).  children_out (int32[2], may be NULL): [0] = sum of
 * popcount(symbol + 1), the number of occupied children of the level; [1] = the sticky status word state[3]: 0 fine, bit 0 the
 * state on entry is not a rANS state (< 2^23), bit 1 a CDF row does not increase at the decoded symbol, bit 2 the decoder read
 * past the end of the stream. */
int fpcc_simple_dec_pop_dev(int32_t *state, const uint8_t *stream, int64_t stream_len, const uint16_t *rows,
                            int64_t n_rows, int64_t width, uint16_t *symbols_out, int64_t n, int32_t *children_out,
                            void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
