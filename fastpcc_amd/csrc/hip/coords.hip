// Coordinate machinery: Morton keys, sorting, the octree-shaped pyramid of coordinate maps and the 3x3x3 kernel maps.
//
// Design (MI355X-first, not a port of MinkowskiEngine's or torchsparse's hash maps): every coordinate set is a SORTED
// array of unique 64-bit Morton keys.  With x on Morton bit 0 the parent of a key is key >> 3 and its octant is key & 7,
// which is exactly the (x fastest) kernel index of a 2x2x2 stride-2 kernel.  Hence
//   * a stride-2 map is a run-length pass over the sorted keys (no hashing),
//   * transposed / generative maps are the child_row table read the other way,
//   * the 27-neighbour table of a level follows from its parent's table with two dependent, cache-friendly loads,
//   * rows that are neighbours in space are neighbours in memory, so gathers hit L2.
// All kernels are HBM/L2-bound integer work; they use 256-thread blocks, one element (or one parent) per thread and
// fully coalesced streaming accesses for everything that is not a table lookup.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"
#include <atomic>

namespace fpcc {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ uint64_t spread21(uint32_t v) {
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__device__ __forceinline__ uint32_t gather21(uint64_t x) {
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
    x = (x ^ (x >> 32)) & 0x1fffffull;
    return static_cast<uint32_t>(x);
}

__device__ __forceinline__ uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return spread21(x) | (spread21(y) << 1) | (spread21(z) << 2);
}

__global__ void k_morton(const int32_t *__restrict__ c, int64_t n, int64_t ld, int a0, int a1, int a2,
                         int64_t *__restrict__ keys) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t *r = c + i * ld;
    keys[i] = static_cast<int64_t>(morton3((uint32_t)r[a0], (uint32_t)r[a1], (uint32_t)r[a2]));
}

// ---- Hilbert keys ---------------------------------------------------------------------------------------------------
// hilbert3d_encode_lut of the reference (lib/space_filling_curves/src/hilbert3d.cu:28-60) walks the coordinate bits from the
// top through a 12-state machine: (state, Morton octant x | y << 1 | z << 2) -> (Hilbert digit, next state).  The machine
// is generated here from the curve's geometry -- the root cube's octant order (reflected Gray path) and, per visited octant,
// the signed axis permutation that maps the root curve onto the child's -- and uploaded once (fpcc_hilbert_init);
// tests/golden/hilbert.json holds keys of the reference's own table to pin it.
__device__ uint8_t g_hilbert[12 * 8];         // next_state * 8 | digit

__global__ void k_hilbert(const int32_t *__restrict__ c, int64_t n, int64_t ld, int a0, int a1, int a2, int bits,
                          int64_t *__restrict__ keys) {
    __shared__ uint8_t tab[96];
    if (threadIdx.x < 96) tab[threadIdx.x] = g_hilbert[threadIdx.x];
    __syncthreads();
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t *r = c + i * ld;
    const uint32_t x = (uint32_t)r[a0], y = (uint32_t)r[a1], z = (uint32_t)r[a2];
    uint32_t state = 0;
    uint64_t key = 0;
    for (int b = bits - 1; b >= 0; --b) {
        const uint32_t t = tab[state | ((x >> b) & 1u) | (((y >> b) & 1u) << 1) | (((z >> b) & 1u) << 2)];
        key = (key << 3) | (t & 7u);
        state = t & ~7u;
    }
    keys[i] = static_cast<int64_t>(key);
}

__global__ void k_keys_from_coords(const int4 *__restrict__ c, int64_t n, int level, int bits, int64_t *__restrict__ keys) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 v = c[i];   // (batch, x, y, z): one 16-byte load per row
    uint64_t m = morton3((uint32_t)(v.y >> level), (uint32_t)(v.z >> level), (uint32_t)(v.w >> level));
    keys[i] = static_cast<int64_t>(((uint64_t)(uint32_t)v.x << (3 * bits)) | m);
}

__global__ void k_coords_from_keys(const int64_t *__restrict__ keys, int64_t n, int level, int bits,
                                   const int32_t *__restrict__ off, int4 *__restrict__ out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = (uint64_t)keys[i];
    uint64_t m = k & ((1ull << (3 * bits)) - 1ull);
    int4 v;
    v.x = (int32_t)(k >> (3 * bits));
    v.y = (int32_t)(gather21(m) << level);
    v.z = (int32_t)(gather21(m >> 1) << level);
    v.w = (int32_t)(gather21(m >> 2) << level);
    if (off) { v.y += off[0]; v.z += off[1]; v.w += off[2]; }
    out[i] = v;
}

__global__ void k_iota(int32_t *p, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}

// flag[i] = 1 where a new group (keys >> shift differs from the predecessor) starts
__global__ void k_head_flags(const int64_t *__restrict__ keys, int64_t n, int shift, int32_t *__restrict__ flag) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) ? 1 : 0;
}

__global__ void k_unique_scatter(const int64_t *__restrict__ keys, int64_t n, const int32_t *__restrict__ flag,
                                 const int32_t *__restrict__ pos, int64_t *__restrict__ ukeys,
                                 int32_t *__restrict__ first, int32_t *__restrict__ count) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) {
        int32_t u = pos[i] - 1;
        ukeys[u] = keys[i];
        first[u] = (int32_t)i;
    }
    if (i == n - 1) count[0] = pos[i];
}

__global__ void k_coarsen_scatter(const int64_t *__restrict__ keys, int64_t n, const int32_t *__restrict__ flag,
                                  const int32_t *__restrict__ pos, int32_t *__restrict__ parent_of,
                                  int64_t *__restrict__ pkeys, int32_t *__restrict__ child_row,
                                  int32_t *__restrict__ count) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t p = pos[i] - 1;
    parent_of[i] = p;
    if (flag[i]) {
        // first child of its parent: siblings are the following rows (at most 8, sorted by octant).  All eight candidate keys are
        // requested at once (round 6; a loop that stopped at the first foreign key made up to eight DEPENDENT loads of it)
        int64_t kj[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) kj[t] = keys[i + t < n ? i + t : n - 1];
        const int64_t pk = kj[0] >> 3;
        int32_t rows[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rows[k] = -1;
        bool mine = true;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            mine = mine && i + t < n && (kj[t] >> 3) == pk;
            const int oct = (int)(kj[t] & 7);
#pragma unroll
            for (int k = 0; k < 8; ++k) if (mine && k == oct) rows[k] = (int32_t)(i + t);   // static indexing keeps rows[] in registers
        }
        pkeys[p] = pk;
        int4 *dst = reinterpret_cast<int4 *>(child_row + (int64_t)p * 8);
        dst[0] = make_int4(rows[0], rows[1], rows[2], rows[3]);
        dst[1] = make_int4(rows[4], rows[5], rows[6], rows[7]);
    }
    if (i == n - 1) count[0] = pos[i];
}

// ---- integer codec: one level of the encoder's octree analysis -----------------------------------------------------------------------
// The first child of every parent (head flag set) walks its <= 7 following siblings and writes the parent's whole row of every table;
// the spare threads zero the padding rows of the kernel map.  Keys: Morton code with z on bit 0 (child k = 4 dx + 2 dy + dz = key & 7),
// sample index from bit `batch_shift` of the PARENT key upwards.
__global__ void k_octree_level(const int64_t *__restrict__ keys, int64_t n, const int32_t *__restrict__ flag,
                               const int32_t *__restrict__ pos, int64_t m, int batch_shift, int64_t *__restrict__ pkeys,
                               int32_t *__restrict__ coords, int32_t *__restrict__ bits, int32_t *__restrict__ table, int64_t table_rows,
                               int16_t *__restrict__ symbols) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (table && i < table_rows - m) {
        int4 *t = reinterpret_cast<int4 *>(table + (m + i) * 8);
        t[0] = make_int4(0, 0, 0, 0);
        t[1] = make_int4(0, 0, 0, 0);
    }
    if (i >= n || !flag[i]) return;
    const int64_t p = pos[i] - 1;
    if (p >= m) return;                                   // a caller's wrong row count must not write out of bounds
    const int64_t pk = keys[i] >> 3;
    int32_t rows[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rows[k] = 0;
    for (int64_t j = i; j < n && j < i + 8; ++j) {
        const int64_t kj = keys[j];
        if ((kj >> 3) != pk) break;
        const int oct = (int)(kj & 7);
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k == oct) rows[k] = (int32_t)j + 1;
    }
    if (pkeys) pkeys[p] = pk;
    if (coords) {
        const uint64_t lo = (uint64_t)pk & ((1ull << batch_shift) - 1);
        reinterpret_cast<int4 *>(coords)[p] = make_int4((int32_t)(pk >> batch_shift), (int32_t)gather21(lo >> 2), (int32_t)gather21(lo >> 1),
                                                        (int32_t)gather21(lo));
    }
    if (table) {
        int4 *t = reinterpret_cast<int4 *>(table + p * 8);
        t[0] = make_int4(rows[0], rows[1], rows[2], rows[3]);
        t[1] = make_int4(rows[4], rows[5], rows[6], rows[7]);
    }
    int sym = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sym |= (rows[k] != 0) << (7 - k);
    if (bits) {
        int4 *b = reinterpret_cast<int4 *>(bits + p * 8);
        b[0] = make_int4(rows[0] != 0, rows[1] != 0, rows[2] != 0, rows[3] != 0);
        b[1] = make_int4(rows[4] != 0, rows[5] != 0, rows[6] != 0, rows[7] != 0);
    }
    if (symbols) symbols[p] = (int16_t)(sym - 1);
}

struct ByteToInt {
    __host__ __device__ int32_t operator()(uint8_t b) const { return b ? 1 : 0; }
};

// one thread per PARENT (round 6; was one per candidate): its 8 mask bytes as one load, its 8 scan values as two 16-byte loads, its
// child_row row as two 16-byte stores; keys and parent rows of the kept children go to consecutive output rows
template <bool MASK_ALIGNED>
__global__ void k_refine_scatter(const int64_t *__restrict__ pkeys, int64_t n_cand, const uint8_t *__restrict__ mask,
                                 const int32_t *__restrict__ pos, int64_t *__restrict__ keys_out,
                                 int32_t *__restrict__ parent_of, int32_t *__restrict__ child_row,
                                 int32_t *__restrict__ count) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (8 * p >= n_cand) return;
    uint64_t mk = 0;
    if (MASK_ALIGNED) {
        mk = reinterpret_cast<const uint64_t *>(mask)[p];
    } else {                                              // a mask that starts inside another tensor at an odd byte
#pragma unroll
        for (int k = 0; k < 8; ++k) mk |= (uint64_t)mask[8 * p + k] << (8 * k);
    }
    const int4 lo = reinterpret_cast<const int4 *>(pos)[2 * p], hi = reinterpret_cast<const int4 *>(pos)[2 * p + 1];
    const int32_t ps[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const int64_t pk = pkeys[p] << 3;
    int32_t rows[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        rows[k] = -1;
        if ((mk >> (8 * k)) & 0xffull) {
            rows[k] = ps[k] - 1;
            keys_out[rows[k]] = pk | k;
            parent_of[rows[k]] = (int32_t)p;
        }
    }
    int4 *dst = reinterpret_cast<int4 *>(child_row + 8 * p);
    dst[0] = make_int4(rows[0], rows[1], rows[2], rows[3]);
    dst[1] = make_int4(rows[4], rows[5], rows[6], rows[7]);
    if (8 * p + 8 == n_cand) count[0] = ps[7];
}

__global__ void k_compact_coords(const int64_t *__restrict__ pkeys, int64_t n_cand, const uint8_t *__restrict__ mask,
                                 const int32_t *__restrict__ pos, int level, int bits, const int32_t *__restrict__ off,
                                 int32_t *__restrict__ xyz, int32_t *__restrict__ count) {
    int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= n_cand) return;
    if (mask[c]) {
        const uint64_t k = ((uint64_t)pkeys[c >> 3] << 3) | (uint64_t)(c & 7);
        const uint64_t m = k & ((1ull << (3 * bits)) - 1ull);
        int32_t *o = xyz + (int64_t)(pos[c] - 1) * 3;
        o[0] = (int32_t)(gather21(m) << level) + (off ? off[0] : 0);
        o[1] = (int32_t)(gather21(m >> 1) << level) + (off ? off[1] : 0);
        o[2] = (int32_t)(gather21(m >> 2) << level) + (off ? off[2] : 0);
    }
    if (c == n_cand - 1) count[0] = pos[c];
}

// offset d = (dx+1) + 3*(dy+1) + 9*(dz+1), x fastest, centred: MinkowskiEngine's HYPER_CUBE enumeration for odd sizes
__global__ void k_nbr27_search(const int64_t *__restrict__ keys, int64_t n, int bits, int32_t *__restrict__ nbr) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = (uint64_t)keys[i];
    const uint64_t mmask = (1ull << (3 * bits)) - 1ull;
    const uint64_t batch = k & ~mmask;
    const uint64_t m = k & mmask;
    const int32_t x = (int32_t)gather21(m), y = (int32_t)gather21(m >> 1), z = (int32_t)gather21(m >> 2);
    const int32_t lim = 1 << bits;
    for (int d = 0; d < 27; ++d) {
        const int32_t nx = x + (d % 3) - 1, ny = y + (d / 3) % 3 - 1, nz = z + d / 9 - 1;
        int32_t found = -1;
        if (d == 13) {
            found = (int32_t)i;
        } else if (nx >= 0 && ny >= 0 && nz >= 0 && nx < lim && ny < lim && nz < lim) {
            const int64_t want = (int64_t)(batch | morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
            int64_t lo = 0, hi = n;   // first index with keys[idx] >= want
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (keys[mid] < want) lo = mid + 1; else hi = mid;
            }
            if (lo < n && keys[lo] == want) found = (int32_t)lo;
        }
        nbr[(int64_t)d * n + i] = found;
    }
}

// Round 6 formulation.  The 27 neighbours of a child lie in the 2 x 2 x 2 block of parents {ox - 1, ox} x {oy - 1, oy} x {oz - 1, oz} (parent
// offsets, o = the child's octant bits); their 8 x 8 children form a 4 x 4 x 4 CUBE indexed per axis by v = 2 b + c (b: which parent of
// the block, c: which child of it), and neighbour d in {-1, 0, 1} is cube entry v = d + 2 - o: the answers are the 3 x 3 x 3 sub-cube at
// offset s = 1 - o in {0, 1}^3.  So:
//   1. the 8 parent rows: 8 independent loads;
//   2. their child rows: 16 independent 16-byte loads (an absent parent reads row 0 and is masked afterwards) -- ALL in flight at once.
//      Rounds 2-5 walked the parents in a loop that skipped absent ones: a branch around each pair of loads, i.e. nine DEPENDENT round
//      trips per thread (75-83 % of the wave cycles parked, 42 % of 8 TB/s);
//   3. each of the 27 outputs is a select among 8 cube registers by the three bits of s (7 v_cndmask): no dynamic register index, no LDS,
//      no exec-masked stores; the offset-major table is stored straight from registers (one coalesced store per offset).
// MODE 0: the table [27][n].  MODE 1: instead of the table one word per row with bit d set where neighbour d exists (all a
// convolution of a constant input needs): 4 bytes written per row instead of 108.  MODE 2: the table and, from the same registers, the
// table once more ROW-MAJOR [n][32] (entries 27 .. 31 = -1; through an LDS tile so that a wave's store is 1 KB of consecutive
// addresses) and the masks (each may be NULL).
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_nbr27_from_parent(const int64_t *__restrict__ keys, const int32_t *__restrict__ parent_of, int64_t n,
                                                               const int32_t *__restrict__ pnbr, int64_t m,
                                                               const int32_t *__restrict__ child_row, int32_t *__restrict__ nbr,
                                                               int32_t *__restrict__ rows_out, uint32_t *__restrict__ masks_out) {
    constexpr int kPlane = kThreads + 1;
    __shared__ int32_t s_out[MODE == 2 ? 27 * kPlane : 1];
    const int64_t i_raw = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (MODE != 2 && i_raw >= n) return;
    const int64_t i = i_raw < n ? i_raw : n - 1;               // (MODE 2: a thread past the end recomputes the last row and stores nothing)
    const int oct = keys ? (int)(keys[i] & 7) : (int)(i & 7);   // keys == NULL: generated set, row = 8*parent + octant
    const int ox = oct & 1, oy = (oct >> 1) & 1, oz = oct >> 2;
    const int32_t p = parent_of ? parent_of[i] : (int32_t)(i >> 3);
    int32_t q[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int px = ox - 1 + (b & 1), py = oy - 1 + ((b >> 1) & 1), pz = oz - 1 + (b >> 2);      // in {-1, 0, 1}
        const int pd = (px + 1) + 3 * (py + 1) + 9 * (pz + 1);
        q[b] = pnbr[(int64_t)pd * m + (pd == 13 ? 0 : p)];                                           // (the centre is p itself: loaded and dropped, no branch)
        q[b] = pd == 13 ? p : q[b];
    }
    // cube[vx][vy][vz], v = 2 b + c per axis
    int32_t cube[4][4][4];
    if (child_row) {
        int4 lo[8], hi[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int64_t row = q[b] < 0 ? 0 : q[b];
            lo[b] = *reinterpret_cast<const int4 *>(child_row + row * 8);
            hi[b] = *reinterpret_cast<const int4 *>(child_row + row * 8 + 4);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int32_t kid[8] = {lo[b].x, lo[b].y, lo[b].z, lo[b].w, hi[b].x, hi[b].y, hi[b].z, hi[b].w};
#pragma unroll
            for (int c = 0; c < 8; ++c)
                cube[2 * (b & 1) + (c & 1)][2 * ((b >> 1) & 1) + ((c >> 1) & 1)][2 * (b >> 2) + (c >> 2)] = q[b] < 0 ? -1 : kid[c];
        }
    } else {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                cube[2 * (b & 1) + (c & 1)][2 * ((b >> 1) & 1) + ((c >> 1) & 1)][2 * (b >> 2) + (c >> 2)] = q[b] < 0 ? -1 : q[b] * 8 + c;
    }
    const bool sx = ox == 0, sy = oy == 0, sz = oz == 0;       // s = 1 - o
    uint32_t bits = 0;
    int32_t *mine = s_out + (MODE == 2 ? threadIdx.x : 0);
#pragma unroll
    for (int d = 0; d < 27; ++d) {
        const int dx = d % 3, dy = (d / 3) % 3, dz = d / 9;
        const int32_t a00 = sx ? cube[dx + 1][dy][dz] : cube[dx][dy][dz], a01 = sx ? cube[dx + 1][dy + 1][dz] : cube[dx][dy + 1][dz];
        const int32_t a10 = sx ? cube[dx + 1][dy][dz + 1] : cube[dx][dy][dz + 1], a11 = sx ? cube[dx + 1][dy + 1][dz + 1] : cube[dx][dy + 1][dz + 1];
        const int32_t b0 = sy ? a01 : a00, b1 = sy ? a11 : a10;
        const int32_t v = sz ? b1 : b0;
        bits |= (uint32_t)(v >= 0) << d;
        if (MODE != 1 && i_raw < n) nbr[(int64_t)d * n + i] = v;
        if (MODE == 2) mine[d * kPlane] = v;
    }
    if (MODE == 1) { nbr[i] = (int32_t)bits; return; }
    if (MODE == 2) {
        if (masks_out && i_raw < n) masks_out[i] = bits;
        if (rows_out) {
            // the tile row-major: lane e of pass j moves the 16-byte piece (row e / 8, piece e % 8) -- a wave's store is 1 KB of
            // consecutive addresses (each thread writing its own line piece by piece, 16 bytes at a 128-byte stride per instruction,
            // doubled the kernel's time)
            __syncthreads();
            const int64_t row0 = blockIdx.x * (int64_t)blockDim.x;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = j * kThreads + (int)threadIdx.x;
                const int r = e >> 3, piece = e & 7;
                if (row0 + r < n) {
                    int4 v;
                    v.x = 4 * piece < 27 ? s_out[(4 * piece) * kPlane + r] : -1;
                    v.y = 4 * piece + 1 < 27 ? s_out[(4 * piece + 1) * kPlane + r] : -1;
                    v.z = 4 * piece + 2 < 27 ? s_out[(4 * piece + 2) * kPlane + r] : -1;
                    v.w = 4 * piece + 3 < 27 ? s_out[(4 * piece + 3) * kPlane + r] : -1;
                    reinterpret_cast<int4 *>(rows_out + (row0 + r) * 32)[piece] = v;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// One octree step of the integer codec: everything its traversal derives from a level's 8-bit child occupancy, in one scatter
// pass behind a scan (the reference -- and round 2 of this build -- does it with nonzero / index_select / shifts / adds / cat /
// scatter tensor operators: ~20 launches per level).
struct OccCount {
    const int16_t *symbols;     // symbol + 1 = the 8 occupancy bits, bit (7 - k) = child k  (or NULL)
    const uint8_t *bits;        // [n][8], non-zero = child k occupied                        (or NULL)
    __host__ __device__ __forceinline__ int operator()(int64_t i) const {
        if (symbols) return __builtin_popcount(((unsigned)symbols[i] + 1u) & 0xffu);
        int c = 0;
        for (int k = 0; k < 8; ++k) c += bits[8 * i + k] != 0;
        return c;
    }
};

__global__ __launch_bounds__(256) void k_octree_children(OccCount occ, const int32_t *__restrict__ pos_incl, const int32_t *__restrict__ coords,
                                                         int64_t n, int64_t m, int32_t fxp_one, int32_t *__restrict__ child_coords,
                                                         int32_t *__restrict__ parent_row, int32_t *__restrict__ octant,
                                                         int32_t *__restrict__ table, int64_t table_rows, uint8_t *__restrict__ bits_out,
                                                         int32_t *__restrict__ bits_fxp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) {
        // zero the padding rows of the table (rows [m, table_rows)): one thread each, past the parents
        const int64_t r = m + (i - n);
        if (table && r < table_rows) {
#pragma unroll
            for (int k = 0; k < 8; ++k) table[8 * r + k] = 0;
        }
        return;
    }
    unsigned mask = 0;
    if (occ.symbols) {
        mask = ((unsigned)occ.symbols[i] + 1u) & 0xffu;                   // bit (7 - k) = child k
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) mask |= (occ.bits[8 * i + k] != 0 ? 1u : 0u) << (7 - k);
    }
    int64_t j = pos_incl[i] - __builtin_popcount(mask);
    int32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (coords) { c0 = coords[4 * i]; c1 = coords[4 * i + 1] << 1; c2 = coords[4 * i + 2] << 1; c3 = coords[4 * i + 3] << 1; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool on = (mask >> (7 - k)) & 1u;
        if (bits_out) bits_out[8 * i + k] = on ? 1 : 0;
        if (bits_fxp) bits_fxp[8 * i + k] = on ? fxp_one : 0;
        if (!on || j >= m) continue;
        if (child_coords) {
            child_coords[4 * j] = c0; child_coords[4 * j + 1] = c1 + (k >> 2); child_coords[4 * j + 2] = c2 + ((k >> 1) & 1);
            child_coords[4 * j + 3] = c3 + (k & 1);
        }
        if (parent_row) parent_row[j] = (int32_t)i;
        if (octant) octant[j] = k;
        if (table) {
#pragma unroll
            for (int q = 0; q < 8; ++q) table[8 * j + q] = q == k ? (int32_t)i + 1 : 0;
        }
        ++j;
    }
}

template <typename InIt>
int64_t scan_bytes(InIt in, int64_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, in, (int32_t *)nullptr, (size_t)n, rocprim::plus<int32_t>());
    return (int64_t)bytes;
}

}  // namespace
}  // namespace fpcc

using namespace fpcc;

extern "C" int fpcc_morton3d_encode(const int32_t *coords, int64_t n, int64_t row_stride, int col_bit0, int col_bit1,
                                    int col_bit2, int64_t *keys_out, void *stream) {
    if (n < 0 || (n > 0 && (!coords || !keys_out))) return fail_arg("morton3d_encode: null pointer");
    if (col_bit0 < 0 || col_bit1 < 0 || col_bit2 < 0 || col_bit0 >= row_stride || col_bit1 >= row_stride ||
        col_bit2 >= row_stride)
        return fail_arg("morton3d_encode: column outside the row");
    if (n == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_morton, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), coords, n,
                       row_stride, col_bit0, col_bit1, col_bit2, keys_out);
    FPCC_LAUNCHED(k_morton);
    return FPCC_OK;
}

namespace {
struct AxisMap { int perm[3]; int flip[3]; };
inline bool same(const AxisMap &a, const AxisMap &b) {
    for (int i = 0; i < 3; ++i) if (a.perm[i] != b.perm[i] || a.flip[i] != b.flip[i]) return false;
    return true;
}
// the 96-entry machine from the geometry (see k_hilbert); states in breadth-first order from the root
void build_hilbert_table(uint8_t (&tab)[96]) {
    static const int base[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 1, 1}, {1, 1, 1}, {1, 0, 1}, {0, 0, 1}};
    static const AxisMap child[8] = {{{2, 0, 1}, {0, 0, 0}}, {{1, 2, 0}, {0, 0, 0}}, {{1, 2, 0}, {0, 0, 0}}, {{0, 1, 2}, {1, 1, 0}},
                                     {{0, 1, 2}, {1, 1, 0}}, {{1, 2, 0}, {0, 1, 1}}, {{1, 2, 0}, {0, 1, 1}}, {{2, 0, 1}, {1, 0, 1}}};
    AxisMap states[12] = {{{0, 1, 2}, {0, 0, 0}}};
    int n_states = 1;
    for (int s = 0; s < n_states && s < 12; ++s) {
        for (int k = 0; k < 8; ++k) {
            int o[3];
            for (int i = 0; i < 3; ++i) o[i] = states[s].flip[i] ^ base[k][states[s].perm[i]];
            AxisMap nxt;
            for (int i = 0; i < 3; ++i) {
                nxt.perm[i] = child[k].perm[states[s].perm[i]];
                nxt.flip[i] = states[s].flip[i] ^ child[k].flip[states[s].perm[i]];
            }
            int id = 0;
            while (id < n_states && !same(states[id], nxt)) ++id;
            if (id == n_states && n_states < 12) states[n_states++] = nxt;
            tab[s * 8 + (o[0] | o[1] << 1 | o[2] << 2)] = static_cast<uint8_t>(id * 8 + k);
        }
    }
}
}  // namespace

extern "C" int fpcc_hilbert3d_encode(const int32_t *coords, int64_t n, int64_t row_stride, int col_x, int col_y, int col_z,
                                     int bits, int64_t *keys_out, void *stream) {
    if (n < 0 || row_stride < 1 || bits < 1 || bits > 21) return fail_arg("hilbert3d_encode: bad sizes (bits must be in [1, 21])");
    if (col_x < 0 || col_y < 0 || col_z < 0 || col_x >= row_stride || col_y >= row_stride || col_z >= row_stride)
        return fail_arg("hilbert3d_encode: column outside the row");
    if (n == 0) return FPCC_OK;
    if (!coords || !keys_out) return fail_arg("hilbert3d_encode: null pointer");
    // g_hilbert is one symbol PER DEVICE: upload on first use on each device, stream-ordered before the kernel
    static std::atomic<bool> uploaded[64];
    static uint8_t tab[96];
    static const bool built = (build_hilbert_table(tab), true);
    (void)built;
    int dev = 0;
    if (int rc = check_hip(hipGetDevice(&dev), "hipGetDevice")) return rc;
    if (dev < 0 || dev >= 64) return fail_arg("hilbert3d_encode: device index out of range");
    if (!uploaded[dev].load(std::memory_order_acquire)) {
        if (int rc = check_hip(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_hilbert), tab, sizeof(tab), 0, hipMemcpyHostToDevice, as_stream(stream)),
                               "hilbert table upload")) return rc;
        if (int rc = check_hip(hipStreamSynchronize(as_stream(stream)), "hilbert table upload")) return rc;
        uploaded[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_hilbert, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), coords, n, row_stride,
                       col_x, col_y, col_z, bits, keys_out);
    FPCC_LAUNCHED(k_hilbert);
    return FPCC_OK;
}

static int check_bits(int level, int bits) {
    if (level < 0 || level > 21 || bits < 1 || bits > 21) return fail_arg("level must be in [0,21], bits in [1,21]");
    return FPCC_OK;
}

extern "C" int fpcc_keys_from_coords(const int32_t *coords, int64_t n, int level, int bits, int64_t *keys_out,
                                     void *stream) {
    if (int rc = check_bits(level, bits)) return rc;
    if (n < 0 || (n > 0 && (!coords || !keys_out))) return fail_arg("keys_from_coords: null pointer");
    if ((reinterpret_cast<uintptr_t>(coords) & 15) != 0) return fail_arg("keys_from_coords: coords must be 16-byte aligned");
    if (n == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_keys_from_coords, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<const int4 *>(coords), n, level, bits, keys_out);
    FPCC_LAUNCHED(k_keys_from_coords);
    return FPCC_OK;
}

extern "C" int fpcc_coords_from_keys(const int64_t *keys, int64_t n, int level, int bits, const int32_t *offset_xyz,
                                     int32_t *coords_out, void *stream) {
    if (int rc = check_bits(level, bits)) return rc;
    if (n < 0 || (n > 0 && (!keys || !coords_out))) return fail_arg("coords_from_keys: null pointer");
    if ((reinterpret_cast<uintptr_t>(coords_out) & 15) != 0) return fail_arg("coords_from_keys: output must be 16-byte aligned");
    if (n == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_coords_from_keys, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), keys, n,
                       level, bits, offset_xyz, reinterpret_cast<int4 *>(coords_out));
    FPCC_LAUNCHED(k_coords_from_keys);
    return FPCC_OK;
}

extern "C" int64_t fpcc_sort_keys(const int64_t *keys_in, int64_t n, int end_bit, int64_t *keys_out, int32_t *perm_out,
                                  void *ws, int64_t ws_bytes, void *stream) {
    if (n < 0 || end_bit < 1 || end_bit > 64) return fail_arg("sort_keys: bad n or end_bit");
    size_t sort_bytes = 0;
    // keys are non-negative (bit 63 clear), so an unsigned sort of the low end_bit bits is an ordinary ascending sort
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)(n > 0 ? n : 1), 0u,
                                             (unsigned)end_bit);
    if (e != hipSuccess) return check_hip(e, "radix_sort_pairs(size query)");
    const int64_t iota_bytes = align_up(4 * (n > 0 ? n : 1), 256);
    const int64_t need = iota_bytes + align_up((int64_t)sort_bytes, 256);
    if (!ws) return need;
    if (ws_bytes < need) { set_error("sort_keys: workspace %lld < %lld", (long long)ws_bytes, (long long)need); return FPCC_E_WORKSPACE; }
    if (n == 0) return FPCC_OK;
    if (!keys_in || !keys_out || !perm_out) return fail_arg("sort_keys: null pointer");
    int32_t *iota = static_cast<int32_t *>(ws);
    void *tmp = static_cast<char *>(ws) + iota_bytes;
    hipLaunchKernelGGL(k_iota, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), iota, n);
    FPCC_LAUNCHED(k_iota);
    FPCC_HIP(rocprim::radix_sort_pairs(tmp, sort_bytes, reinterpret_cast<const uint64_t *>(keys_in),
                                       reinterpret_cast<uint64_t *>(keys_out), (const int32_t *)iota, perm_out,
                                       (size_t)n, 0u, (unsigned)end_bit, as_stream(stream)));
    return FPCC_OK;
}

// shared body of unique / coarsen: head flags on keys >> shift, inclusive scan
static int64_t flags_and_scan(const int64_t *keys, int64_t n, int shift, void *ws, int64_t ws_bytes, void *stream,
                              int32_t **flag_out, int32_t **pos_out, const char *who) {
    const int64_t nn = n > 0 ? n : 1;
    const int64_t arr = align_up(4 * nn, 256);
    const int64_t tmp_bytes = align_up(scan_bytes((const int32_t *)nullptr, nn), 256);
    const int64_t need = 2 * arr + tmp_bytes;
    if (!ws) return need;
    if (ws_bytes < need) { set_error("%s: workspace %lld < %lld", who, (long long)ws_bytes, (long long)need); return FPCC_E_WORKSPACE; }
    int32_t *flag = static_cast<int32_t *>(ws);
    int32_t *pos = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + arr);
    void *tmp = static_cast<char *>(ws) + 2 * arr;
    if (n > 0) {
        hipLaunchKernelGGL(k_head_flags, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), keys, n,
                           shift, flag);
        FPCC_LAUNCHED(k_head_flags);
        size_t tb = (size_t)tmp_bytes;
        FPCC_HIP(rocprim::inclusive_scan(tmp, tb, (const int32_t *)flag, pos, (size_t)n, rocprim::plus<int32_t>(),
                                         as_stream(stream)));
    }
    *flag_out = flag;
    *pos_out = pos;
    return FPCC_OK;
}

extern "C" int64_t fpcc_unique_keys(const int64_t *keys, int64_t n, int64_t *ukeys_out, int32_t *first_out,
                                    int32_t *count_out, void *ws, int64_t ws_bytes, void *stream) {
    if (n < 0) return fail_arg("unique_keys: n < 0");
    int32_t *flag = nullptr, *pos = nullptr;
    if (ws && n > 0 && (!keys || !ukeys_out || !first_out || !count_out)) return fail_arg("unique_keys: null pointer");
    int64_t rc = flags_and_scan(keys, n, 0, ws, ws_bytes, stream, &flag, &pos, "unique_keys");
    if (!ws || rc != FPCC_OK) return rc;
    if (n == 0) { FPCC_HIP(hipMemsetAsync(count_out, 0, 4, as_stream(stream))); return FPCC_OK; }
    hipLaunchKernelGGL(k_unique_scatter, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), keys, n,
                       (const int32_t *)flag, (const int32_t *)pos, ukeys_out, first_out, count_out);
    FPCC_LAUNCHED(k_unique_scatter);
    return FPCC_OK;
}

extern "C" int64_t fpcc_coarsen(const int64_t *keys, int64_t n, int32_t *parent_of, int64_t *pkeys, int32_t *child_row,
                                int32_t *count_out, void *ws, int64_t ws_bytes, void *stream) {
    if (n < 0) return fail_arg("coarsen: n < 0");
    int32_t *flag = nullptr, *pos = nullptr;
    if (ws && n > 0 && (!keys || !parent_of || !pkeys || !child_row || !count_out)) return fail_arg("coarsen: null pointer");
    if (ws && (reinterpret_cast<uintptr_t>(child_row) & 15) != 0) return fail_arg("coarsen: child_row must be 16-byte aligned");
    int64_t rc = flags_and_scan(keys, n, 3, ws, ws_bytes, stream, &flag, &pos, "coarsen");
    if (!ws || rc != FPCC_OK) return rc;
    if (n == 0) { FPCC_HIP(hipMemsetAsync(count_out, 0, 4, as_stream(stream))); return FPCC_OK; }
    hipLaunchKernelGGL(k_coarsen_scatter, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), keys, n,
                       (const int32_t *)flag, (const int32_t *)pos, parent_of, pkeys, child_row, count_out);
    FPCC_LAUNCHED(k_coarsen_scatter);
    return FPCC_OK;
}

extern "C" int64_t fpcc_octree_level(const int64_t *keys, int64_t n, int64_t m, int batch_shift, int64_t *pkeys, int32_t *coords,
                                     int32_t *bits, int32_t *table, int64_t table_rows, int16_t *symbols, void *ws, int64_t ws_bytes,
                                     void *stream) {
    if (n < 0 || m < 0 || m > n || batch_shift < 0 || batch_shift > 62) return fail_arg("octree_level: sizes out of range");
    int32_t *flag = nullptr, *pos = nullptr;
    if (ws && n > 0 && !keys) return fail_arg("octree_level: null pointer");
    if (ws && table && (table_rows < m || (reinterpret_cast<uintptr_t>(table) & 15) != 0))
        return fail_arg("octree_level: the kernel map needs >= m rows and 16-byte alignment");
    if (ws && ((reinterpret_cast<uintptr_t>(coords) & 15) != 0 || (reinterpret_cast<uintptr_t>(bits) & 15) != 0))
        return fail_arg("octree_level: coords and bits must be 16-byte aligned");
    int64_t rc = flags_and_scan(keys, n, 3, ws, ws_bytes, stream, &flag, &pos, "octree_level");
    if (!ws || rc != FPCC_OK) return rc;
    const int64_t spare = table ? table_rows - m : 0;
    const int64_t threads = n > spare ? n : spare;
    if (threads == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_octree_level, dim3(blocks_for(threads, kThreads)), dim3(kThreads), 0, as_stream(stream), keys, n,
                       (const int32_t *)flag, (const int32_t *)pos, m, batch_shift, pkeys, coords, bits, table, table_rows, symbols);
    FPCC_LAUNCHED(k_octree_level);
    return FPCC_OK;
}

// Row counts of ALL coarser levels of a sorted key array in one pass: two neighbouring keys fall into different cells of level l iff
// their highest differing bit lies at or above bit 3 l, so a key pair adds one row to every level l <= top, top = (highest differing
// bit) / 3.  hist[t] counts the pairs with top == t (t clamped to `levels`); rows of level l = 1 + sum_{t >= l} hist[t].
__global__ __launch_bounds__(256) void k_level_histogram(const int64_t *__restrict__ keys, int64_t n, int levels, int32_t *hist) {
    __shared__ int32_t s_hist[32];
    if (threadIdx.x < 32) s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t d = (uint64_t)(keys[i] ^ keys[i - 1]);
        if (d) {
            const int top = (63 - __clzll((long long)d)) / 3;
            atomicAdd(&s_hist[top < levels ? top : levels], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x <= levels && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

extern "C" int64_t fpcc_level_histogram(const int64_t *keys, int64_t n, int levels, int32_t *hist, void *stream) {
    if (n < 0 || levels < 1 || levels > 21) return fail_arg("level_histogram: n < 0 or levels outside 1..21");
    if (!hist || (n > 0 && !keys)) return fail_arg("level_histogram: null pointer");
    FPCC_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * (levels + 1), as_stream(stream)));
    if (n < 2) return FPCC_OK;
    const int64_t blocks = blocks_for(n, 256 * 8);
    hipLaunchKernelGGL(k_level_histogram, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, as_stream(stream), keys, n, levels, hist);
    FPCC_LAUNCHED(k_level_histogram);
    return FPCC_OK;
}

// The same pass for a BATCH of clouds (key = cloud << cloud_shift | Morton code; rows are cloud-major): row counts of every cloud at
// every level.  hist[c][t], t <= levels: key pairs INSIDE cloud c with top == t (a pair that straddles two clouds belongs to neither);
// hist[c][levels + 1]: keys of cloud c.  Rows of cloud c at level l = (hist[c][levels + 1] > 0) + sum_{t >= l} hist[c][t].
__global__ __launch_bounds__(256) void k_level_histogram_clouds(const int64_t *__restrict__ keys, int64_t n, int levels, int cloud_shift,
                                                                int n_clouds, int32_t *hist) {
    extern __shared__ int32_t s_cloud_hist[];
    const int width = levels + 2, total = n_clouds * width;
    for (int i = threadIdx.x; i < total; i += blockDim.x) s_cloud_hist[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = (uint64_t)keys[i];
        const int c = (int)(k >> cloud_shift);
        if (c >= n_clouds) continue;                                  // reported by the host wrapper (counts do not add up to n)
        atomicAdd(&s_cloud_hist[c * width + levels + 1], 1);
        if (i > 0) {
            const uint64_t d = k ^ (uint64_t)keys[i - 1];
            if (d && (d >> cloud_shift) == 0) {
                const int top = (63 - __clzll((long long)d)) / 3;
                atomicAdd(&s_cloud_hist[c * width + (top < levels ? top : levels)], 1);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += blockDim.x)
        if (s_cloud_hist[i]) atomicAdd(&hist[i], s_cloud_hist[i]);
}

extern "C" int64_t fpcc_level_histogram_clouds(const int64_t *keys, int64_t n, int levels, int cloud_shift, int n_clouds, int32_t *hist,
                                               void *stream) {
    if (n < 0 || levels < 1 || levels > 21 || n_clouds < 1 || n_clouds > 64 || cloud_shift < 3 || cloud_shift > 63)
        return fail_arg("level_histogram_clouds: n < 0, levels outside 1..21, clouds outside 1..64 or bad shift");
    if (!hist || (n > 0 && !keys)) return fail_arg("level_histogram_clouds: null pointer");
    const int total = n_clouds * (levels + 2);
    FPCC_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * total, as_stream(stream)));
    if (n < 1) return FPCC_OK;
    const int64_t blocks = blocks_for(n, 256 * 8);
    hipLaunchKernelGGL(k_level_histogram_clouds, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), sizeof(int32_t) * total,
                       as_stream(stream), keys, n, levels, cloud_shift, n_clouds, hist);
    FPCC_LAUNCHED(k_level_histogram_clouds);
    return FPCC_OK;
}

extern "C" int64_t fpcc_refine(const int64_t *pkeys, int64_t m, const uint8_t *mask, int64_t *keys_out,
                               int32_t *parent_of, int32_t *child_row, int32_t *count_out, void *ws, int64_t ws_bytes,
                               void *stream) {
    if (m < 0) return fail_arg("refine: m < 0");
    const int64_t n_cand = 8 * m;
    const int64_t nn = n_cand > 0 ? n_cand : 1;
    auto in = rocprim::make_transform_iterator(mask, ByteToInt());
    const int64_t arr = align_up(4 * nn, 256);
    const int64_t tmp_bytes = align_up(scan_bytes(in, nn), 256);
    const int64_t need = arr + tmp_bytes;
    if (!ws) return need;
    if (ws_bytes < need) { set_error("refine: workspace %lld < %lld", (long long)ws_bytes, (long long)need); return FPCC_E_WORKSPACE; }
    if (m == 0) { FPCC_HIP(hipMemsetAsync(count_out, 0, 4, as_stream(stream))); return FPCC_OK; }
    if (!pkeys || !mask || !keys_out || !parent_of || !child_row || !count_out) return fail_arg("refine: null pointer");
    int32_t *pos = static_cast<int32_t *>(ws);
    void *tmp = static_cast<char *>(ws) + arr;
    size_t tb = (size_t)tmp_bytes;
    FPCC_HIP(rocprim::inclusive_scan(tmp, tb, in, pos, (size_t)n_cand, rocprim::plus<int32_t>(), as_stream(stream)));
    if (reinterpret_cast<uintptr_t>(child_row) & 15) return fail_arg("refine: child_row must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(mask) & 7)
        hipLaunchKernelGGL(k_refine_scatter<false>, dim3(blocks_for(m, kThreads)), dim3(kThreads), 0, as_stream(stream), pkeys,
                           n_cand, mask, (const int32_t *)pos, keys_out, parent_of, child_row, count_out);
    else
        hipLaunchKernelGGL(k_refine_scatter<true>, dim3(blocks_for(m, kThreads)), dim3(kThreads), 0, as_stream(stream), pkeys,
                           n_cand, mask, (const int32_t *)pos, keys_out, parent_of, child_row, count_out);
    FPCC_LAUNCHED(k_refine_scatter);
    return FPCC_OK;
}

extern "C" int64_t fpcc_octree_children(const int16_t *symbols, const uint8_t *bits, const int32_t *coords, int64_t n, int64_t m,
                                        int32_t fxp_one, int32_t *child_coords, int32_t *parent_row, int32_t *octant, int32_t *table,
                                        int64_t table_rows, uint8_t *bits_out, int32_t *bits_fxp, void *ws, int64_t ws_bytes,
                                        void *stream) {
    if (n < 0 || m < 0 || table_rows < 0 || (table && table_rows < m)) return fail_arg("octree_children: bad sizes");
    if ((symbols != nullptr) == (bits != nullptr) && ws) return fail_arg("octree_children: give symbols or bits (one of them)");
    OccCount occ{symbols, bits};
    const int64_t nn = n > 0 ? n : 1;
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), occ);
    const int64_t arr = align_up(4 * nn, 256);
    const int64_t tmp_bytes = align_up(scan_bytes(in, nn), 256);
    const int64_t need = arr + tmp_bytes;
    if (!ws) return need;
    if (ws_bytes < need) { set_error("octree_children: workspace %lld < %lld", (long long)ws_bytes, (long long)need); return FPCC_E_WORKSPACE; }
    if (n == 0) return FPCC_OK;
    int32_t *pos = static_cast<int32_t *>(ws);
    void *tmp = static_cast<char *>(ws) + arr;
    size_t tb = (size_t)tmp_bytes;
    FPCC_HIP(rocprim::inclusive_scan(tmp, tb, in, pos, (size_t)n, rocprim::plus<int32_t>(), as_stream(stream)));
    const int64_t pad = table && table_rows > m ? table_rows - m : 0;
    hipLaunchKernelGGL(k_octree_children, dim3(blocks_for(n + pad, 256)), dim3(256), 0, as_stream(stream), occ, (const int32_t *)pos,
                       coords, n, m, fxp_one, child_coords, parent_row, octant, table, table_rows, bits_out, bits_fxp);
    FPCC_LAUNCHED(k_octree_children);
    return FPCC_OK;
}

extern "C" int fpcc_nbr27_search(const int64_t *keys, int64_t n, int bits, int32_t *nbr, void *stream) {
    if (bits < 1 || bits > 21) return fail_arg("nbr27_search: bits must be in [1,21]");
    if (n < 0 || (n > 0 && (!keys || !nbr))) return fail_arg("nbr27_search: null pointer");
    if (n == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_nbr27_search, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), keys, n, bits,
                       nbr);
    FPCC_LAUNCHED(k_nbr27_search);
    return FPCC_OK;
}

extern "C" int fpcc_nbr27_from_parent(const int64_t *keys, const int32_t *parent_of, int64_t n,
                                      const int32_t *parent_nbr, int64_t m, const int32_t *child_row, int32_t *nbr,
                                      void *stream) {
    if (n < 0 || m < 0 || (n > 0 && (!parent_nbr || !nbr))) return fail_arg("nbr27_from_parent: null pointer");
    if (!keys && (parent_of || child_row)) return fail_arg("nbr27_from_parent: keys may only be omitted for a full generated set");
    if (!parent_of && !(child_row == nullptr && n == 8 * m))
        return fail_arg("nbr27_from_parent: parent_of may only be omitted for a full generated set (n == 8m, child_row NULL)");
    if (n == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_nbr27_from_parent<0>, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), keys,
                       parent_of, n, parent_nbr, m, child_row, nbr, (int32_t *)nullptr, (uint32_t *)nullptr);
    FPCC_LAUNCHED(k_nbr27_from_parent);
    return FPCC_OK;
}

extern "C" int fpcc_nbr27_from_parent_ex(const int64_t *keys, const int32_t *parent_of, int64_t n, const int32_t *parent_nbr, int64_t m,
                                         const int32_t *child_row, int32_t *nbr, int32_t *rows_out, uint32_t *masks_out, void *stream) {
    if (n < 0 || m < 0 || (n > 0 && (!parent_nbr || !nbr))) return fail_arg("nbr27_from_parent_ex: null pointer");
    if (!keys && (parent_of || child_row)) return fail_arg("nbr27_from_parent_ex: keys may only be omitted for a full generated set");
    if (!parent_of && !(child_row == nullptr && n == 8 * m))
        return fail_arg("nbr27_from_parent_ex: parent_of may only be omitted for a full generated set (n == 8m, child_row NULL)");
    if (rows_out && (reinterpret_cast<uintptr_t>(rows_out) & 15)) return fail_arg("nbr27_from_parent_ex: rows_out must be 16-byte aligned");
    if (n == 0) return FPCC_OK;
    if (rows_out || masks_out)
        hipLaunchKernelGGL(k_nbr27_from_parent<2>, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), keys,
                           parent_of, n, parent_nbr, m, child_row, nbr, rows_out, masks_out);
    else
        hipLaunchKernelGGL(k_nbr27_from_parent<0>, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), keys,
                           parent_of, n, parent_nbr, m, child_row, nbr, rows_out, masks_out);
    FPCC_LAUNCHED(k_nbr27_from_parent);
    return FPCC_OK;
}

extern "C" int fpcc_mask27_from_parent(const int64_t *keys, const int32_t *parent_of, int64_t n, const int32_t *parent_nbr,
                                       int64_t m, const int32_t *child_row, uint32_t *masks_out, void *stream) {
    if (n < 0 || m < 0 || (n > 0 && (!parent_nbr || !masks_out))) return fail_arg("mask27_from_parent: null pointer");
    if (!keys && (parent_of || child_row)) return fail_arg("mask27_from_parent: keys may only be omitted for a full generated set");
    if (!parent_of && !(child_row == nullptr && n == 8 * m))
        return fail_arg("mask27_from_parent: parent_of may only be omitted for a full generated set (n == 8m, child_row NULL)");
    if (n == 0) return FPCC_OK;
    hipLaunchKernelGGL(k_nbr27_from_parent<1>, dim3(blocks_for(n, kThreads)), dim3(kThreads), 0, as_stream(stream), keys,
                       parent_of, n, parent_nbr, m, child_row, reinterpret_cast<int32_t *>(masks_out), (int32_t *)nullptr, (uint32_t *)nullptr);
    FPCC_LAUNCHED(k_nbr27_from_parent);
    return FPCC_OK;
}

extern "C" int64_t fpcc_compact_coords(const int64_t *pkeys, int64_t m, const uint8_t *mask, int level, int bits,
                                       const int32_t *offset_xyz, int32_t *xyz_out, int32_t *count_out, void *ws,
                                       int64_t ws_bytes, void *stream) {
    if (m < 0) return fail_arg("compact_coords: m < 0");
    if (int rc = check_bits(level, bits)) return rc;
    const int64_t n_cand = 8 * m;
    const int64_t nn = n_cand > 0 ? n_cand : 1;
    auto in = rocprim::make_transform_iterator(mask, ByteToInt());
    const int64_t arr = align_up(4 * nn, 256);
    const int64_t tmp_bytes = align_up(scan_bytes(in, nn), 256);
    const int64_t need = arr + tmp_bytes;
    if (!ws) return need;
    if (ws_bytes < need) { set_error("compact_coords: workspace %lld < %lld", (long long)ws_bytes, (long long)need); return FPCC_E_WORKSPACE; }
    if (m == 0) { FPCC_HIP(hipMemsetAsync(count_out, 0, 4, as_stream(stream))); return FPCC_OK; }
    if (!pkeys || !mask || !xyz_out || !count_out) return fail_arg("compact_coords: null pointer");
    int32_t *pos = static_cast<int32_t *>(ws);
    void *tmp = static_cast<char *>(ws) + arr;
    size_t tb = (size_t)tmp_bytes;
    FPCC_HIP(rocprim::inclusive_scan(tmp, tb, in, pos, (size_t)n_cand, rocprim::plus<int32_t>(), as_stream(stream)));
    hipLaunchKernelGGL(k_compact_coords, dim3(blocks_for(n_cand, kThreads)), dim3(kThreads), 0, as_stream(stream), pkeys,
                       n_cand, mask, (const int32_t *)pos, level, bits, offset_xyz, xyz_out, count_out);
    FPCC_LAUNCHED(k_compact_coords);
    return FPCC_OK;
}
