// Micro-benchmark: what one SIMD sustains on v_mfma_f32_32x32x2_f32 beside the load streams of the sparse-conv kernels.
//   mode 0: MFMAs only (operands in registers)
//   mode 1: + B stream: 16 x 16-byte loads per 64 MFMAs from an L2-resident 1.77 MB buffer (packed-weight pattern)
//   mode 2: + A gather: 4 x 16-byte loads per 64 MFMAs, 32 random 128-byte rows of a large buffer per stage
//   mode 3: mode 2 with sequential rows
// grid = 256 CUs x waves_per_simd blocks of 256 threads.  Prints TFLOP/s, cycles per MFMA per SIMD and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 3) void k_probe(const float *__restrict__ wp, const float *__restrict__ x, const int *__restrict__ rows,
                                                  int n_rows, int stages, float *out, unsigned long long *clk) {
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x16 acc[4];
    for (int nb = 0; nb < 4; ++nb) for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    f32x4 ra[4], rb[4][4];
    for (int g = 0; g < 4; ++g) { ra[g] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + g); for (int nb = 0; nb < 4; ++nb) rb[g][nb] = f32x4{.5f, .25f, .125f, 1.f} * (float)(nb + 1); }
    const float *bp0 = wp + lane * 4;
    unsigned seed = wave_id * 2654435761u + 12345u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE >= 1) {
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 2) ra[g] = *reinterpret_cast<const f32x4 *>(x + (long)rows[(wave_id * 32 + li) % n_rows] * 128 + 8 * g + 4 * lh);
            for (int nb = 0; nb < 4; ++nb) { __builtin_amdgcn_sched_barrier(0); rb[g][nb] = *reinterpret_cast<const f32x4 *>(bp0 + (g * 4 + nb) * 256); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int s = 0; s < stages; ++s) {
        seed = seed * 1664525u + 1013904223u;
        const float *bp = bp0 + ((MODE == 10 || MODE == 11) ? 0 : (long)((seed >> 8) % 108) * 4096);   // one of 27 x 4 chunks of 16 KB
        const float *ap = x;
        if (MODE == 8 || MODE == 11) ap = x + (long)((((seed >> 4) + li * 977u) * 2654435761u >> 7) % 2048) * 128 + ((seed >> 20) & 3) * 32 + 4 * lh;
        if (MODE == 9) ap = x + (long)((seed >> 4) % 2048) * 128 + ((seed >> 20) & 3) * 32 + 4 * lh;
        if (MODE == 2) ap = x + (long)rows[((seed >> 4) + li * 977u) % n_rows] * 128 + ((seed >> 20) & 3) * 32 + 4 * lh;
        if (MODE == 3) ap = x + (long)(((seed >> 4) % (n_rows - 32)) + li) * 128 + ((seed >> 20) & 3) * 32 + 4 * lh;
        __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 av = ra[g];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, rb[g][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, rb[g][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, rb[g][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, rb[g][nb].w, acc[nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x6);
            if (MODE >= 2 && MODE != 10) ra[g] = *reinterpret_cast<const f32x4 *>(ap + 8 * g);
            if (MODE >= 1) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) { __builtin_amdgcn_sched_barrier(0x6); rb[g][nb] = *reinterpret_cast<const f32x4 *>(bp + (g * 4 + nb) * 256); }
            }
            __builtin_amdgcn_sched_barrier(0x6);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("" ::: "memory");
    float sum = 0.f;
    for (int nb = 0; nb < 4; ++nb) for (int r = 0; r < 16; ++r) sum += acc[nb][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (lane == 0) { clk[wave_id * 2] = t1 - t0; clk[wave_id * 2 + 1] = r1 - r0; }
}

// mode 4: B direct from L2 (16 loads / stage), A by LINE-COALESCED loads (8 lanes x 16 B = one 128-byte row segment, 8 rows
// per instruction) staged through a private, XOR-swizzled LDS tile and read back in MFMA operand layout
// mode 7: mode 1 with the B loads in scalar-base + 32-bit lane offset form
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_probe2(const float *__restrict__ wp, const float *__restrict__ x, const int *__restrict__ rows,
                                                   int n_rows, int stages, float *out, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) float s_a[4][2][32 * 32];
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, wv = threadIdx.x >> 6;
    const int wave_id = blockIdx.x * 4 + wv;
    const int r8 = lane >> 3, q = lane & 7;
    f32x16 acc[4];
    for (int nb = 0; nb < 4; ++nb) for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    f32x4 ra[4], stg[4], rb[4][4];
    for (int g = 0; g < 4; ++g) { ra[g] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + g); stg[g] = ra[g]; for (int nb = 0; nb < 4; ++nb) rb[g][nb] = f32x4{.5f, .25f, .125f, 1.f} * (float)(nb + 1); }
    const unsigned lane_off = lane * 16;
    unsigned seed = wave_id * 2654435761u + 12345u;
    float *mine = &s_a[wv][0][0];
    // write position of my piece (row 8j + r8, piece q) and read position (row li, piece 2g + lh), swizzled by (row >> 1) & 7
    auto wr = [&](int buf, int j) -> f32x4 * { const int row = 8 * j + r8; return reinterpret_cast<f32x4 *>(mine + buf * 1024 + row * 32 + ((q ^ ((row >> 1) & 7)) << 2)); };
    auto rd = [&](int buf, int g) -> const f32x4 * { return reinterpret_cast<const f32x4 *>(mine + buf * 1024 + li * 32 + (((2 * g + lh) ^ ((li >> 1) & 7)) << 2)); };
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < stages; ++s) {
        seed = seed * 1664525u + 1013904223u;
        const char *bpc = reinterpret_cast<const char *>(wp) + (size_t)((seed >> 8) % 108) * 16384;      // wave-uniform
        const float *ap[4];
        if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ap[j] = x + (long)rows[((seed >> 4) + (8 * j + r8) * 977u) % n_rows] * 128 + ((seed >> 20) & 3) * 32 + 4 * q;
        }
        __builtin_amdgcn_sched_barrier(0x6);
        if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) stg[j] = *reinterpret_cast<const f32x4 *>(ap[j]);      // chunk s+2
        }
        __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 av = ra[g];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, rb[g][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, rb[g][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, rb[g][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, rb[g][nb].w, acc[nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x6);
            if (MODE == 4) ra[g] = *rd((s + 1) & 1, g);                                  // chunk s+1, written at the end of stage s-1
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) { __builtin_amdgcn_sched_barrier(0x6); rb[g][nb] = *reinterpret_cast<const f32x4 *>(bpc + (g * 4 + nb) * 1024 + lane_off); }
            __builtin_amdgcn_sched_barrier(0x6);
        }
        if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *wr(s & 1, j) = stg[j];                          // chunk s+2 -> the buffer chunk s occupied
        }
        __builtin_amdgcn_sched_barrier(0x6);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int nb = 0; nb < 4; ++nb) for (int r = 0; r < 16; ++r) sum += acc[nb][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (lane == 0) { clk[wave_id * 2] = t1 - t0; clk[wave_id * 2 + 1] = r1 - r0; }
}

// mode 12 / 13: ONE accumulator (a one-block unit of the wave kernel): 16 dependent MFMAs per stage; mode 12 adds the unit's loads
// (4 gathers of an L2-resident table + 4 B loads), mode 13 has none; mode 14: two accumulators, 32 MFMAs, 4 + 8 loads
template <int MODE>
__global__ __launch_bounds__(256, 4) void k_probe3(const float *__restrict__ wp, const float *__restrict__ x, int stages, float *out, unsigned long long *clk) {
    constexpr int NBW = MODE == 14 ? 2 : 1;
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x16 acc[NBW];
    for (int nb = 0; nb < NBW; ++nb) for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    f32x4 ra[4], rb[4][NBW];
    for (int g = 0; g < 4; ++g) { ra[g] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + g); for (int nb = 0; nb < NBW; ++nb) rb[g][nb] = f32x4{.5f, .25f, .125f, 1.f} * (float)(nb + 1); }
    const float *bp0 = wp + lane * 4;
    unsigned seed = wave_id * 2654435761u + 12345u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < stages; ++s) {
        seed = seed * 1664525u + 1013904223u;
        const float *bp = bp0 + (long)((seed >> 8) % 108) * 4096;
        const float *ap = x + (long)((((seed >> 4) + li * 977u) * 2654435761u >> 7) % 2048) * 128 + ((seed >> 20) & 3) * 32 + 4 * lh;
        __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 av = ra[g];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, rb[g][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, rb[g][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, rb[g][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, rb[g][nb].w, acc[nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x6);
            if (MODE != 13) {
                ra[g] = *reinterpret_cast<const f32x4 *>(ap + 8 * g);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) { __builtin_amdgcn_sched_barrier(0x6); rb[g][nb] = *reinterpret_cast<const f32x4 *>(bp + (g * 4 + nb) * 256); }
            }
            __builtin_amdgcn_sched_barrier(0x6);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int nb = 0; nb < NBW; ++nb) for (int r = 0; r < 16; ++r) sum += acc[nb][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (lane == 0) { clk[wave_id * 2] = t1 - t0; clk[wave_id * 2 + 1] = r1 - r0; }
}

int main(int argc, char **argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 400;
    const int n_rows = 272488;
    float *wp, *x, *out; int *rows; unsigned long long *clk;
    hipMalloc(&wp, 27 * 128 * 128 * 4 + 65536); hipMalloc(&x, (size_t)n_rows * 128 * 4); hipMalloc(&rows, n_rows * 4);
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 256 * 8 * 4 * 16);
    hipMemset(wp, 0, 27 * 128 * 128 * 4 + 65536); hipMemset(x, 0, (size_t)n_rows * 128 * 4);
    std::vector<int> h(n_rows); unsigned s = 1; for (int i = 0; i < n_rows; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 4) % n_rows; }
    hipMemcpy(rows, h.data(), n_rows * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode : {0, 13, 12, 14, 1, 2})
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int grid = 256 * wps;
            float best = 1e30f;
            std::vector<unsigned long long> c(grid * 8);
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                if (mode == 4) hipLaunchKernelGGL(k_probe2<4>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                if (mode == 7) hipLaunchKernelGGL(k_probe2<7>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                if (mode == 8) hipLaunchKernelGGL(k_probe<8>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                if (mode == 9) hipLaunchKernelGGL(k_probe<9>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                if (mode == 10) hipLaunchKernelGGL(k_probe<10>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                if (mode == 11) hipLaunchKernelGGL(k_probe<11>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                if (mode == 12) hipLaunchKernelGGL(k_probe3<12>, dim3(grid), dim3(256), 0, 0, wp, x, stages, out, clk);
                if (mode == 13) hipLaunchKernelGGL(k_probe3<13>, dim3(grid), dim3(256), 0, 0, wp, x, stages, out, clk);
                if (mode == 14) hipLaunchKernelGGL(k_probe3<14>, dim3(grid), dim3(256), 0, 0, wp, x, stages, out, clk);
                if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(grid), dim3(256), 0, 0, wp, x, rows, n_rows, stages, out, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            hipMemcpy(c.data(), clk, grid * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double cyc = 0, real = 0; for (int w = 0; w < grid * 4; ++w) { cyc += c[w * 2]; real += c[w * 2 + 1]; }
            cyc /= grid * 4; real /= grid * 4;
            const int per_stage = mode == 12 || mode == 13 ? 16 : mode == 14 ? 32 : 64;
            const double mfmas = (double)stages * per_stage * grid * 4;
            const double tf = mfmas * 4096 / (best * 1e-3) / 1e12;
            printf("mode %d waves/SIMD %d: %.3f ms  %.1f TFLOP/s  %.1f SIMD cycles per MFMA  clock %.2f GHz (in-kernel)\n", mode, wps, best, tf,
                   cyc / (stages * (double)per_stage) / wps, cyc / real * 0.1);
        }
    return 0;
}
