// In which order does v_mfma_f32_16x16x4_f32 add its four products?  (The codec's summation orders are FMA chains; a 16 x 16 tile
// would shorten the dependent chain of the small pyramid levels fourfold, but only if its result can be written as a chain.)
// One wave, A[16][4] x B[4][16] + C with random values of mixed magnitude; the device result of every output element is compared
// with host evaluations: the 24 sequential fused-multiply-add chains over the permutations of k, pairwise trees, and unfused forms.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_order_probe mfma_order_probe.hip && ./mfma_order_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float *a, const float *b, const float *c, float *d) {
    // 16x16x4: lane l holds A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16]; C/D: lane l, register r -> row 4 (l / 16) + r, column l % 16
    const int l = threadIdx.x;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[(4 * (l / 16) + r) * 16 + (l % 16)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(l % 16) * 4 + l / 16], b[(l / 16) * 16 + (l % 16)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(4 * (l / 16) + r) * 16 + (l % 16)] = acc[r];
}

__global__ void k32(const float *a, const float *b, const float *c, float *d) {
    // 32x32x2: lane l holds A[i = l % 32][k = l / 32], B[k][j = l % 32]; D register r -> row (r & 3) + 8 (r >> 2) + 4 (l / 32), column l % 32
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c[((r & 3) + 8 * (r >> 2) + 4 * (l / 32)) * 32 + (l % 32)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(l % 32) * 2 + l / 32], b[(l / 32) * 32 + (l % 32)], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (l / 32)) * 32 + (l % 32)] = acc[r];
}

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

int main() {
    srand(7);
    auto rnd = [] { float m = (float)rand() / RAND_MAX * 2 - 1; int e = rand() % 24 - 12; return ldexpf(m, e); };
    const int trials = 200;
    int perm[24][4], np = 0;
    int p[4] = {0, 1, 2, 3};
    do { memcpy(perm[np++], p, sizeof p); } while (std::next_permutation(p, p + 4));
    std::vector<int> chain_ok(24, 0);
    int tree01_23 = 0, unfused = 0, total = 0, ok32 = 0, total32 = 0;
    float *da, *db, *dc, *dd;
    hipMalloc(&da, 4096); hipMalloc(&db, 4096); hipMalloc(&dc, 4096); hipMalloc(&dd, 4096);
    for (int t = 0; t < trials; ++t) {
        float a[64], b[64], c[256], d[256];
        for (auto &v : a) v = rnd();
        for (auto &v : b) v = rnd();
        for (auto &v : c) v = rnd();
        hipMemcpy(da, a, sizeof a, hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof b, hipMemcpyHostToDevice);
        hipMemcpy(dc, c, sizeof c, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d, dd, sizeof d, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            ++total;
            const float got = d[i * 16 + j];
            for (int q = 0; q < 24; ++q) {
                float s = c[i * 16 + j];
                for (int kk = 0; kk < 4; ++kk) s = fmaf(a[i * 4 + perm[q][kk]], b[perm[q][kk] * 16 + j], s);
                chain_ok[q] += bits(s) == bits(got);
            }
            const double e01 = (double)a[i * 4] * b[j] + (double)a[i * 4 + 1] * b[16 + j], e23 = (double)a[i * 4 + 2] * b[32 + j] + (double)a[i * 4 + 3] * b[48 + j];
            tree01_23 += bits((float)((double)c[i * 16 + j] + (e01 + e23))) == bits(got);
            float u = c[i * 16 + j];
            for (int kk = 0; kk < 4; ++kk) u = u + a[i * 4 + kk] * b[kk * 16 + j];
            unfused += bits(u) == bits(got);
        }
        // 32x32x2: chain k = 0, 1
        float a2[64], b2[64], c2[1024], d2[1024];
        for (auto &v : a2) v = rnd();
        for (auto &v : b2) v = rnd();
        for (auto &v : c2) v = rnd();
        hipMemcpy(da, a2, sizeof a2, hipMemcpyHostToDevice); hipMemcpy(db, b2, sizeof b2, hipMemcpyHostToDevice);
        hipMemcpy(dc, c2, sizeof c2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d2, dd, sizeof d2, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            ++total32;
            float s = fmaf(a2[i * 2 + 1], b2[32 + j], fmaf(a2[i * 2], b2[j], c2[i * 32 + j]));
            ok32 += bits(s) == bits(d2[i * 32 + j]);
        }
    }
    printf("v_mfma_f32_32x32x2_f32: sequential fma chain k = 0, 1 matches %d of %d elements\n", ok32, total32);
    printf("v_mfma_f32_16x16x4_f32 over %d elements:\n", total);
    for (int q = 0; q < 24; ++q)
        if (chain_ok[q] > total / 2 || q == 0) printf("  fma chain in k order %d%d%d%d: %d match\n", perm[q][0], perm[q][1], perm[q][2], perm[q][3], chain_ok[q]);
    int best = (int)(std::max_element(chain_ok.begin(), chain_ok.end()) - chain_ok.begin());
    printf("  best chain order %d%d%d%d: %d of %d;  exact pair tree (c + ((p0 + p1) + (p2 + p3))): %d;  unfused chain: %d\n", perm[best][0], perm[best][1],
           perm[best][2], perm[best][3], chain_ok[best], total, tree01_23, unfused);
    return 0;
}
