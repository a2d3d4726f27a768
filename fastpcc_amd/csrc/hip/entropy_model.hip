// libfpcc_hip.so -- rate term of the noisy deep-factorised entropy bottleneck as ONE kernel (training path).
//
//   logp(y) = log( cdf(y + h) - cdf(y - h) ),  cdf = sigmoid(logits_cdf),  logits_cdf = a 1-3-3-3-3-1 network per channel
//   with softplus(weights), bias and tanh(factor) * tanh(.) gates
// (/root/reference/lib/entropy_models/distributions/deep_factorized.py:24-40, uniform_noise.py:30-63, the bits loss of
// continuous_batched.py:62-69).  The reference evaluates this with ~60 small tensor kernels per call and twice that again
// in autograd; the codec calls it for every coded pyramid level, which made ~2000 launches of a training step.
// Here a thread keeps the 58 transformed parameters of its channel in registers, walks its elements, evaluates both
// logit chains, the numerically stable log-difference (survival functions right of the median) and back-propagates
// by hand: d sum(logp) / dy per element, and the 58 parameter gradients accumulated per thread, then reduced per block.
// Row blocks write partial sums; a second kernel adds them in ascending block order (reproducible).
#include "common.h"

#include <algorithm>

namespace fpcc {
namespace {

constexpr int kW[5] = {3, 9, 9, 9, 3};            // weights per layer (f_out x f_in)
constexpr int kFo[5] = {3, 3, 3, 3, 1};
constexpr int kFi[5] = {1, 3, 3, 3, 3};
constexpr int kNW = 33, kNB = 13, kNF = 12, kNP = kNW + kNB + kNF;     // 58 parameters per channel
constexpr int kOut = kNP + 1;                                          // + sum of log-probabilities

struct DfacArgs {
    const float *y; int64_t n; int c; int ldy;
    const float *w[5]; const float *b[5]; const float *f[4];           // raw parameters [c][f_out][f_in] / [c][f_out]
    float half;
    float *dy; int lddy;                                                // d sum(logp) / d y
    float *partial;                                                     // [blocks][c][kOut]
};

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float logsigmoid_f(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }

struct Chain {
    float h[5][3];      // layer inputs (h[0][0] = v)
    float t[4][3];      // tanh of the gated pre-activations
    float logit;
};

__device__ __forceinline__ void chain_forward(const float (&W)[kNW], const float (&B)[kNB], const float (&A)[kNF], float v, Chain &s) {
    s.h[0][0] = v;
    int wo = 0, bo = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= kFo[i]) continue;
            float z = B[bo + j];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < kFi[i]) z = fmaf(W[wo + j * kFi[i] + k], s.h[i][k], z);
            if (i < 4) {
                const float th = tanhf(z);
                s.t[i][j] = th;
                s.h[i + 1][j] = fmaf(A[3 * i + j], th, z);
            } else {
                s.logit = z;
            }
        }
        wo += kW[i];
        bo += kFo[i];
    }
}

// back-propagate g = d(logp)/d(logit) through one chain; returns d(logp)/dv
__device__ __forceinline__ float chain_backward(const float (&W)[kNW], const float (&A)[kNF], const Chain &s, float g,
                                                float (&gW)[kNW], float (&gB)[kNB], float (&gA)[kNF]) {
    float dh[3] = {g, 0.0f, 0.0f};
    int wo = kNW, bo = kNB;
#pragma unroll
    for (int i = 4; i >= 0; --i) {
        wo -= kW[i];
        bo -= kFo[i];
        float dz[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= kFo[i]) continue;
            if (i < 4) {
                const float th = s.t[i][j];
                gA[3 * i + j] = fmaf(dh[j], th, gA[3 * i + j]);
                dz[j] = dh[j] * fmaf(A[3 * i + j], 1.0f - th * th, 1.0f);
            } else {
                dz[j] = dh[j];
            }
            gB[bo + j] += dz[j];
        }
        float nd[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= kFo[i]) continue;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < kFi[i]) {
                    gW[wo + j * kFi[i] + k] = fmaf(dz[j], s.h[i][k], gW[wo + j * kFi[i] + k]);
                    nd[k] = fmaf(W[wo + j * kFi[i] + k], dz[j], nd[k]);
                }
        }
        dh[0] = nd[0]; dh[1] = nd[1]; dh[2] = nd[2];
    }
    return dh[0];
}

__global__ __launch_bounds__(256) void k_dfac_bits(DfacArgs a) {
    const int ch = blockIdx.y;
    float W[kNW], B[kNB], A[kNF], gW[kNW], gB[kNB], gA[kNF];
    float sW[kNW], dA[kNF];                         // derivatives of the parameter transforms
    {
        int wo = 0, bo = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int q = 0; q < 9; ++q)
                if (q < kW[i]) {
                    const float raw = a.w[i][ch * kW[i] + q];
                    W[wo + q] = softplus_f(raw);
                    sW[wo + q] = sigmoid_f(raw);
                }
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < kFo[i]) B[bo + q] = a.b[i][ch * kFo[i] + q];
            if (i < 4) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float th = tanhf(a.f[i][ch * 3 + q]);
                    A[3 * i + q] = th;
                    dA[3 * i + q] = 1.0f - th * th;
                }
            }
            wo += kW[i];
            bo += kFo[i];
        }
    }
#pragma unroll
    for (int q = 0; q < kNW; ++q) gW[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < kNB; ++q) gB[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < kNF; ++q) gA[q] = 0.0f;
    float sum_logp = 0.0f;

    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < a.n; r += (int64_t)gridDim.x * 256) {
        const float yv = a.y[r * a.ldy + ch];
        Chain hi, lo;
        chain_forward(W, B, A, yv + a.half, hi);
        chain_forward(W, B, A, yv - a.half, lo);
        // right of the median: survival functions (logsigmoid(-x)); left: cdfs
        const bool right = hi.logit > 0.0f;
        const float big = right ? logsigmoid_f(-lo.logit) : logsigmoid_f(hi.logit);
        const float small = right ? logsigmoid_f(-hi.logit) : logsigmoid_f(lo.logit);
        const float e = expf(small - big);
        sum_logp += log1pf(-e) + big;
        const float d_big = 1.0f / (1.0f - e), d_small = -e / (1.0f - e);
        float g_hi, g_lo;
        if (right) {
            g_lo = d_big * -sigmoid_f(lo.logit);
            g_hi = d_small * -sigmoid_f(hi.logit);
        } else {
            g_hi = d_big * sigmoid_f(-hi.logit);
            g_lo = d_small * sigmoid_f(-lo.logit);
        }
        const float dv = chain_backward(W, A, hi, g_hi, gW, gB, gA) + chain_backward(W, A, lo, g_lo, gW, gB, gA);
        if (a.dy) a.dy[r * a.lddy + ch] = dv;
    }

    // block reduction of the 58 gradients + the log-probability sum: wave shuffles, then LDS across the 4 waves
    __shared__ float part[4][kOut];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto reduce = [&](float v, int slot) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) part[wave][slot] = v;
    };
#pragma unroll
    for (int q = 0; q < kNW; ++q) reduce(gW[q] * sW[q], q);                  // through softplus
#pragma unroll
    for (int q = 0; q < kNB; ++q) reduce(gB[q], kNW + q);
#pragma unroll
    for (int q = 0; q < kNF; ++q) reduce(gA[q] * dA[q], kNW + kNB + q);      // through tanh
    reduce(sum_logp, kNP);
    __syncthreads();
    float *dst = a.partial + ((int64_t)blockIdx.x * a.c + ch) * kOut;
    for (int q = threadIdx.x; q < kOut; q += 256) dst[q] = (part[0][q] + part[1][q]) + (part[2][q] + part[3][q]);
}

__global__ void k_dfac_reduce(const float *__restrict__ partial, int blocks, int c, float *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;          // (channel, slot)
    if (e >= c * kOut) return;
    float t = 0.0f;
    for (int b = 0; b < blocks; ++b) t += partial[(int64_t)b * c * kOut + e];
    out[e] = t;
}

int dfac_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024)); }

}  // namespace
}  // namespace fpcc

using namespace fpcc;

extern "C" int64_t fpcc_deep_factorized_ws_bytes(int64_t n, int c) {
    if (n < 0 || c < 1) return FPCC_E_ARG;
    return (int64_t)dfac_blocks(n) * c * kOut * 4;
}

extern "C" int fpcc_deep_factorized_bits_f32(const float *y, int64_t n, int c, int ldy, const float *const *weights,
                                             const float *const *biases, const float *const *factors, float half_width,
                                             float *dy, int lddy, float *out, void *ws, int64_t ws_bytes, void *stream) {
    if (n < 0 || c < 1 || c > 65535 || ldy < c || (dy && lddy < c)) return fail_arg("deep_factorized_bits: sizes out of range");
    if (!out || !weights || !biases || !factors || (n > 0 && !y)) return fail_arg("deep_factorized_bits: null pointer");
    hipStream_t s = as_stream(stream);
    if (n == 0) return check_hip(hipMemsetAsync(out, 0, (size_t)c * kOut * 4, s), "hipMemsetAsync");
    const int blocks = dfac_blocks(n);
    if (!ws || ws_bytes < (int64_t)blocks * c * kOut * 4) return fail_arg("deep_factorized_bits: workspace too small");
    DfacArgs a{y, n, c, ldy, {}, {}, {}, half_width, dy, lddy, static_cast<float *>(ws)};
    for (int i = 0; i < 5; ++i) {
        if (!weights[i] || !biases[i] || (i < 4 && !factors[i])) return fail_arg("deep_factorized_bits: null parameter pointer");
        a.w[i] = weights[i];
        a.b[i] = biases[i];
        if (i < 4) a.f[i] = factors[i];
    }
    hipLaunchKernelGGL(k_dfac_bits, dim3(blocks, c), dim3(256), 0, s, a);
    if (int rc = check_hip(hipGetLastError(), "k_dfac_bits")) return rc;
    hipLaunchKernelGGL(k_dfac_reduce, dim3(blocks_for((int64_t)c * kOut, 256)), dim3(256), 0, s, static_cast<const float *>(ws), blocks,
                       c, out);
    return check_hip(hipGetLastError(), "k_dfac_reduce");
}

// ---------------------------------------------------------------------------------------------------------------
// Rate term of the scale-indexed noisy normal (the Gaussian-conditional bottleneck, training):
//      s = exp(a + b i),   lp = log( Phi((y + h) / s) - Phi((y - h) / s) ),   out = sum lp,  dy = d lp / dy,  di = d lp / di
// log Phi in the three segments of distributions/special_math.py:138-258 (x > 5: -Phi(-x); x < -10: asymptotic series of
// order 3; else log Phi(x)), the difference taken on the survival side right of the median
// (distributions/uniform_noise.py:36-63).  The derivatives are analytic, in the log domain so that a probability that
// underflows does not turn them into inf * 0:   d lp / dy = (phi(z+) - phi(z-)) / (s p) = (e^(l+ - lp) - e^(l- - lp)) / s,
// d lp / ds = -(z+ phi(z+) - z- phi(z-)) / (s p),  l+- = log phi(z+-).
namespace fpcc {
namespace {

__device__ __forceinline__ float ndtr_f(float x) {
    const float h = 0.70710678118654752f;
    const float w = x * h, z = fabsf(w);
    const float y = z < h ? 1.0f + erff(w) : (w > 0.0f ? 2.0f - erfcf(z) : erfcf(z));
    return 0.5f * y;
}

__device__ __forceinline__ float log_ndtr_f(float x) {
    if (x > 5.0f) return -ndtr_f(-x);
    if (x > -10.0f) return logf(ndtr_f(x));
    const float x2 = x * x;
    const float series = 1.0f + 3.0f / (x2 * x2) - (1.0f / x2 + 15.0f / (x2 * x2 * x2));
    return -0.5f * x2 - logf(-x) - 0.91893853320467274f + logf(series);
}

__global__ __launch_bounds__(256) void k_noisy_normal_bits(const float *__restrict__ y, const float *__restrict__ index, int64_t n,
                                                          float a, float b, float h, float *__restrict__ dy,
                                                          float *__restrict__ di, float *__restrict__ partial) {
    __shared__ float s_part[4];
    float acc = 0.0f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float yy = y[e];
        const float s = expf(a + b * index[e]);
        const float inv = 1.0f / s;
        const float zp = (yy + h) * inv, zm = (yy - h) * inv;
        const float logcdf_p = log_ndtr_f(zp), logsf_p = log_ndtr_f(-zp);
        const bool right = logsf_p < logcdf_p;
        const float big = right ? log_ndtr_f(-zm) : logcdf_p;
        const float small = right ? logsf_p : log_ndtr_f(zm);
        const float lp = big + log1pf(-expf(small - big));
        acc += lp;
        const float ep = expf(-0.5f * zp * zp - 0.91893853320467274f - lp);
        const float em = expf(-0.5f * zm * zm - 0.91893853320467274f - lp);
        if (dy) dy[e] = (ep - em) * inv;
        if (di) di[e] = -b * (zp * ep - zm * em);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

__global__ void k_sum_partials(const float *__restrict__ partial, int blocks, float *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float t = 0.0f;
        for (int i = 0; i < blocks; ++i) t += partial[i];
        out[0] = t;
    }
}

}  // namespace
}  // namespace fpcc

extern "C" int64_t fpcc_noisy_normal_ws_bytes(int64_t n) { return n < 0 ? FPCC_E_ARG : (int64_t)dfac_blocks(n) * 4; }

extern "C" int fpcc_noisy_normal_bits_f32(const float *y, const float *index, int64_t n, float log_scale_offset,
                                          float log_scale_factor, float half_width, float *dy, float *dindex, float *out,
                                          void *ws, int64_t ws_bytes, void *stream) {
    if (n < 0) return fail_arg("noisy_normal_bits: n < 0");
    if (!out || (n > 0 && (!y || !index))) return fail_arg("noisy_normal_bits: null pointer");
    hipStream_t s = as_stream(stream);
    if (n == 0) return check_hip(hipMemsetAsync(out, 0, 4, s), "hipMemsetAsync");
    const int blocks = dfac_blocks(n);
    if (!ws || ws_bytes < (int64_t)blocks * 4) return fail_arg("noisy_normal_bits: workspace too small");
    hipLaunchKernelGGL(k_noisy_normal_bits, dim3(blocks), dim3(256), 0, s, y, index, n, log_scale_offset, log_scale_factor,
                       half_width, dy, dindex, static_cast<float *>(ws));
    if (int rc = check_hip(hipGetLastError(), "k_noisy_normal_bits")) return rc;
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, s, static_cast<const float *>(ws), blocks, out);
    return check_hip(hipGetLastError(), "k_sum_partials");
}
