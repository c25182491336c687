// Small fp32 "head" ops of the I3D wrapper for gfx950: Linear (+ folded BatchNorm1d + ReLU)
// and row-wise L2 normalisation (reference: I3Res50.fc large_i3d.py:147,245;
// mlp.forward aux_code/model_loaders.py:250-254). These are negligible FLOPs
// (B x 2048 x 512) and stay in exact fp32: one wavefront per output element, 16-byte
// coalesced loads along K, wavefront-shuffle tree reduction.
#include "common.h"

namespace tedspad {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void linear_kernel(const float *x, const float *w, const float *scale, const float *shift,
                                                      float *y, int B, int K, int N, int relu) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= B * N) return;
    const int n = wave % N, b = wave / N;
    const float *px = x + (size_t)b * K, *pw = w + (size_t)n * K;
    float acc = 0.f;
    const int K4 = (K % 4 == 0) ? K : 0;  // vector path needs 16-byte rows
    for (int k = lane * 4; k < K4; k += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(px + k), c = *reinterpret_cast<const f32x4 *>(pw + k);
        acc += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
    }
    for (int k = K4 + lane; k < K; k += 64) acc += px[k] * pw[k];
    acc = wave_sum(acc);
    if (lane == 0) {
        float v = acc * (scale ? scale[n] : 1.f) + (shift ? shift[n] : 0.f);
        if (relu) v = __builtin_fmaxf(v, 0.f);
        y[(size_t)b * N + n] = v;
    }
}

// The same product for a SMALL batch (B <= 16 rows: the mlp heads on 8-24 embeddings): one wavefront per output COLUMN n walks K once and keeps the B partial
// sums -- the weight row is read once instead of once per batch row (the kernel above launched B x N waves that each re-read their weight row: 60-100 us for the
// privacy branch's 2048 x 2048 layer, 3 ms per training iteration over its 52 calls). Same lane / k assignment and the same shuffle tree per output.
template <int BM>
__global__ __launch_bounds__(256) void linear_cols_kernel(const float *x, const float *w, const float *scale, const float *shift,
                                                           float *y, int B, int K, int N, int relu) {
    const int n = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const float *pw = w + (size_t)n * K;
    float acc[BM];
#pragma unroll
    for (int b = 0; b < BM; ++b) acc[b] = 0.f;
    const int K4 = (K % 4 == 0) ? K : 0;
    for (int k = lane * 4; k < K4; k += 256) {
        const f32x4 c = *reinterpret_cast<const f32x4 *>(pw + k);
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (b < B) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(x + (size_t)b * K + k);
                acc[b] += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
            }
    }
    for (int k = K4 + lane; k < K; k += 64) {
        const float c = pw[k];
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (b < B) acc[b] += x[(size_t)b * K + k] * c;
    }
    const float sc = scale ? scale[n] : 1.f, sh = shift ? shift[n] : 0.f;
#pragma unroll
    for (int b = 0; b < BM; ++b)
        if (b < B) {
            const float t = wave_sum(acc[b]);
            if (lane == 0) {
                float v = t * sc + sh;
                if (relu) v = __builtin_fmaxf(v, 0.f);
                y[(size_t)b * N + n] = v;
            }
        }
}

// ... and for a SHORT reduction (K <= 32: the weight gradients dW = dY^T . X of those heads, where the reduction runs over the batch): one THREAD per output
// column n keeps its weight row in registers and walks a block of 32 rows of x (all lanes read the same row: a broadcast), writing coalesced rows of y.
// (One wave per output spent a whole wavefront on a 12-term dot product: 4.2 M waves for the 2048 x 2048 layer.)
template <int KM>
__global__ __launch_bounds__(256) void linear_smallk_kernel(const float *x, const float *w, const float *scale, const float *shift,
                                                             float *y, int B, int K, int N, int relu) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b0 = blockIdx.y * 32;
    if (n >= N) return;
    float c[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) c[k] = k < K ? w[(size_t)n * K + k] : 0.f;
    const float sc = scale ? scale[n] : 1.f, sh = shift ? shift[n] : 0.f;
    for (int b = b0; b < min(B, b0 + 32); ++b) {
        const float *px = x + (size_t)b * K;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) acc += px[k] * c[k];
        float v = acc * sc + sh;
        if (relu) v = __builtin_fmaxf(v, 0.f);
        y[(size_t)b * N + n] = v;
    }
}

__global__ __launch_bounds__(256) void l2norm_kernel(const float *x, float *y, int B, int N, float eps) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= B) return;
    const float *px = x + (size_t)wave * N;
    float s = 0.f;
    for (int i = lane; i < N; i += 64) s += px[i] * px[i];
    s = wave_sum(s);
    const float inv = 1.f / __builtin_fmaxf(sqrtf(s), eps);
    for (int i = lane; i < N; i += 64) y[(size_t)wave * N + i] = px[i] * inv;
}

// BatchNorm1d in train mode on a (B, C) fp32 matrix (B is tiny: one thread per feature walks the batch).
__global__ void bn1d_train_fwd_kernel(const float *x, const float *gamma, const float *beta, float eps, float momentum, float *rmean,
                                      float *rvar, float *y, float *mean_o, float *invstd_o, int B, int C, int relu) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f, ss = 0.f;
    for (int b = 0; b < B; ++b) { const float v = x[(size_t)b * C + c]; s += v; }
    const float mean = s / B;
    for (int b = 0; b < B; ++b) { const float d = x[(size_t)b * C + c] - mean; ss += d * d; }
    const float var = ss / B;
    const float invstd = rsqrtf(var + eps);
    for (int b = 0; b < B; ++b) {
        float v = (x[(size_t)b * C + c] - mean) * invstd * gamma[c] + beta[c];
        if (relu) v = __builtin_fmaxf(v, 0.f);
        y[(size_t)b * C + c] = v;
    }
    mean_o[c] = mean; invstd_o[c] = invstd;
    if (rmean) {
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (B > 1 ? var * B / (B - 1) : var);
    }
}

__global__ void bn1d_train_bwd_kernel(const float *dy, const float *x, const float *y, const float *mean, const float *invstd,
                                      const float *gamma, float *dx, float *dgamma, float *dbeta, int B, int C, int relu) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float sb = 0.f, sg = 0.f;
    for (int b = 0; b < B; ++b) {
        float g = dy[(size_t)b * C + c];
        if (relu && !(y[(size_t)b * C + c] > 0.f)) g = 0.f;
        sb += g; sg += g * (x[(size_t)b * C + c] - mean[c]) * invstd[c];
    }
    dbeta[c] = sb; dgamma[c] = sg;
    for (int b = 0; b < B; ++b) {
        float g = dy[(size_t)b * C + c];
        if (relu && !(y[(size_t)b * C + c] > 0.f)) g = 0.f;
        const float xh = (x[(size_t)b * C + c] - mean[c]) * invstd[c];
        dx[(size_t)b * C + c] = gamma[c] * invstd[c] * (g - sb / B - xh * sg / B);
    }
}

// y = x / max(|x|, eps)  ->  dx = (dy - y (y . dy)) / max(|x|, eps)   (rows with |x| < eps: dx = dy / eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *x, const float *dy, float *dx, int B, int N, float eps) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= B) return;
    const float *px = x + (size_t)wave * N, *pg = dy + (size_t)wave * N;
    float s = 0.f, d = 0.f;
    for (int i = lane; i < N; i += 64) { s += px[i] * px[i]; d += px[i] * pg[i]; }
    s = wave_sum(s); d = wave_sum(d);
    const float nr = sqrtf(s);
    if (nr < eps) {
        for (int i = lane; i < N; i += 64) dx[(size_t)wave * N + i] = pg[i] / eps;
    } else {
        const float inv = 1.f / nr;
        for (int i = lane; i < N; i += 64) dx[(size_t)wave * N + i] = (pg[i] - px[i] * d * inv * inv) * inv;
    }
}

__global__ void mul_kernel(const float *a, const float *b, float *o, long n, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) o[i] = a[i] * b[i] * scale;
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_bn1d_train_fwd(const float *x, const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                          float *running_var, float *y, float *mean, float *invstd, int32_t B, int32_t C, int32_t relu, void *stream) {
    TS_REQUIRE(x && gamma && beta && y && mean && invstd && B > 0 && C > 0, "tedspad_bn1d_train_fwd: bad arguments");
    hipLaunchKernelGGL(bn1d_train_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, momentum, running_mean,
                       running_var, y, mean, invstd, B, C, relu);
    return check_launch("tedspad_bn1d_train_fwd");
}

extern "C" int32_t tedspad_bn1d_train_bwd(const float *dy, const float *x, const float *y, const float *mean, const float *invstd, const float *gamma,
                                          float *dx, float *dgamma, float *dbeta, int32_t B, int32_t C, int32_t relu, void *stream) {
    TS_REQUIRE(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && B > 0 && C > 0 && (!relu || y), "tedspad_bn1d_train_bwd: bad arguments");
    hipLaunchKernelGGL(bn1d_train_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, x, y, mean, invstd, gamma, dx, dgamma, dbeta, B, C, relu);
    return check_launch("tedspad_bn1d_train_bwd");
}

extern "C" int32_t tedspad_l2_normalize_rows_bwd(const float *x, const float *dy, float *dx, int32_t B, int32_t N, float eps, void *stream) {
    TS_REQUIRE(x && dy && dx && B > 0 && N > 0, "tedspad_l2_normalize_rows_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, dx, B, N, eps);
    return check_launch("tedspad_l2_normalize_rows_bwd");
}

extern "C" int32_t tedspad_mul_f32(const float *a, const float *b, float *out, int64_t n, float scale, void *stream) {
    TS_REQUIRE(a && b && out && n > 0, "tedspad_mul_f32: bad arguments");
    long g = (n + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(mul_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n, scale);
    return check_launch("tedspad_mul_f32");
}

extern "C" int32_t tedspad_linear_fwd(const float *x, const float *w, const float *scale, const float *shift, float *y,
                                      int32_t B, int32_t K, int32_t N, int32_t relu, void *stream) {
    TS_REQUIRE(x && w && y && B > 0 && K > 0 && N > 0, "tedspad_linear_fwd: bad arguments");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)w) % 16 == 0, "tedspad_linear_fwd: x and w must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (K <= 32 && B >= 64) {            // short reduction, many rows: a thread per output column
        const dim3 g((unsigned)((N + 255) / 256), (unsigned)((B + 31) / 32));
        if (K <= 16) hipLaunchKernelGGL(linear_smallk_kernel<16>, g, dim3(256), 0, s, x, w, scale, shift, y, B, K, N, relu);
        else hipLaunchKernelGGL(linear_smallk_kernel<32>, g, dim3(256), 0, s, x, w, scale, shift, y, B, K, N, relu);
        return check_launch("tedspad_linear_fwd");
    }
    if (B <= 16 && K >= 256) {           // small batch: a wavefront per output column, the weight row read once
        const dim3 g((unsigned)((N + 3) / 4));
        if (B <= 8) hipLaunchKernelGGL(linear_cols_kernel<8>, g, dim3(256), 0, s, x, w, scale, shift, y, B, K, N, relu);
        else hipLaunchKernelGGL(linear_cols_kernel<16>, g, dim3(256), 0, s, x, w, scale, shift, y, B, K, N, relu);
        return check_launch("tedspad_linear_fwd");
    }
    const long waves = (long)B * N;
    hipLaunchKernelGGL(linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, w, scale, shift, y, B, K, N, relu);
    return check_launch("tedspad_linear_fwd");
}

extern "C" int32_t tedspad_l2_normalize_rows(const float *x, float *y, int32_t B, int32_t N, float eps, void *stream) {
    TS_REQUIRE(x && y && B > 0 && N > 0, "tedspad_l2_normalize_rows: bad arguments");
    hipLaunchKernelGGL(l2norm_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, B, N, eps);
    return check_launch("tedspad_l2_normalize_rows");
}
