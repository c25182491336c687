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

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_linear_fwd(const float *x, const float *w, const float *scale, const float *shift, float *y,
                                      int32_t B, int32_t K, int32_t N, int32_t relu, void *stream) {
    TS_REQUIRE(x && w && y && B > 0 && K > 0 && N > 0, "tedspad_linear_fwd: bad arguments");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)w) % 16 == 0, "tedspad_linear_fwd: x and w must be 16-byte aligned");
    const long waves = (long)B * N;
    hipLaunchKernelGGL(linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y, B, K, N, relu);
    return check_launch("tedspad_linear_fwd");
}

extern "C" int32_t tedspad_l2_normalize_rows(const float *x, float *y, int32_t B, int32_t N, float eps, void *stream) {
    TS_REQUIRE(x && y && B > 0 && N > 0, "tedspad_l2_normalize_rows: bad arguments");
    hipLaunchKernelGGL(l2norm_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, B, N, eps);
    return check_launch("tedspad_l2_normalize_rows");
}
