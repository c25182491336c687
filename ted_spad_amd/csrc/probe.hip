// Sustained shader clock under matrix load (bench.py `roofline.peak_at_clock`): every CU runs a dependent-free MFMA loop on random-looking f16 operands for a few
// milliseconds; a workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around it. The chip lowers its clock under MFMA load
// (MI355X_MICROARCH.md, DVFS give-back), so the nominal 2.4 GHz peak is not what a kernel can be measured against on a given box.
#include "common.h"

namespace tedspad {
namespace {

__global__ __launch_bounds__(256) void clock_probe_kernel(int iters, unsigned long long *out) {
    const int lane = threadIdx.x & 63;
    half8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.01f * (float)((lane * 7 + i * 13 + blockIdx.x) % 61) - 0.3f);
        b[i] = (_Float16)(0.02f * (float)((lane * 11 + i * 5) % 53) - 0.5f);
    }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (s == 123.456f) out[0] = 0;      // keeps the accumulators alive
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_clock_probe(int32_t iters, int32_t workgroups, void *out, void *stream) {
    TS_REQUIRE(iters > 0 && workgroups > 0 && out, "tedspad_clock_probe: bad arguments");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, (unsigned long long *)out);
    return check_launch("tedspad_clock_probe");
}
