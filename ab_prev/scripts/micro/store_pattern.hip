// Micro-benchmark: HBM store throughput of the conv epilogue's write pattern on MI355X.
// A wave-wide 16-byte-per-lane store covers (1024 / SEG) rows x SEG contiguous bytes of a [M][512 B] output
// (SEG = 128: the 64-channel output tile of conv_igemm on a 256-channel tensor; 256: a 128-channel tile; 512: full rows).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern scripts/micro/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int SEG>
__global__ __launch_bounds__(256) void store_kernel(uint4 *y, long M, int rows_per_wg) {
    constexpr int LPR = SEG / 16;            // lanes per row segment
    constexpr int NT = 512 / SEG;            // column tiles
    const int tile_n = blockIdx.x % NT;
    const long m0 = (long)(blockIdx.x / NT) * rows_per_wg;
    const int tid = threadIdx.x;
    const uint4 v = make_uint4(tid, tid, tid, tid);
    for (int r = tid / LPR; r < rows_per_wg; r += 256 / LPR) {
        const long m = m0 + r;
        if (m < M) y[m * 32 + tile_n * LPR + tid % LPR] = v;
    }
}

template <int SEG>
float run(uint4 *y, long M, int rows_per_wg) {
    const int nt = 512 / SEG;
    const int wgs = (int)((M + rows_per_wg - 1) / rows_per_wg) * nt;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(store_kernel<SEG>, dim3(wgs), dim3(256), 0, 0, y, M, rows_per_wg);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(store_kernel<SEG>, dim3(wgs), dim3(256), 0, 0, y, M, rows_per_wg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const long M = 907500;                   // layer-1 pixels at 75 clips
    uint4 *y; hipMalloc(&y, M * 512);
    const double gb = M * 512 / 1e9;
    for (int rows : {128, 256, 1024}) {
        float a = run<128>(y, M, rows), b = run<256>(y, M, rows), c = run<512>(y, M, rows);
        printf("rows/WG %4d: 128-B segments %.1f us (%.2f TB/s)   256-B %.1f us (%.2f TB/s)   512-B full rows %.1f us (%.2f TB/s)\n", rows,
               a * 1e3, gb / a / 1e9 * 1e3, b * 1e3, gb / b / 1e9 * 1e3, c * 1e3, gb / c / 1e9 * 1e3);
    }
    return 0;
}
