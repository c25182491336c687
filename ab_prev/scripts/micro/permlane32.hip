// v_permlane32_swap_b32 semantics probe (gfx950): prints what lanes 0, 1, 32, 33 hold after swap(a = lane, b = 100 + lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0 = %3u  r1 = %3u\n", l, h[l], h[64 + l]);
    return 0;
}
