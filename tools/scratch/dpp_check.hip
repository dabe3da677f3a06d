// Checks the lane direction of the wave-shift DPP controls on the running GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned x = threadIdx.x + 100;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(7u, x, 0x138, 0xF, 0xF, false);       // wave_shr:1
    o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(7u, x, 0x130, 0xF, 0xF, false);  // wave_shl:1
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("shr: lane0=%u lane1=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[1], h[31], h[32], h[63]);
    printf("shl: lane0=%u lane1=%u lane31=%u lane32=%u lane62=%u lane63=%u\n", h[64], h[65], h[95], h[96], h[126], h[127]);
    return 0;
}
