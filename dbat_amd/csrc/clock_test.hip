#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_clock(unsigned long long *out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
__global__ void k_burn(double *p, int iters) {
    double x = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;
    p[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
__global__ void k_tiny(double *p) { if (threadIdx.x == 0) p[0] += 1.0; }
int main() {
    unsigned long long *d; CK(hipMalloc(&d, 64)); double *p; CK(hipMalloc(&p, 8 * 256 * 4096));
    CK(hipMemset(p, 0, 8 * 256 * 4096));
    unsigned long long h[3];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto meas = [&](const char *tag) {
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d, 20000);
        CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        printf("%s: shader cycles %llu over %.1f us => %.0f MHz\n", tag, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
    };
    meas("cold");
    meas("second");
    // chain of 200 dependent tiny kernels
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, 0, p);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("200 tiny dependent kernels: %.3f ms => %.2f us each\n", ms, ms * 5);
    meas("after tiny chain");
    hipLaunchKernelGGL(k_burn, dim3(4096), dim3(256), 0, 0, p, 2000000);
    meas("right after 4096-block burn");
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, 0, p);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("200 tiny dependent kernels after burn: %.3f ms => %.2f us each\n", ms, ms * 5);
    meas("end");
    return 0;
}
