// How fast do the f64 matrix cores really go?  One workgroup per CU, NW waves per SIMD,
// each issuing independent v_mfma_f64_16x16x4_f64 back to back; reports shader cycles per
// MFMA (s_memtime), the shader clock implied by the 100 MHz real-time counter, and the
// chip-wide TFLOP/s.   hipcc --offload-arch=gfx950 -O3 bench/mfma_clock.hip -o /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(int iters, unsigned long long *out, double *sink) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    const int iters = 20000;
    for (int waves_per_simd : {1, 2}) {
        const int threads = 256 * waves_per_simd, blocks = 256;
        unsigned long long *out; double *sink;
        hipMalloc(&out, blocks * 16); hipMalloc(&sink, (size_t)blocks * threads * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, 100, out, sink);   // warm-up
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, iters, out, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
        cyc /= blocks; wall /= blocks;
        const double n_mfma = 8.0 * iters;                       // per wave
        const double flops = n_mfma * 2048.0 * (threads / 64) * blocks;
        printf("waves/SIMD %d: s_memtime ticks per MFMA per wave %.1f, ticks per 10ns %.2f, kernel %.3f ms, %.1f TFLOP/s f64\n",
               waves_per_simd, cyc / n_mfma, cyc / wall, ms, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
