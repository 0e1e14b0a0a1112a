// Does f64 vector arithmetic run beside f64 MFMA on one SIMD of gfx950?  (measurement only; bench/r05_coexec.sh)
//   mode 0: waves 0-3 issue N independent v_mfma_f64_16x16x4_f64, waves 4-7 leave
//   mode 1: waves 0-3 leave, waves 4-7 issue N x 8 independent v_fma_f64
//   mode 2: both (wave w and wave w+4 share a SIMD: a workgroup's waves go round the four SIMDs)
//   mode 3: ONE wave per SIMD does both, interleaved (8 v_fma_f64 after every MFMA)
// Prints cycles (s_memtime) per MFMA and per v_fma for each mode.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(int n, long long *out, double *sink) {
    const int w = threadIdx.x >> 6;
    const bool mf = (MODE == 5 || MODE == 7) ? true : (MODE == 4 ? false : w < 4);
    if (MODE == 6 && !mf) return;
    if (MODE == 6 || MODE == 7) {
        __syncthreads();
        const long long u0 = __builtin_readcyclecounter();
        d4 c = {0, 0, 0, 0};
#pragma unroll 1
        for (int i = 0; i < n; i += 4) {
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(threadIdx.x * 1e-3, 1.0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(threadIdx.x * 1e-3, 1.0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(threadIdx.x * 1e-3, 1.0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(threadIdx.x * 1e-3, 1.0, c, 0, 0, 0);
        }
        const long long u1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = u1 - u0;
        sink[blockIdx.x * 512 + threadIdx.x] = c[0] + c[1] + c[2] + c[3];
        return;
    }
    if (MODE == 0 && !mf) return;
    if (MODE == 1 && mf) return;
    if (MODE == 3 && !mf) return;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + a;
    double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 3 || (mf && MODE != 1 && MODE != 4)) {
#pragma unroll 1
        for (int i = 0; i < n; i += 4) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 3) { f0 = __builtin_fma(f0, b, a); f1 = __builtin_fma(f1, b, a); f2 = __builtin_fma(f2, b, a); f3 = __builtin_fma(f3, b, a); f4 = __builtin_fma(f4, b, a); f5 = __builtin_fma(f5, b, a); f6 = __builtin_fma(f6, b, a); f7 = __builtin_fma(f7, b, a); }
            __builtin_amdgcn_sched_barrier(0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 3) { f0 = __builtin_fma(f0, b, a); f1 = __builtin_fma(f1, b, a); f2 = __builtin_fma(f2, b, a); f3 = __builtin_fma(f3, b, a); f4 = __builtin_fma(f4, b, a); f5 = __builtin_fma(f5, b, a); f6 = __builtin_fma(f6, b, a); f7 = __builtin_fma(f7, b, a); }
            __builtin_amdgcn_sched_barrier(0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 3) { f0 = __builtin_fma(f0, b, a); f1 = __builtin_fma(f1, b, a); f2 = __builtin_fma(f2, b, a); f3 = __builtin_fma(f3, b, a); f4 = __builtin_fma(f4, b, a); f5 = __builtin_fma(f5, b, a); f6 = __builtin_fma(f6, b, a); f7 = __builtin_fma(f7, b, a); }
            __builtin_amdgcn_sched_barrier(0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 3) { f0 = __builtin_fma(f0, b, a); f1 = __builtin_fma(f1, b, a); f2 = __builtin_fma(f2, b, a); f3 = __builtin_fma(f3, b, a); f4 = __builtin_fma(f4, b, a); f5 = __builtin_fma(f5, b, a); f6 = __builtin_fma(f6, b, a); f7 = __builtin_fma(f7, b, a); }
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
            f0 = __builtin_fma(f0, b, a); f1 = __builtin_fma(f1, b, a); f2 = __builtin_fma(f2, b, a); f3 = __builtin_fma(f3, b, a);
            f4 = __builtin_fma(f4, b, a); f5 = __builtin_fma(f5, b, a); f6 = __builtin_fma(f6, b, a); f7 = __builtin_fma(f7, b, a);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

int main() {
    const int n = 4096, nb = 256;
    long long *out; double *sink;
    hipMalloc(&out, nb * 8 * sizeof(long long)); hipMalloc(&sink, nb * 512 * sizeof(double));
    std::vector<long long> h(nb * 8);
    const char *names[8] = {"MFMA alone (waves 0-3)", "v_fma_f64 alone (waves 4-7)", "both, two waves per SIMD", "one wave: MFMA + 8 v_fma interleaved", "v_fma_f64, two waves per SIMD", "MFMA, two waves per SIMD", "MFMA, ONE accumulator chain, one wave per SIMD", "MFMA, ONE accumulator chain, two waves per SIMD"};
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(out, 0, nb * 8 * sizeof(long long));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(512), 0, 0, n, out, sink);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(512), 0, 0, n, out, sink);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(512), 0, 0, n, out, sink);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(512), 0, 0, n, out, sink);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(nb), dim3(512), 0, 0, n, out, sink);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(nb), dim3(512), 0, 0, n, out, sink);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(nb), dim3(512), 0, 0, n, out, sink);
            if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(nb), dim3(512), 0, 0, n, out, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), out, nb * 8 * sizeof(long long), hipMemcpyDeviceToHost);
        double m = 0, v = 0; int cm = 0, cv = 0;
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) if (h[b * 8 + w]) { if (mode == 5 || mode == 7 || (mode != 4 && w < 4)) { m += h[b * 8 + w]; ++cm; } else { v += h[b * 8 + w]; ++cv; } }
        printf("%-40s", names[mode]);
        if (cm) printf("  MFMA waves: %.1f ticks per MFMA%s", m / cm / n, mode == 3 ? " (+ 8 v_fma)" : "");
        if (cv) printf("  VALU waves: %.2f ticks per v_fma_f64", v / cv / n / 8);
        printf("\n");
    }
    // the counter's rate: s_memtime ticks per second against the host clock is not needed -- ratios only
    return 0;
}
