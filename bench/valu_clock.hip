// Issue / latency facts of gfx950 that the evaluation chains of k_build_sig live on (round 3):
//   * cycles per dependent v_fma_f64 with 1, 2, 4 independent chains per wave, 1 and 2 waves per SIMD;
//   * v_rcp_f64, v_sqrt_f64, ds_read_b64 (dependent), ds_bpermute_b32 (dependent);
//   * one wave multiplying (v_mfma_f64_16x16x4_f64 back to back) beside one wave of v_fma_f64 on the same SIMD:
//     do the matrix pipe and the f64 vector pipe overlap?
//   hipcc --offload-arch=gfx950 -O3 bench/valu_clock.hip -o bench/valu_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NC>
__device__ __forceinline__ double fma_chain(int iters, double a, double b) {
    double x[NC];
    for (int c = 0; c < NC; ++c) x[c] = a + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) x[c] = __builtin_fma(x[c], b, a);
    }
    double s = 0;
    for (int c = 0; c < NC; ++c) s += x[c];
    return s;
}

// mode 0..2: 1, 2, 4 fma chains; 3: rcp chain; 4: sqrt chain; 5: LDS pointer chase (ds_read_b64); 6: bpermute chain;
// 7: odd waves multiply, even waves run ONE fma chain; 8: odd waves multiply, even waves idle (exit at once);
// 9: all waves: 4 MFMAs then 32 dependent fmas, interleaved by the compiler as it likes
__global__ void k(int mode, int iters, unsigned long long *out, double *sink) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (double)((i * 17 + 5) & 1023);
    const double a = 1e-9 * threadIdx.x, b = 1.0 - 1e-12;
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    double s = 0;
    if (mode == 0) s = fma_chain<1>(iters, a, b);
    else if (mode == 1) s = fma_chain<2>(iters, a, b);
    else if (mode == 2) s = fma_chain<4>(iters, a, b);
    else if (mode == 3) { double x = 1.5 + a; for (int it = 0; it < 8 * iters; ++it) x = __builtin_amdgcn_rcp(x) + 0.5; s = x; }
    else if (mode == 4) { double x = 1.5 + a; for (int it = 0; it < 8 * iters; ++it) x = __builtin_amdgcn_sqrt(x) + 0.5; s = x; }
    else if (mode == 5) { int i = threadIdx.x; for (int it = 0; it < 8 * iters; ++it) i = (int)lds[i & 1023]; s = i; }
    else if (mode == 6) { int v = threadIdx.x; for (int it = 0; it < 8 * iters; ++it) v = __builtin_amdgcn_ds_bpermute(4 * ((v + 1) & 63), v); s = v; }
    else if (mode == 7 || mode == 8) {
        if ((wave >> 2) & 1) {          // waves 4..7: the second wave of every SIMD (waves go round the SIMDs)
            d4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            for (int i = 0; i < 8; ++i) s += acc[i][0];
        } else if (mode == 7) s = fma_chain<1>(iters * 8, a, b);       // 64 fmas per 8 MFMAs of the neighbour
    } else if (mode == 9) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
        double x = a;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 32; ++u) x = __builtin_fma(x, b, a);
        }
        s = x + acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = c1 - c0;
}

int main() {
    const int iters = 4000, blocks = 256;
    unsigned long long *out; double *sink;
    hipMalloc(&out, blocks * 16 * 8); hipMalloc(&sink, (size_t)blocks * 512 * 8);
    const char *names[] = {"fma x1 chain", "fma x2 chains", "fma x4 chains", "rcp_f64 chain (+add)", "sqrt_f64 chain (+add)",
                           "ds_read_b64 chase", "ds_bpermute chain", "mfma wave beside fma wave", "mfma wave alone", "4 mfma + 32 fma in one wave"};
    for (int mode = 0; mode < 10; ++mode)
        for (int wps : {1, 2}) {
            if (mode >= 7 && mode <= 8 && wps == 1) continue;
            const int threads = 256 * wps;
            hipMemset(out, 0, blocks * 16 * 8);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, mode, 10, out, sink);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, mode, iters, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks * 16);
            hipMemcpy(h.data(), out, blocks * 16 * 8, hipMemcpyDeviceToHost);
            double lo = 0, hi = 0;      // waves 0..3 and 4..7
            for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += h[b * 16 + w] / (4.0 * blocks);
            const double nops = mode <= 2 ? 8.0 * iters * (1 << mode) : (mode == 9 ? 36.0 * iters : 8.0 * iters);
            if (mode == 7 || mode == 8)
                printf("%-30s %d waves/SIMD: fma waves %.1f ticks per fma (64 per 8 MFMAs), mfma waves %.1f ticks per MFMA, kernel %.3f ms\n",
                       names[mode], wps, lo / (64.0 * iters), hi / (8.0 * iters), ms);
            else
                printf("%-30s %d waves/SIMD: %.2f s_memtime ticks per op per wave (waves 0-3), kernel %.3f ms = %.2f ns per op per wave\n",
                       names[mode], wps, lo / nops, ms, ms * 1e6 / nops);
        }
    return 0;
}
