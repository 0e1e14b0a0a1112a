// micro-benchmarks: f64 MFMA issue rate, LDS f64 atomics, barrier cost (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(unsigned long long *out, double *sink, int iters, int nacc) {
    d4 acc[9];
    for (int s = 0; s < 9; ++s) acc[s] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (nacc == 9) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 9; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[s], 0, 0, 0);
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 9; ++s) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < 9; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    sink[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_ldsatomic(unsigned long long *out, double *sink, int iters, int mode) {
    __shared__ double buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int lane = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        int idx = mode == 0 ? (lane + 64 * (i & 63)) : mode == 1 ? ((lane & 7) + 8 * (i & 63)) : (lane * 37 + i * 13) & 8191;
        if (mode == 3) buf[(lane + 64 * (i & 63))] = i;            // plain store
        else unsafeAtomicAdd(&buf[idx], 1.0);
    }
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[threadIdx.x] = buf[threadIdx.x];
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_barrier(unsigned long long *out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    unsigned long long *d, h; double *sink;
    CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 8 * 1024 * 1024));
    for (int nacc : {9, 1}) for (int nw : {1, 4, 8}) {
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64 * nw), 0, 0, d, sink, 2000, nacc);
        CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        printf("mfma f64 16x16x4: %d independent acc, %d waves/CU: %.1f cycles per MFMA per wave\n", nacc, nw, h / (2000.0 * 9));
    }
    const char *names[] = {"ds_add_f64 distinct addr, conflict-free", "ds_add_f64 8 lanes per address", "ds_add_f64 scattered", "ds_write_b64 plain"};
    for (int mode = 0; mode < 4; ++mode) for (int nw : {1, 4}) {
        hipLaunchKernelGGL(k_ldsatomic, dim3(1), dim3(64 * nw), 0, 0, d, sink, 4096, mode);
        CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        printf("%s, %d waves: %.1f cycles per wave-instruction (all waves total per instr: %.1f)\n", names[mode], nw, h / 4096.0, h / 4096.0 / nw);
    }
    for (int nw : {4, 5, 8}) {
        hipLaunchKernelGGL(k_barrier, dim3(1), dim3(64 * nw), 0, 0, d, 10000);
        CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        printf("__syncthreads, %d waves: %.1f cycles\n", nw, h / 10000.0);
    }
    return 0;
}
