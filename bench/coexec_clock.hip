// Do v_mfma_f64_16x16x4_f64 and v_fma_f64 overlap on gfx950 -- inside ONE wave (hand-interleaved stream) and between
// the two waves of a SIMD?  (round 4: the question behind "one wave per SIMD, MFMAs of round r interleaved with the
// evaluation of round r+1" for k_build_sig.)  All streams are inline asm, so the order is the one written here.
//   hipcc --offload-arch=gfx950 -O3 bench/coexec_clock.hip -o bench/coexec_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a))

// one iteration = 4 MFMAs (independent accumulators), each followed by NF fmas spread over NCH independent chains
template <int NF, int NCH>
__device__ __forceinline__ double mix(int iters, double a, double b) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double x[NCH];
    for (int c = 0; c < NCH; ++c) x[c] = a + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            MFMA(acc[i]);
#pragma unroll
            for (int f = 0; f < NF; ++f) FMA(x[f % NCH]);
        }
    }
    double s = 0;
    for (int c = 0; c < NCH; ++c) s += x[c];
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    return s;
}
template <int NCH>
__device__ __forceinline__ double fmas(int iters, double a, double b) {      // 32 fmas per iteration
    double x[NCH];
    for (int c = 0; c < NCH; ++c) x[c] = a + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < 32; ++f) FMA(x[f % NCH]);
    }
    double s = 0;
    for (int c = 0; c < NCH; ++c) s += x[c];
    return s;
}
__device__ __forceinline__ double mfmas(int iters, double a, double b) {     // 4 MFMAs per iteration
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) MFMA(acc[i]);
    }
    return acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
}

// mode 0..5: every wave runs the mixed stream with NF = 0, 4, 8, 12, 16, 24 fmas per MFMA (4 chains)
// mode 6: NF = 12 on 2 chains; mode 7: NF = 12 on 1 chain
// mode 8/9/10: waves 4..7 multiply back to back, waves 0..3 run fmas on 1 / 2 / 4 chains (two waves per SIMD)
// mode 11: fmas on 4 chains alone (no MFMA neighbour)
__global__ void k(int mode, int iters, unsigned long long *out, double *sink) {
    const double a = 1e-9 * threadIdx.x, b = 1.0 - 1e-12;
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    double s = 0;
    switch (mode) {
    case 0: s = mix<0, 4>(iters, a, b); break;
    case 1: s = mix<4, 4>(iters, a, b); break;
    case 2: s = mix<8, 4>(iters, a, b); break;
    case 3: s = mix<12, 4>(iters, a, b); break;
    case 4: s = mix<16, 4>(iters, a, b); break;
    case 5: s = mix<24, 4>(iters, a, b); break;
    case 6: s = mix<12, 2>(iters, a, b); break;
    case 7: s = mix<12, 1>(iters, a, b); break;
    case 8: s = wave >= 4 ? mfmas(iters, a, b) : fmas<1>(iters, a, b); break;
    case 9: s = wave >= 4 ? mfmas(iters, a, b) : fmas<2>(iters, a, b); break;
    case 10: s = wave >= 4 ? mfmas(iters, a, b) : fmas<4>(iters, a, b); break;
    case 11: s = wave >= 4 ? 0.0 : fmas<4>(iters, a, b); break;
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = c1 - c0;
}

int main() {
    const int iters = 4000, blocks = 256;
    unsigned long long *out; double *sink;
    hipMalloc(&out, blocks * 16 * 8); hipMalloc(&sink, (size_t)blocks * 512 * 8);
    const int nf[8] = {0, 4, 8, 12, 16, 24, 12, 12}, nch[8] = {4, 4, 4, 4, 4, 4, 2, 1};
    for (int mode = 0; mode < 12; ++mode)
        for (int wps : {1, 2}) {
            if (mode >= 8 && wps == 1) continue;
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
            for (int bb = 0; bb < blocks; ++bb) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += h[bb * 16 + w] / (4.0 * blocks);
            if (mode < 8)
                printf("mix: 1 MFMA + %2d fma (%d chains)   %d waves/SIMD: %.1f ticks per MFMA group per wave (waves 0-3), kernel %.3f ms\n",
                       nf[mode], nch[mode], wps, lo / (4.0 * iters), ms);
            else if (mode < 11)
                printf("mfma wave beside fma wave (%d chains): fma waves %.2f ticks per fma, mfma waves %.1f ticks per MFMA, kernel %.3f ms\n",
                       1 << (mode - 8), lo / (32.0 * iters), hi / (4.0 * iters), ms);
            else
                printf("fma wave (4 chains) beside an idle wave: %.2f ticks per fma, kernel %.3f ms\n", lo / (32.0 * iters), ms);
        }
    return 0;
}
