// df_potf2 (dbat_amd/csrc/chol_df.hpp) alone: one workgroup factors the same 64 x 64 block again and again --
// shader ticks per call with a warm instruction cache, and the result against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dbat_amd/csrc bench/potf2_micro.hip -o bench/potf2_micro.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "chol_df.hpp"

using namespace dbat;

__global__ __launch_bounds__(256) void k_potf2(const double *A, int iters, unsigned long long *out, double *res, int *info) {
    __shared__ double smem[64 * DF_TLD];
    __shared__ long long stg[32];
    const int t = threadIdx.x;
    unsigned long long acc = 0;
    if (t < 32) stg[t] = 0;
    for (int it = 0; it < iters; ++it) {
        for (int e = t; e < 4096; e += 256) { const int c = e >> 6, r = e & 63; smem[c * DF_TLD + r] = r >= c ? A[c * 64 + r] : 0.0; }
        __syncthreads();
        const unsigned long long c0 = __builtin_readcyclecounter();
#ifdef DBAT_POTF2_STAGES
        df_potf2(smem, 64, 0, info, stg);
#else
        df_potf2(smem, 64, 0, info, nullptr);
#endif
        __syncthreads();
        acc += __builtin_readcyclecounter() - c0;
    }
    if (t == 0) out[0] = acc;
    if (t < 16) out[1 + t] = (unsigned long long)stg[16 + t];
    for (int e = t; e < 4096; e += 256) { const int c = e >> 6, r = e & 63; res[e] = smem[c * DF_TLD + r]; res[4096 + e] = smem[c * DF_TLD + 64 + r]; }
}

int main() {
    std::vector<double> A(4096);
    for (int c = 0; c < 64; ++c)
        for (int r = 0; r < 64; ++r) A[c * 64 + r] = (r == c ? 70.0 + r : 1.0 / (1 + abs(r - c))) ;
    double *dA, *dres; unsigned long long *out; int *info;
    hipMalloc(&dA, 4096 * 8); hipMalloc(&dres, 8192 * 8); hipMalloc(&out, 17 * 8); hipMalloc(&info, 4);
    hipMemcpy(dA, A.data(), 4096 * 8, hipMemcpyHostToDevice); hipMemset(info, 0, 4);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_potf2, dim3(1), dim3(256), 0, 0, dA, iters, out, dres, info); hipDeviceSynchronize(); }
    unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    std::vector<double> R(8192); hipMemcpy(R.data(), dres, 8192 * 8, hipMemcpyDeviceToHost);
    // host check: L L' = A (lower), L^-T rows: L * Linv' = I
    double err = 0, erri = 0;
    for (int r = 0; r < 64; ++r)
        for (int c = 0; c <= r; ++c) {
            double s = 0;
            for (int m = 0; m <= c; ++m) s += R[m * 64 + r] * R[m * 64 + c];
            err = fmax(err, fabs(s - A[c * 64 + r]));
        }
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
            double s = 0;                                  // (L Linv)(i, j), Linv(m, j) = smem[m][64 + j] = R[4096 + m*64 + j]
            for (int m = j; m <= i; ++m) s += R[m * 64 + i] * R[4096 + m * 64 + j];   // (L^-1 is lower triangular: nothing is stored above)
            erri = fmax(erri, fabs(s - (i == j ? 1.0 : 0.0)));
        }
    int hinfo; hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
#ifdef DBAT_POTF2_STAGES
    {
        unsigned long long st[17]; hipMemcpy(st, out, 17 * 8, hipMemcpyDeviceToHost);
        const char *nm[7] = {"top (entry / after the barrier)", "rows loaded", "eliminated", "rows stored", "barrier", "trailing update", "barrier"};
        for (int i = 0; i < 7; ++i) printf("  station %d %-32s %7.0f ticks per call (all panels)\n", i, nm[i], (double)st[1 + i] / iters);
    }
#endif
    printf("df_potf2: %.0f shader ticks per call (%.2f us at 2.4 GHz), |L L' - A| = %.3g, |L Linv - I| = %.3g, info %d\n",
           (double)h / iters, (double)h / iters / 2400.0, err, erri, hinfo);
    return 0;
}
