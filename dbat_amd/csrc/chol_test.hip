// Standalone check + timing of chol.hpp against rocsolver_dpotrf/dpotrs.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 chol_test.hip -o chol_test -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "chol_df.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 6000;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const int band = argc > 3 ? atoi(argv[3]) : 0;      // >0: banded test matrix with this half-bandwidth and a dense tail
    const int tail = argc > 4 ? atoi(argv[4]) : 0;
    const int64_t lda = n + 8 - (n % 8 == 0 ? 0 : n % 8) + 8;   // >= n+1, multiple of 8
    rocblas_handle h; rocblas_create_handle(&h);
    hipStream_t st; CK(hipStreamCreate(&st)); rocblas_set_stream(h, st);
    std::vector<double> M((size_t)n * n), b(n);
    srand(1);
    for (auto &v : M) v = (rand() / (double)RAND_MAX) - 0.5;
    for (auto &v : b) v = (rand() / (double)RAND_MAX) - 0.5;
    double *dM, *dA0, *dA, *dB, *dq, *dy, *dlinv; int *dinfo;
    CK(hipMalloc(&dlinv, dbat::BlockChol::linv_doubles(n) * 8));
    CK(hipMalloc(&dM, (size_t)n * n * 8)); CK(hipMalloc(&dA0, (size_t)lda * (n + 1) * 8));
    CK(hipMalloc(&dA, (size_t)lda * (n + 1) * 8)); CK(hipMalloc(&dB, n * 8)); CK(hipMalloc(&dq, n * 8));
    CK(hipMalloc(&dy, n * 8)); CK(hipMalloc(&dinfo, 4));
    CK(hipMemcpy(dM, M.data(), (size_t)n * n * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dA0, 0, (size_t)lda * (n + 1) * 8));
    const double one = 1.0, zero = 0.0;
    rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &one, dM, n, dM, n, &zero, dA0, (int)lda);
    std::vector<double> A((size_t)lda * (n + 1));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(A.data(), dA0, A.size() * 8, hipMemcpyDeviceToHost));
    std::vector<int> rowfirst(n, 0);
    const int tail0 = n - tail;
    if (band > 0) {
        // keep only |i-j| <= band (rounded to 6x6 blocks) outside the dense tail rows
        for (int i = 0; i < n; ++i) rowfirst[i] = i < tail0 ? std::max(0, (i / 6) * 6 - band) : 0;
        for (int j = 0; j < n; ++j)
            for (int i = j; i < n; ++i)
                if (i < tail0 && j < rowfirst[i]) A[(size_t)j * lda + i] = 0.0;
    }
    dbat::CholEnvelope env;
    if (band > 0) env.build(n, tail0, rowfirst); else env.build_dense(n);
    for (int i = 0; i < n; ++i) A[(size_t)i * lda + i] += n;        // well conditioned SPD
    for (int i = 0; i < n; ++i) A[(size_t)i * lda + n] = b[i];      // rhs row
    CK(hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---- own
    float best = 1e9;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, st));
        CK(hipEventRecord(e0, st));
        dbat::BlockChol::solve(h, st, dA, lda, n, dq, dy, dlinv, dinfo, env);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<double> q(n); int info;
    CK(hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    printf("own   blocked chol+solve n=%d: %.3f ms  (%.2f TFLOP/s)  info=%d\n", n, best, (double)n * n * n / 3 / best / 1e9, info);
    // ---- dataflow, in place
    {
        dbat::DataflowChol df; df.setup_inplace(env, lda);
        float bdf = 1e9;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, st));
            CK(hipMemsetAsync(dq, 0, n * 8, st));
            CK(hipEventRecord(e0, st));
            df.solve(st, dA, lda, dq, dlinv, dinfo);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < bdf) bdf = ms;
        }
        std::vector<double> qd(n); int infod;
        CK(hipMemcpy(qd.data(), dq, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&infod, dinfo, 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        for (int i = 0; i < n; ++i) { num += (qd[i] - q[i]) * (qd[i] - q[i]); den += q[i] * q[i]; }
        printf("dataflow (in place) chol+solve n=%d: %.3f ms  tasks=%d grid=%d info=%d  rel |q_df - q_own| = %.3e\n", n, bdf,
               df.ntasks, df.grid, infod, std::sqrt(num / den));
        df.release();
    }
    // ---- rocsolver
    float best2 = 1e9;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, st));
        CK(hipMemcpyAsync(dB, b.data(), n * 8, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(e0, st));
        rocsolver_dpotrf(h, rocblas_fill_lower, n, dA, (int)lda, dinfo);
        rocsolver_dpotrs(h, rocblas_fill_lower, n, 1, dA, (int)lda, dB, n);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best2) best2 = ms;
    }
    std::vector<double> q2(n);
    CK(hipMemcpy(q2.data(), dB, n * 8, hipMemcpyDeviceToHost));
    printf("rocsolver potrf+potrs n=%d: %.3f ms  (%.2f TFLOP/s)\n", n, best2, (double)n * n * n / 3 / best2 / 1e9);
    double num = 0, den = 0;
    for (int i = 0; i < n; ++i) { num += (q[i] - q2[i]) * (q[i] - q2[i]); den += q2[i] * q2[i]; }
    printf("rel |q_own - q_rocsolver| = %.3e\n", std::sqrt(num / den));
    // residual check of own solution: ||A q - b|| / ||b||
    double rn = 0, bn = 0;
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int j = 0; j < n; ++j) { const double a = i >= j ? A[(size_t)j * lda + i] : A[(size_t)i * lda + j]; s += a * q[j]; }
        rn += (s - b[i]) * (s - b[i]); bn += b[i] * b[i];
    }
    printf("own residual ||Aq-b||/||b|| = %.3e\n", std::sqrt(rn / bn));
    return 0;
}
