// Permuted (nested dissection) dataflow Cholesky on a synthetic camera network:
// g x g cameras on a grid in acquisition (lawn-mower) order, each coupled to the
// cameras within `reach` grid steps, nio dense IO unknowns.  Compares with the
// in-place dataflow factorisation (natural order) and checks the residual.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 chol_grid_test.hip -o chol_grid_test.bin -lrocblas
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "chol_df.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char **argv) {
    const int g = argc > 1 ? atoi(argv[1]) : 32;
    const double reach = argc > 2 ? atof(argv[2]) : 3.6;
    const int nio = argc > 3 ? atoi(argv[3]) : 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const int nc = g * g, n = 6 * nc + nio;
    const int64_t lda = ((n + 1 + 7) / 8) * 8;
    std::vector<double> xyz(3 * nc);
    for (int j = 0; j < g; ++j)
        for (int i = 0; i < g; ++i) {
            const int c = j * g + i;
            xyz[3 * c] = (j % 2 == 0) ? i : g - 1 - i; xyz[3 * c + 1] = j; xyz[3 * c + 2] = 0;
        }
    const int aw = (nc + 63) / 64;
    std::vector<uint64_t> adj((size_t)nc * aw, 0);
    std::vector<int> first(n, 0);
    for (int a = 0; a < nc; ++a) {
        int f = a;
        for (int b = 0; b < nc; ++b) {
            const double dx = xyz[3 * a] - xyz[3 * b], dy = xyz[3 * a + 1] - xyz[3 * b + 1];
            if (dx * dx + dy * dy <= reach * reach) { adj[(size_t)a * aw + (b >> 6)] |= 1ull << (b & 63); f = std::min(f, b); }
        }
        for (int k = 0; k < 6; ++k) first[6 * a + k] = 6 * f;
    }
    // SPD matrix with that pattern: random symmetric entries, dominant diagonal
    std::vector<double> A((size_t)lda * (n + 1), 0.0), b(n);
    srand(3);
    auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
    for (int c = 0; c < n; ++c)
        for (int r = c; r < n; ++r) {
            bool nz = r >= 6 * nc || c >= 6 * nc;
            if (!nz) { const int ca = r / 6, cb = c / 6; nz = (adj[(size_t)ca * aw + (cb >> 6)] >> (cb & 63)) & 1ull; }
            if (nz) A[(size_t)c * lda + r] = r == c ? 200.0 + rnd() : rnd();
        }
    for (int i = 0; i < n; ++i) { b[i] = rnd(); A[(size_t)i * lda + n] = b[i]; }
    double *dA0, *dA, *dq, *dlinv, *dld; int *dinfo;
    CK(hipMalloc(&dA0, A.size() * 8)); CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dq, n * 8)); CK(hipMalloc(&dld, n * 8));
    CK(hipMalloc(&dinfo, 4));
    CK(hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dbat::CholEnvelope env; env.build(n, 6 * nc, first);
    std::vector<double> q1(n), q2(n);
    for (int mode = 0; mode < 2; ++mode) {
        dbat::DataflowChol df;
        if (mode == 0) df.setup_inplace(env, lda); else df.setup_permuted(nc, nio, adj.data(), aw, xyz.data());
        CK(hipMalloc(&dlinv, df.linv_doubles() * 8));
        float best = 1e9;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, st));
            CK(hipMemsetAsync(dq, 0, n * 8, st));
            CK(hipEventRecord(e0, st));
            df.solve(st, dA, lda, dq, dlinv, dinfo, dld);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        int info; CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy((mode ? q2 : q1).data(), dq, n * 8, hipMemcpyDeviceToHost));
        printf("%s: n=%d tile rows=%d tasks=%d tiles=%d  %.3f ms  info=%d\n", mode ? "permuted (ND)" : "in place     ", n, df.nT,
               df.ntasks, df.ntiles, best, info);
        if (getenv("DF_TRACE") && mode == 1) {
            const size_t nt = (size_t)(df.ntasks + df.nT) * 16;
            CK(hipMalloc(&df.d_trace, nt * 8)); CK(hipMemset(df.d_trace, 0, nt * 8));
            CK(hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, st));
            df.solve(st, dA, lda, dq, dlinv, dinfo, dld); CK(hipStreamSynchronize(st));
            std::vector<long long> tr(nt); CK(hipMemcpy(tr.data(), df.d_trace, nt * 8, hipMemcpyDeviceToHost));
            std::vector<dbat::DfTask> tk(df.ntasks);
            CK(hipMemcpy(tk.data(), df.d_tasks, tk.size() * sizeof(dbat::DfTask), hipMemcpyDeviceToHost));
            long long t0 = tr[0];
            for (int q = 0; q < df.ntasks; ++q) if (tr[q * 16] && tr[q * 16] < t0) t0 = tr[q * 16];
            printf("diag k: task_start deps_ready potf2_done [us] (#updates)\n");
            for (int q = 0; q < df.ntasks; ++q)
                if (tk[q].i == tk[q].k)
                    printf("%d: %.0f %.0f %.0f\n", tk[q].k, (tr[q * 16] - t0) * 0.01, (tr[q * 16 + 1] - t0) * 0.01, (tr[q * 16 + 3] - t0) * 0.01);
            (void)hipFree(df.d_trace); df.d_trace = nullptr;
        }
        df.release();
        CK(hipFree(dlinv));
    }
    double num = 0, den = 0;
    for (int i = 0; i < n; ++i) { num += (q1[i] - q2[i]) * (q1[i] - q2[i]); den += q1[i] * q1[i]; }
    printf("rel |q_perm - q_inplace| = %.3e\n", std::sqrt(num / den));
    double rn = 0, bn = 0;
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int j = 0; j < n; ++j) { const double a = i >= j ? A[(size_t)j * lda + i] : A[(size_t)i * lda + j]; s += a * q2[j]; }
        rn += (s - b[i]) * (s - b[i]); bn += b[i] * b[i];
    }
    printf("permuted residual ||Aq-b||/||b|| = %.3e\n", std::sqrt(rn / bn));
    return 0;
}
