// How does rocprofv3's FETCH_SIZE count narrow gathers on gfx950?  The guide calibrates it on wide
// streaming reads only (it reports 1/2 of their bytes).  The camera-major kernels gather one
// 24-byte object point per observation; this program runs
//   k_stream : reads N x 16 B consecutively (16 B per lane)            -> known bytes: 16 N
//   k_gather : reads N x 24 B at sorted random indices into a table    -> distinct bytes: 24 * (#distinct points)
// with a table of 24 MB (the C3 object points: fits the 256 MB Infinity Cache, not the 4 MB L2) and of
// 1.2 GB (nothing fits), ten gathers per table entry as in C3.  Run under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o calib -- ./gather_calib
// and compare the per-kernel counter with the byte counts printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

__global__ void k_stream(const double2 *__restrict__ a, double *__restrict__ out, size_t n) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i].x + a[i].y;
    if (s == 1.2345) out[0] = s;
}
__global__ void k_gather(const double *__restrict__ tab, const int *__restrict__ idx, double *__restrict__ out, size_t n) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double *q = tab + 3 * (size_t)idx[i];
        s += q[0] + q[1] + q[2];
    }
    if (s == 1.2345) out[0] = s;
}

int main() {
    const size_t n = 10000000;                       // observations
    for (size_t npts : {(size_t)1000000, (size_t)50000000}) {
        std::vector<int> idx(n);
        std::mt19937_64 rng(1);
        // camera-major order: chunks of 2048 observations whose points ascend (a camera sees a compact patch)
        for (size_t c = 0; c < n; c += 2048) {
            const size_t m = std::min<size_t>(2048, n - c);
            const size_t span = std::min<size_t>(npts, 20000), base = rng() % (npts - span + 1);
            for (size_t i = 0; i < m; ++i) idx[c + i] = (int)(base + rng() % span);
            std::sort(idx.begin() + c, idx.begin() + c + m);
        }
        std::vector<int> u(idx);
        std::sort(u.begin(), u.end());
        const size_t distinct = std::unique(u.begin(), u.end()) - u.begin();
        double *tab, *out; int *didx; double2 *str;
        hipMalloc(&tab, npts * 24); hipMalloc(&out, 8); hipMalloc(&didx, n * 4); hipMalloc(&str, n * 16);
        hipMemset(tab, 0, npts * 24); hipMemset(str, 0, n * 16);
        hipMemcpy(didx, idx.data(), n * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, str, out, n);
            hipLaunchKernelGGL(k_gather, dim3(2048), dim3(256), 0, 0, tab, didx, out, n);
        }
        hipDeviceSynchronize();
        printf("table %zu points (%.0f MB): k_stream reads %.1f MB; k_gather reads %.1f MB of indices + %.1f MB gathered "
               "(%.1f MB distinct table bytes, %.1f MB if every 64-byte line it touches is fetched once)\n",
               npts, npts * 24e-6, n * 16e-6, n * 4e-6, n * 24e-6, distinct * 24e-6, 0.0);
        hipFree(tab); hipFree(out); hipFree(didx); hipFree(str);
    }
    return 0;
}
