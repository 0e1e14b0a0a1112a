// What does one column of the 16-column panel elimination of df_potf2 (chol_df.hpp) cost, and
// which way of broadcasting the multipliers is cheapest?  One wave per SIMD, as in k_chol_df.
//   V0  multipliers by v_readlane (SGPR operands), chain on wave-uniform values
//   V1  multipliers through LDS (one ds_write of the column, broadcast ds_read_b128): slower (3500 ticks), not run
//   V2  multipliers as DPP row_newbcast operands of v_fmac_f64 (what df_potf2 does now)
//   P*  probes: dependent v_fma_f64 chain, independent v_fma_f64, readlane+fma pairs, v_rsq_f64 chain
// hipcc --offload-arch=gfx950 -O3 bench/potf_micro.hip -o /tmp/potf_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// a += (m of lane K of the same 16-lane row) * b: DPP row_newbcast on the f64 operation itself
template <int K, bool FIRST = false>
__device__ __forceinline__ void fmac_bcast(double &a, double m, double b) {
    if constexpr (FIRST)     // m, b were just written by VALU instructions the assembler cannot see from here
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m), "v"(b), "n"(K));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m), "v"(b), "n"(K));
}
template <int K, bool FIRST = true>
struct UpdFrom {
    static __device__ __forceinline__ void run(double (&a)[16], double m, double nly) {
        if constexpr (K < 16) { fmac_bcast<K, FIRST>(a[K], m, nly); UpdFrom<K + 1, false>::run(a, m, nly); }
    }
};
__device__ __forceinline__ double swap16_f64(double v) {       // value of lane ^ 16
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_swizzle((int)(b & 0xffffffffll), 0x401F);
    const int hi = __builtin_amdgcn_ds_swizzle((int)(b >> 32), 0x401F);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int J>
struct ElimCol {
    static __device__ __forceinline__ void run(double (&a)[16], double &piv, int lane) {
        const double araw = a[J];
        const double sw = swap16_f64(araw);
        const double m = (lane & 16) ? sw : araw;    // every 16-lane row holds the diagonal rows' column
        const double c1 = J < 15 ? readlane_f64(araw, J + 1 < 16 ? J + 1 : 0) : 0.0;
        const double rn = J < 15 ? readlane_f64(a[J + 1 < 16 ? J + 1 : 0], J + 1 < 16 ? J + 1 : 0) : 0.0;
        const double y0 = __builtin_amdgcn_rsq(piv);
        const double e = __builtin_fma(-piv, y0 * y0, 1.0);
        const double y = __builtin_fma(y0, __builtin_fma(e, 0.375, 0.5) * e, y0);
        const double y2 = y * y;
        piv = __builtin_fma(-c1, c1 * y2, rn);
        const double ly = araw * y2;
        a[J] = araw * y;
        if constexpr (J < 15) a[J + 1] = __builtin_fma(-c1, ly, a[J + 1]);
        UpdFrom<J + 2>::run(a, m, -ly);
        if constexpr (J < 15) ElimCol<J + 1>::run(a, piv, lane);
    }
};

// Variants of the column of V2 (round 4):
//   X = 1  1/d = t (1 + e + e^2), t = y0^2 (two operations off the dependent chain, one more in total)
//   X = 2  no scalar copy of the next pivot: it is the updated a[J+1] of lane J+1, fetched after the update
//          (two v_readlane and two f64 operations fewer per column, a longer chain)
//   X = 3  both
//   X & 4  y = y0 (1 + e/2): second order (v_rsq_f64 would have to be good to 2^-27);  X & 8: 1/d = t (1 + e) likewise
template <int J, int X>
struct ElimColX {
    static __device__ __forceinline__ void run(double (&a)[16], double &piv, int lane) {
        const double araw = a[J];
        const double sw = swap16_f64(araw);
        const double m = (lane & 16) ? sw : araw;
        constexpr int J1 = J + 1 < 16 ? J + 1 : 0;
        const double y0 = __builtin_amdgcn_rsq(piv);
        const double t = y0 * y0;
        const double e = __builtin_fma(-piv, t, 1.0);
        double y;
        if constexpr (X & 4) y = __builtin_fma(0.5 * y0, e, y0); else y = __builtin_fma(y0, __builtin_fma(e, 0.375, 0.5) * e, y0);
        double y2;
        if constexpr (X & 8) y2 = __builtin_fma(t, e, t);
        else if constexpr (X & 1) y2 = __builtin_fma(t, __builtin_fma(e, e, e), t); else y2 = y * y;
        const double ly = araw * y2;
        if constexpr (X & 2) {
            if constexpr (J < 15) {
                fmac_bcast<J1, true>(a[J1], m, -ly);
                piv = readlane_f64(a[J1], J1);
            }
            a[J] = araw * y;
            UpdFrom<J + 2, false>::run(a, m, -ly);
        } else {
            const double c1 = J < 15 ? readlane_f64(araw, J1) : 0.0;
            const double rn = J < 15 ? readlane_f64(a[J1], J1) : 0.0;
            piv = __builtin_fma(-c1 * c1, y2, rn);
            a[J] = araw * y;
            if constexpr (J < 15) a[J + 1] = __builtin_fma(-c1, ly, a[J + 1]);
            UpdFrom<J + 2>::run(a, m, -ly);
        }
        if constexpr (J < 15) ElimColX<J + 1, X>::run(a, piv, lane);
    }
};

template <int V>
__device__ __forceinline__ void eliminate(double (&a)[16], double *col /* LDS, [2][16] of this wave */, int lane) {
    if constexpr (V == 0) {
        double piv = readlane_f64(a[0], 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double araw = a[j];
            const double c1 = j < 15 ? readlane_f64(araw, j + 1) : 0.0;
            const double rn = j < 15 ? readlane_f64(a[j + 1], j + 1) : 0.0;
            const double y0 = __builtin_amdgcn_rsq(piv);
            const double e = __builtin_fma(-piv, y0 * y0, 1.0);
            const double y = __builtin_fma(y0, __builtin_fma(e, 0.375, 0.5) * e, y0);
            const double y2 = y * y;
            piv = __builtin_fma(-c1, c1 * y2, rn);
            const double ly = araw * y2;
            a[j] = araw * y;
            if (j < 15) a[j + 1] = __builtin_fma(-c1, ly, a[j + 1]);
#pragma unroll
            for (int k = j + 2; k < 16; ++k) a[k] = __builtin_fma(-readlane_f64(araw, k), ly, a[k]);
        }
    } else if constexpr (V == 2) {
        double piv = readlane_f64(a[0], 0);
        ElimCol<0>::run(a, piv, lane);
    } else if constexpr (V >= 3) {
        double piv = readlane_f64(a[0], 0);
        ElimColX<0, V - 2>::run(a, piv, lane);
    } else {
        // column j of the diagonal rows -> LDS; every lane reads the multipliers back (same address: broadcast)
        double piv = readlane_f64(a[0], 0);
        if (lane < 16) col[lane] = a[0];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double araw = a[j];
            double c[16];
            const double *cj = col + 16 * (j & 1);
#pragma unroll
            for (int k = (j + 1) & ~1; k < 16; k += 2) {     // 16-byte broadcast reads
                const double2 v = *reinterpret_cast<const double2 *>(cj + k);
                c[k] = v.x; c[k + 1] = v.y;
            }
            const double rn = j < 15 ? readlane_f64(a[j + 1], j + 1) : 0.0;
            const double y0 = __builtin_amdgcn_rsq(piv);
            const double e = __builtin_fma(-piv, y0 * y0, 1.0);
            const double y = __builtin_fma(y0, __builtin_fma(e, 0.375, 0.5) * e, y0);
            const double y2 = y * y;
            const double ly = araw * y2;
            a[j] = araw * y;
            if (j < 15) {
                const double c1 = c[j + 1];
                piv = __builtin_fma(-c1, c1 * y2, rn);
                a[j + 1] = __builtin_fma(-c1, ly, a[j + 1]);
                if (lane < 16) col[16 * ((j + 1) & 1) + lane] = a[j + 1];     // the next column is final
            }
#pragma unroll
            for (int k = j + 2; k < 16; ++k) a[k] = __builtin_fma(-c[k], ly, a[k]);
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void k_elim(const double *A /* [16][32] col-major: 32 rows */, int iters,
                                              unsigned long long *out, double *res) {
    __shared__ double col[4][32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double a0[16];
    for (int q = 0; q < 16; ++q) a0[q] = lane < 32 ? A[q * 32 + lane] : 0.0;
    double a[16];
    double sum = 0;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = a0[q] + (double)it * 0.0;
        eliminate<V>(a, col[w], lane);
        sum += a[15];
        __builtin_amdgcn_wave_barrier();
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (lane == 0) { out[2 * w] = c1 - c0; out[2 * w + 1] = w1 - w0; }
    if (w == 0 && lane < 32)
        for (int q = 0; q < 16; ++q) res[q * 32 + lane] = a[q];
    if (sum == 1234.5) res[0] = sum;
}

// probes: cycles per instruction of simple streams, one wave per SIMD
template <int P>
__global__ __launch_bounds__(256) void k_probe(int iters, unsigned long long *out, double *sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double x = 1.0 + lane * 1e-6, y = 0.999999, z[8];
    for (int i = 0; i < 8; ++i) z[i] = 1.0 + i + lane;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (P == 0) {                      // 16 dependent fma
#pragma unroll
            for (int i = 0; i < 16; ++i) x = __builtin_fma(x, y, 1e-9);
        } else if constexpr (P == 1) {               // 16 independent fma (8 chains x 2)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = __builtin_fma(z[i], y, 1e-9);
        } else if constexpr (P == 2) {               // 8 x (readlane pair + fma), independent targets
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = __builtin_fma(-readlane_f64(x, i + 1), y, z[i]);
            x += 1e-9;
        } else if constexpr (P == 3) {               // 4 dependent rsq + fma
#pragma unroll
            for (int i = 0; i < 4; ++i) x = __builtin_fma(__builtin_amdgcn_rsq(x), y, 1.0);
        } else if constexpr (P == 4) {               // 16 readlane_b32 only (feeding one add each 8)
            int s = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += __builtin_amdgcn_readlane(__double2loint(x) + i, i);
            x += (double)s * 1e-300;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = x;
    for (int i = 0; i < 8; ++i) s += z[i];
    sink[threadIdx.x] = s;
    if (lane == 0) { out[2 * w] = c1 - c0; out[2 * w + 1] = w1 - w0; }
}

typedef double d4 __attribute__((ext_vector_type(4)));
// the 64 x 64 x 64 tile product of chol.hpp (mfma_tile64) from LDS operands, 4 waves
template <int LD, int VAR>
__global__ __launch_bounds__(256) void k_tile64(int iters, unsigned long long *out, double *sink) {
    __shared__ double Pm[64 * LD], Qm[64 * LD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * LD; i += 256) { Pm[i] = 1e-3 * (i % 17); Qm[i] = 1e-3 * (i % 13); }
    __syncthreads();
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (VAR == 0) {
#pragma unroll 4
            for (int kk = 0; kk < 16; ++kk) {
                const int mrow = 4 * kk + (lane >> 4);
                const double a = Pm[mrow * LD + 16 * w + (lane & 15)];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const double b = Qm[mrow * LD + 16 * rt + (lane & 15)];
                    acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[rt], 0, 0, 0);
                }
            }
        } else if (VAR == 2) {                       // row blocks interleaved (r = 4 j + rt): the four b operands are 32 contiguous bytes
            const double *pa = Pm + (lane >> 4) * LD + 16 * w + (lane & 15), *pb = Qm + (lane >> 4) * LD + 4 * (lane & 15);
            double an = pa[0];
            double2 b01 = *reinterpret_cast<const double2 *>(pb), b23 = *reinterpret_cast<const double2 *>(pb + 2);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const double a = an;
                const double2 c01 = b01, c23 = b23;
                if (kk + 1 < 16) {
                    an = pa[4 * (kk + 1) * LD];
                    b01 = *reinterpret_cast<const double2 *>(pb + 4 * (kk + 1) * LD);
                    b23 = *reinterpret_cast<const double2 *>(pb + 4 * (kk + 1) * LD + 2);
                }
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c01.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c01.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c23.x, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c23.y, acc[3], 0, 0, 0);
            }
        } else {                                     // operands of k-step kk+1 on their way while kk multiplies
            const double *pa = Pm + (lane >> 4) * LD + 16 * w + (lane & 15), *pb = Qm + (lane >> 4) * LD + (lane & 15);
            double an = pa[0], bn[4] = {pb[0], pb[16], pb[32], pb[48]};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const double a = an, b0 = bn[0], b1 = bn[1], b2 = bn[2], b3 = bn[3];
                if (kk + 1 < 16) {
                    an = pa[4 * (kk + 1) * LD];
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) bn[rt] = pb[4 * (kk + 1) * LD + 16 * rt];
                }
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[3], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[threadIdx.x] = s;
    if (lane == 0) { out[2 * w] = c1 - c0; out[2 * w + 1] = w1 - w0; }
}

// accuracy of v_rsq_f64 / v_rcp_f64: max |1 - x y0^2|, max |1 - x r0| over a sweep of x
__global__ void k_rsq_acc(double *out) {
    double me = 0, mr = 0;
    for (int i = threadIdx.x; i < (1 << 22); i += blockDim.x) {
        const double x = exp2((double)(i % 97) - 48.0) * (1.0 + (double)i * (1.0 / (1 << 22)));
        const double y0 = __builtin_amdgcn_rsq(x);
        me = fmax(me, fabs(__builtin_fma(-x, y0 * y0, 1.0)));
        const double r0 = __builtin_amdgcn_rcp(x);
        mr = fmax(mr, fabs(__builtin_fma(-x, r0, 1.0)));
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = me; sh[1][threadIdx.x] = mr;
    __syncthreads();
    if (threadIdx.x == 0) { for (int k = 1; k < 256; ++k) { me = fmax(me, sh[0][k]); mr = fmax(mr, sh[1][k]); } out[0] = me; out[1] = mr; }
}

int main() {
    // SPD 16 x 16 block + 16 more rows
    std::vector<double> A(16 * 32);
    for (int c = 0; c < 16; ++c)
        for (int r = 0; r < 32; ++r) {
            double v = r < 16 ? (r == c ? 20.0 + r : 1.0 / (1 + abs(r - c))) : sin(0.37 * r + c);
            if (r < 16 && r < c) v = 0.0;
            A[c * 32 + r] = v;
        }
    double *dA, *dres[3], *sink; unsigned long long *out;
    hipMalloc(&dA, A.size() * 8); hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&dres[0], A.size() * 8); hipMalloc(&dres[1], A.size() * 8); hipMalloc(&dres[2], A.size() * 8); hipMalloc(&sink, 256 * 8); hipMalloc(&out, 64);
    const int iters = 2000;
    std::vector<double> r0(A.size()), r1(A.size()), r2(A.size());
    for (int v = 0; v < 8; ++v) {
        if (v == 1) continue;                 // (V1, the LDS broadcast, is slower and is not maintained)
        for (int rep = 0; rep < 2; ++rep) {
            if (v == 0) hipLaunchKernelGGL(k_elim<0>, dim3(1), dim3(256), 0, 0, dA, iters, out, dres[0]);
            else if (v == 2) hipLaunchKernelGGL(k_elim<2>, dim3(1), dim3(256), 0, 0, dA, iters, out, dres[2]);
            else if (v == 3) hipLaunchKernelGGL(k_elim<3>, dim3(1), dim3(256), 0, 0, dA, iters, out, dres[1]);
            else if (v == 4) hipLaunchKernelGGL(k_elim<4>, dim3(1), dim3(256), 0, 0, dA, iters, out, dres[1]);
            else if (v == 5) hipLaunchKernelGGL(k_elim<5>, dim3(1), dim3(256), 0, 0, dA, iters, out, dres[1]);
            else if (v == 6) hipLaunchKernelGGL(k_elim<9>, dim3(1), dim3(256), 0, 0, dA, iters, out, dres[1]);     // X = 7
            else hipLaunchKernelGGL(k_elim<16>, dim3(1), dim3(256), 0, 0, dA, iters, out, dres[1]);               // X = 14
            hipDeviceSynchronize();
        }
        if (v >= 3) {
            hipMemcpy(r1.data(), dres[1], A.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(r0.data(), dres[0], A.size() * 8, hipMemcpyDeviceToHost);
            double mdx = 0; for (size_t i = 0; i < A.size(); ++i) mdx = fmax(mdx, fabs(r0[i] - r1[i]));
            printf("   max |V0 - V%d| = %.3g\n", v, mdx);
        }
        unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("V%d: %.0f shader ticks, %.3f us per 16-column panel (wave 0), %.1f ticks per column\n", v,
               (double)h[0] / iters, (double)h[1] * 0.01 / iters, (double)h[0] / iters / 16);
    }
    hipMemcpy(r0.data(), dres[0], A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), dres[1], A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), dres[2], A.size() * 8, hipMemcpyDeviceToHost);
    { double md2 = 0; for (size_t i = 0; i < A.size(); ++i) md2 = fmax(md2, fabs(r0[i] - r2[i])); printf("max |V0 - V2 (DPP row_newbcast)| = %.3g\n", md2); }
    double md = 0, chk = 0;
    for (size_t i = 0; i < A.size(); ++i) { md = fmax(md, fabs(r0[i] - r1[i])); chk += r0[i]; }
    // host check of L: L L' == A on the diagonal block
    double err = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c <= r; ++c) {
            double s = 0;
            for (int m = 0; m <= c; ++m) s += r0[m * 32 + r] * r0[m * 32 + c];
            err = fmax(err, fabs(s - A[c * 32 + r]));
        }
    (void)md;
    printf("checksum %.12g, |L L' - A| = %.3g\n", chk, err);
    for (int var = 0; var < 3; ++var) {
        for (int rep = 0; rep < 2; ++rep) {
            if (var == 0) hipLaunchKernelGGL((k_tile64<65, 0>), dim3(1), dim3(256), 0, 0, 2000, out, sink);
            else if (var == 1) hipLaunchKernelGGL((k_tile64<65, 1>), dim3(1), dim3(256), 0, 0, 2000, out, sink);
            else hipLaunchKernelGGL((k_tile64<66, 2>), dim3(1), dim3(256), 0, 0, 2000, out, sink);
            hipDeviceSynchronize();
        }
        unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("64^3 tile product from LDS (64 MFMAs per wave), variant %d: %.0f shader ticks, %.3f us\n", var, (double)h[0] / 2000, (double)h[1] * 0.01 / 2000);
    }
    {
        double *acc; hipMalloc(&acc, 16); hipLaunchKernelGGL(k_rsq_acc, dim3(1), dim3(256), 0, 0, acc); hipDeviceSynchronize();
        double h[2]; hipMemcpy(h, acc, 16, hipMemcpyDeviceToHost);
        printf("v_rsq_f64: max |1 - x y0^2| = %.3g (2^%.1f);  v_rcp_f64: max |1 - x r0| = %.3g (2^%.1f)\n", h[0], log2(h[0]), h[1], log2(h[1]));
    }
    const char *names[] = {"16 dependent v_fma_f64", "16 independent v_fma_f64", "8 x (2 v_readlane + v_fma_f64)",
                           "4 dependent (v_rsq_f64 + v_fma_f64)", "16 v_readlane_b32 + adds"};
    for (int p = 0; p < 5; ++p) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (p) {
                case 0: hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(256), 0, 0, 20000, out, sink); break;
                case 1: hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(256), 0, 0, 20000, out, sink); break;
                case 2: hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(256), 0, 0, 20000, out, sink); break;
                case 3: hipLaunchKernelGGL(k_probe<3>, dim3(1), dim3(256), 0, 0, 20000, out, sink); break;
                case 4: hipLaunchKernelGGL(k_probe<4>, dim3(1), dim3(256), 0, 0, 20000, out, sink); break;
            }
            hipDeviceSynchronize();
        }
        unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("probe %d (%s): %.1f shader ticks per iteration, %.4f us\n", p, names[p], (double)h[0] / 20000,
               (double)h[1] * 0.01 / 20000);
    }
    return 0;
}
