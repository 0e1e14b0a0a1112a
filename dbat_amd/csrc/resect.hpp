// Spatial resection of camera stations from control points on the device.
//
// Restates photogrammetry/resect.m:42-131 (per camera: candidate triangles of control points, the best
// pose by the reprojection error of the check points) and photogrammetry/pm_resect_3pt.m:27-147 (Haralick
// et al. 1994, Grunert's solution: a quartic in the ratio of two ray lengths, up to four poses per
// triangle).  One wave per camera: lane 0 solves the quartics of the camera's candidate triangles and
// builds the candidate poses, all lanes score every pose against the camera's check points
// (embarrassingly parallel over cameras: 21 ... 5 000 of them).  The host side (dbat_amd/initial.py)
// keeps what the reference does with MATLAB built-ins around it: lens correction, the choice of the
// triangles (convex hull), and camera centre / Euler angles from the 3 x 4 matrix.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace dbat {

struct RsCplx { double re, im; };
__host__ __device__ inline RsCplx rs_add(RsCplx a, RsCplx b) { return RsCplx{a.re + b.re, a.im + b.im}; }
__host__ __device__ inline RsCplx rs_sub(RsCplx a, RsCplx b) { return RsCplx{a.re - b.re, a.im - b.im}; }
__host__ __device__ inline RsCplx rs_mul(RsCplx a, RsCplx b) { return RsCplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__host__ __device__ inline RsCplx rs_div(RsCplx a, RsCplx b) {
    const double d = b.re * b.re + b.im * b.im;
    return RsCplx{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
__host__ __device__ inline double rs_abs(RsCplx a) { return hypot(a.re, a.im); }

// Roots of c[0] z^deg + ... + c[deg] (deg <= 4, c[0] != 0) by the Aberth-Ehrlich iteration (simultaneous
// Newton steps with mutual repulsion; cubic convergence on simple roots).  MATLAB's roots() takes the
// eigenvalues of the companion matrix; both return the roots to rounding where they are well conditioned,
// and a near-multiple root (camcal image 21: three roots within 1e-5) to about eps^(1/3) either way.
__host__ __device__ inline void rs_roots(const double *c, int deg, RsCplx *z) {
    double bound = 0;
    for (int i = 1; i <= deg; ++i) bound = fmax(bound, fabs(c[i] / c[0]));
    const double R = 1.0 + bound;
    for (int k = 0; k < deg; ++k) {                     // on a circle inside the Cauchy bound, off the axes
        const double th = 0.4 + 6.283185307179586 * k / deg;
        z[k] = RsCplx{0.5 * R * cos(th), 0.5 * R * sin(th)};
    }
    for (int it = 0; it < 200; ++it) {
        double move = 0, size = 0;
        for (int k = 0; k < deg; ++k) {
            RsCplx p{c[0], 0}, dp{0, 0};
            for (int i = 1; i <= deg; ++i) { dp = rs_add(rs_mul(dp, z[k]), p); p = rs_add(rs_mul(p, z[k]), RsCplx{c[i], 0}); }
            if (rs_abs(p) == 0) continue;
            const RsCplx w = rs_div(p, dp);
            RsCplx s{0, 0};
            for (int j = 0; j < deg; ++j) if (j != k) s = rs_add(s, rs_div(RsCplx{1, 0}, rs_sub(z[k], z[j])));
            const RsCplx step = rs_div(w, rs_sub(RsCplx{1, 0}, rs_mul(w, s)));
            if (!(step.re == step.re) || !(step.im == step.im)) continue;
            z[k] = rs_sub(z[k], step);
            move = fmax(move, rs_abs(step)); size = fmax(size, rs_abs(z[k]));
        }
        if (move <= 4e-16 * size) break;
    }
}

__host__ __device__ inline void rs_cross(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__host__ __device__ inline void rs_unit(double *a) {
    const double n = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] /= n; a[1] /= n; a[2] /= n;
}
// right-handed frame of the triangle (p0, pb, pc): columns r1 = ob/|ob|, r2 = ob x oc, r3 = ob x (ob x oc)
// (pm_resect_3pt.m:104-124); F[3*col + row]
__host__ __device__ inline void rs_frame(const double *p0, const double *pb, const double *pc, double *F) {
    double ob[3], oc[3], n1[3], n2[3];
    for (int i = 0; i < 3; ++i) { ob[i] = pb[i] - p0[i]; oc[i] = pc[i] - p0[i]; }
    rs_cross(ob, oc, n1); rs_cross(ob, n1, n2);
    rs_unit(ob); rs_unit(n1); rs_unit(n2);
    for (int i = 0; i < 3; ++i) { F[i] = ob[i]; F[3 + i] = n1[i]; F[6 + i] = n2[i]; }
}

// The candidate poses of one triangle: X[3][3] object points (X[i] = point i), d[3][3] unit rays.
// P[q]: 3 x 4 camera matrices, column-major (P[3*col + row]).  Returns their number (<= 4).
__host__ __device__ inline int rs_poses_3pt(const double X[3][3], const double d[3][3], bool behind, double P[4][12]) {
    auto dist = [](const double *p, const double *q) { return sqrt((p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2])); };
    auto cosl = [](const double *p, const double *q) { return fmin(1.0, fabs(p[0] * q[0] + p[1] * q[1] + p[2] * q[2])); };   // cos(subspace(p, q))
    const double a = dist(X[2], X[1]), b = dist(X[2], X[0]), c = dist(X[1], X[0]);
    const double ca = cosl(d[1], d[2]), cb = cosl(d[0], d[2]), cg = cosl(d[0], d[1]);
    const double b2 = b * b, m = (a * a - c * c) / b2, p = (a * a + c * c) / b2, bc = (b2 - c * c) / b2, ba = (b2 - a * a) / b2;
    double co[5];
    co[0] = (m - 1) * (m - 1) - 4 * c * c / b2 * ca * ca;
    co[1] = 4 * (m * (1 - m) * cb + 2 * c * c / b2 * ca * ca * cb - (1 - p) * ca * cg);
    co[2] = 2 * (m * m + 2 * m * m * cb * cb + 2 * bc * ca * ca + 2 * ba * cg * cg - 4 * p * ca * cb * cg - 1);
    co[3] = 4 * (-m * (1 + m) * cb + 2 * a * a / b2 * cg * cg * cb - (1 - p) * ca * cg);
    co[4] = (1 + m) * (1 + m) - 4 * a * a / b2 * cg * cg;
    int lead = 0;
    while (lead < 4 && !(fabs(co[lead]) > 0)) ++lead;   // (roots() drops leading zeros)
    const int deg = 4 - lead;
    if (deg < 1 || !(co[lead] == co[lead])) return 0;
    RsCplx z[4];
    rs_roots(co + lead, deg, z);
    double oR[9];
    rs_frame(X[0], X[2], X[1], oR);
    int np_ = 0;
    for (int k = 0; k < deg; ++k) {
        if (!(fabs(z[k].im) / rs_abs(z[k]) < 1e-3)) continue;      // pm_resect_3pt.m:80-82
        const double v = z[k].re;
        const double u = ((-1 + m) * v * v - 2 * m * cb * v + 1 + m) / (2 * (cg - v * ca));
        const double s1 = sqrt(b2 / (1 + v * v - 2 * v * cb));
        const double s3 = v * s1, s2 = u * s1;
        if (!(s1 >= 0 && s2 >= 0 && s3 >= 0)) continue;
        const double s[3] = {s1, s2, s3};
        double cx[3][3];
        for (int i = 0; i < 3; ++i) for (int r = 0; r < 3; ++r) cx[i][r] = (behind ? -s[i] : s[i]) * d[i][r];
        double cF[9], R[9];
        rs_frame(cx[0], cx[2], cx[1], cF);
        for (int r = 0; r < 3; ++r)                             // cRo = cF * oR'
            for (int q = 0; q < 3; ++q) R[3 * q + r] = cF[r] * oR[q] + cF[3 + r] * oR[3 + q] + cF[6 + r] * oR[6 + q];
        double ctr[3];
        for (int q = 0; q < 3; ++q) ctr[q] = X[0][q] - (R[3 * q] * cx[0][0] + R[3 * q + 1] * cx[0][1] + R[3 * q + 2] * cx[0][2]);   // X0 - cRo' cx0
        double *Pq = P[np_++];
        for (int i = 0; i < 9; ++i) Pq[i] = R[i];
        for (int r = 0; r < 3; ++r) Pq[9 + r] = -(R[r] * ctr[0] + R[3 + r] * ctr[1] + R[6 + r] * ctr[2]);
    }
    return np_;
}

// One wave per camera.  pt_start[c] .. pt_start[c+1]: the camera's check points (X object coordinates,
// xn normalised image coordinates); tri_start[c] .. tri_start[c+1]: its candidate triangles, three indices
// local to the camera's point range each, in the order resect.m tries them.
__global__ __launch_bounds__(64) void k_resect(int n_images, const int64_t *__restrict__ pt_start, const double *__restrict__ X,
                                               const double *__restrict__ xn, const int64_t *__restrict__ tri_start,
                                               const int32_t *__restrict__ tri, int behind, double *__restrict__ Pout,
                                               double *__restrict__ rms_out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= n_images) return;
    __shared__ double sP[4][12];
    __shared__ int s_np;
    const int64_t p0 = pt_start[c], npt = pt_start[c + 1] - p0;
    double best = INFINITY, bestP[12];
    for (int i = 0; i < 12; ++i) bestP[i] = NAN;
    for (int64_t t = tri_start[c]; t < tri_start[c + 1]; ++t) {
        if (lane == 0) {
            double Xt[3][3], d[3][3];
            for (int i = 0; i < 3; ++i) {
                const int64_t q = p0 + tri[3 * t + i];
                for (int r = 0; r < 3; ++r) Xt[i][r] = X[3 * q + r];
                d[i][0] = xn[2 * q]; d[i][1] = xn[2 * q + 1]; d[i][2] = 1.0;
                rs_unit(d[i]);
            }
            s_np = rs_poses_3pt(Xt, d, behind != 0, sP);
        }
        __syncthreads();
        const int np_ = s_np;
        double best_t = INFINITY;
        int arg = -1;
        for (int q = 0; q < np_; ++q) {                         // rms of the reprojection error over the check points
            double acc = 0;
            for (int64_t i = lane; i < npt; i += 64) {
                const double *Q = X + 3 * (p0 + i);
                const double h0 = sP[q][0] * Q[0] + sP[q][3] * Q[1] + sP[q][6] * Q[2] + sP[q][9];
                const double h1 = sP[q][1] * Q[0] + sP[q][4] * Q[1] + sP[q][7] * Q[2] + sP[q][10];
                const double h2 = sP[q][2] * Q[0] + sP[q][5] * Q[1] + sP[q][8] * Q[2] + sP[q][11];
                const double e0 = h0 / h2 - xn[2 * (p0 + i)], e1 = h1 / h2 - xn[2 * (p0 + i) + 1];
                acc += e0 * e0 + e1 * e1;
            }
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            const double r = sqrt(acc / (double)npt);
            if (r < best_t) { best_t = r; arg = q; }            // argmin: the first of equal minima
        }
        if (arg >= 0 && best_t < best) {                        // resect.m: a later triangle must be strictly better
            best = best_t;
            for (int i = 0; i < 12; ++i) bestP[i] = sP[arg][i];
        }
        __syncthreads();
    }
    if (lane == 0) {
        rms_out[c] = best;
        for (int i = 0; i < 12; ++i) Pout[12 * (int64_t)c + i] = bestP[i];
    }
}

}  // namespace dbat
