// Per-observation camera model for the bundle hot path (device + host).
//
// Closed-form, scalar restatement of the reference's chain of primitives for
// one image observation (paths relative to /root/reference/code/):
//   res_euler_brown_{0,1,2,3}.m   lens models 2..5 (brown_euler_cam4.m:127-130)
//   eulerpinhole2.m:51-67,97-106  world2cam.m:46-49,76-82  pinhole.m:39,54-66
//   eulerrotmat.m:81,109-124      scale2/aniscale2/aniscale2b/xlat2/affine2/skew
//   brown_dist.m:52-57,83-89      brown_rad.m:48-52,73-94  brown_tang.m:58-70,91-137
//   rad_scale.m:44-50,72-75       tang_scale.m:42-45,66-87
//
// v = lhs - rhs,  lhs = -f * pinhole(M'(Q-q0)),
// rhs = T_post * brown(T_pre(x), -K, -P),  x = [sz*u1 ; -sz*u2] - u0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dbat {

constexpr int MAXK = 5;            // max radial coefficients
constexpr int MAXP = 5;            // max tangential coefficients (P1,P2 + radial scaling terms)
constexpr int MAXIO = 5 + MAXK + MAXP;
constexpr int MAXCOL = 6 + MAXIO;  // camera-side columns of one observation

// Per-camera record, rebuilt from the parameter vector before every pass.
struct CamRec {
    double Mt[9];        // world->camera rotation M' (row-major), M = R1(om)R2(ph)R3(ka)
    double sk, ck;       // sin / cos of kappa: all that the angle derivatives need besides M' (angle_terms below)
    double c[3];         // camera centre
    double f;            // camera constant cc
    double pp[2];        // principal point
    double b[2];         // aspect, skew
    double K[MAXK];
    double P[MAXP];
    double sz;           // pixel size (pxSize(1,cam), multi_res.m:97,138)
    double w[2];         // 1/sigma_mm for x,y rows when IP.std is uniform per camera
    int32_t ncol;        // 6 + number of estimated IO rows of this camera
    int32_t col[MAXCOL]; // reduced-system column of each camera-side column
    int32_t iorow[MAXIO];// IO row behind column 6+j
    uint32_t eo_est;     // bit k set: EO(k) estimated
};

#define DBAT_HD __host__ __device__ __forceinline__

// eulerrotmat.m:81 (seq 123, moving axes) and the transposition of eulerpinhole2.m:54-60: M' with
// M = R1(omega) R2(phi) R3(kappa).  The derivative matrices of eulerrotmat.m:109-124 (dM/d omega = P1 M,
// dM/d phi = R1 R2 P2 R3, dM/d kappa = M P3) are never formed: their products with a vector follow from M'
// itself and (sin kappa, cos kappa) -- angle_terms below.
DBAT_HD void cam_rotation(const double ang[3], double Mt[9], double &sk_out, double &ck_out) {
    const double so = sin(ang[0]), co = cos(ang[0]);
    const double sp = sin(ang[1]), cp = cos(ang[1]);
    const double sk = sin(ang[2]), ck = cos(ang[2]);
    // M = R1 R2 R3 written out (R1 = rot_x(omega), R2 = rot_y(phi), R3 = rot_z(kappa)); M' row-major
    const double M00 = cp * ck,                  M01 = -cp * sk,                 M02 = sp;
    const double M10 = co * sk + so * sp * ck,   M11 = co * ck - so * sp * sk,   M12 = -so * cp;
    const double M20 = so * sk - co * sp * ck,   M21 = so * ck + co * sp * sk,   M22 = co * cp;
    Mt[0] = M00; Mt[1] = M10; Mt[2] = M20;
    Mt[3] = M01; Mt[4] = M11; Mt[5] = M21;
    Mt[6] = M02; Mt[7] = M12; Mt[8] = M22;
    sk_out = sk; ck_out = ck;
}

// y_k = d(M')/d(angle_k) * d for the three Euler angles (eulerpinhole2.m:100 with eulerrotmat.m:109-124), from
// X = M'd:  dM/d omega = P1 M      =>  y_omega = M' (P1' d) = M' (0, d2, -d1)
//           dM/d phi = R1 R2 P2 R3 =>  y_phi = R3' P2' R3 X = (-ck X2, sk X2, ck X0 - sk X1)
//           dM/d kappa = M P3      =>  y_kappa = P3' X = (X1, -X0, 0)
// (P1, P2, P3: the generators of the rotations about x, y, z).  19 operations and two constants where the
// three 3 x 3 derivative matrices cost 27 constants and 27 operations per observation.
DBAT_HD void angle_terms(const CamRec &cam, double d0, double d1, double d2, double X0, double X1, double X2,
                         double (&y)[3][3]) {
    (void)d0;
    y[0][0] = cam.Mt[1] * d2 - cam.Mt[2] * d1;
    y[0][1] = cam.Mt[4] * d2 - cam.Mt[5] * d1;
    y[0][2] = cam.Mt[7] * d2 - cam.Mt[8] * d1;
    y[1][0] = -cam.ck * X2; y[1][1] = cam.sk * X2; y[1][2] = cam.ck * X0 - cam.sk * X1;
    y[2][0] = X1; y[2][1] = -X0; y[2][2] = 0.0;
}

// acc + a b + c d as two accumulating FMAs.  (Written `acc += a * b + c * d` the compiler keeps the association of the
// source: a multiplication, an FMA and an addition; the observation kernels are bound by the vector instructions they issue.)
DBAT_HD double fma2(double acc, double a, double b, double c, double d) {
    return __builtin_fma(a, b, __builtin_fma(c, d, acc));
}

// 1/x: a reciprocal estimate and two Newton steps r <- r + r (1 - x r), each a pair of FMAs (5 operations; the IEEE
// division sequence takes 12).  v_rcp_f64 is good to about 2^-23, two steps give 1/x to within one unit in the last
// place (not always correctly rounded: the residual and Jacobian entries differ from a division's by at most that).
// x = 0, Inf, NaN give what 1.0 / x gives (Inf, 0, NaN: v_div_fixup_f64) -- the refinement alone would turn them all
// into NaN (Inf * (1 - 0 * Inf)).  The HOST evaluates the same sequence from 1.0 / x as its estimate, so that
// dbat_hip_debug_model_eval_host and bench/cpu_ref.cpp run the arithmetic of the kernels they check: there the steps
// find nothing to correct beyond the rounding of x r.
DBAT_HD double recip(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
#else
    double r = 1.0 / x;
#endif
    const double e1 = __builtin_fma(-x, r, 1.0);
    const double r1 = __builtin_fma(r, e1, r);
    const double e2 = __builtin_fma(-x, r1, 1.0);
    const double r2 = __builtin_fma(r1, e2, r1);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_div_fixup(r2, x, 1.0);      // v_div_fixup_f64: the special cases of 1.0 / x (x = 0, Inf, NaN), one instruction
#else
    return (r2 == r2 && r != 0.0 && r - r == 0.0) ? r2 : r;     // the same: a zero, infinite or NaN quotient is returned as it is
#endif
}

// Image-side part: rhs and its derivatives.  a = pre-distortion coordinates.
//   l    = brown_dist(a,-K,-P)
//   dLU  = d l / d a (2x2)
struct ImgSide {
    double rhs[2];
    double dU0[2][2];
    double dB[2][2];
    double dK[2][MAXK];
    double dP[2][MAXP];
};

// NKC, NPC >= 0: the numbers of radial / tangential coefficients are compile-time constants (the loops below
// unroll into straight-line code: no uniform branches between the dependent chains of two observations that the
// scheduler could otherwise interleave); < 0: run-time values nK, nP.
template <int MODEL, bool JAC, int NKC, int NPC>
DBAT_HD void image_side_impl(const CamRec &cam, int nK_, int nP_, double u, double v, ImgSide &o) {
    const int nK = NKC >= 0 ? NKC : nK_, nP = NPC >= 0 ? NPC : nP_;
    // scale2, aniscale2([1;-1]), (aniscale2b for model 5), xlat2(-u0)
    const double s0 = cam.sz * u, s1 = -cam.sz * v;
    double x0, x1;
    if (MODEL == 5) { x0 = (1.0 + cam.b[0]) * s0 - cam.pp[0]; x1 = s1 - cam.pp[1]; }
    else            { x0 = s0 - cam.pp[0];                    x1 = s1 - cam.pp[1]; }
    // T_pre: affine2 before distortion (model 3)
    double a0 = x0, a1 = x1;
    if (MODEL == 3) { a0 = (1.0 + cam.b[0]) * x0 + cam.b[1] * x1; }
    // brown_dist(a, -K, -P)
    const double rho = a0 * a0 + a1 * a1;
    double rs = 0, drs = 0, pw = 1.0;            // rs = sum Kn_j rho^j ; drs = sum j Kn_j rho^(j-1)
    double rpow[MAXK + 1];
    rpow[0] = 1.0;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        if (j < nK) {
            const double kn = -cam.K[j];
            drs += (j + 1) * kn * pw;
            pw *= rho;
            rs += kn * pw;
        }
        rpow[j + 1] = pw;
    }
    double t0 = 0, t1 = 0, ts0 = 0, ts1 = 0, rs2 = 0, drs2 = 0, pTu = 0, pn0 = 0, pn1 = 0;
    if (nP >= 2) {
        pn0 = -cam.P[0]; pn1 = -cam.P[1];
        pTu = pn0 * a0 + pn1 * a1;
        ts0 = pn0 * rho + 2 * pTu * a0;
        ts1 = pn1 * rho + 2 * pTu * a1;
        double q = 1.0;
#pragma unroll
        for (int j = 2; j < MAXP; ++j)
            if (j < nP) {
                const double pn = -cam.P[j];
                drs2 += (j - 1) * pn * q;
                q *= rho;
                rs2 += pn * q;
            }
        t0 = ts0 * (1 + rs2);
        t1 = ts1 * (1 + rs2);
    }
    const double l0 = a0 + a0 * rs + t0;
    const double l1 = a1 + a1 * rs + t1;
    // T_post
    if (MODEL == 4)      { o.rhs[0] = (1.0 + cam.b[0]) * l0 + cam.b[1] * l1; o.rhs[1] = l1; }
    else if (MODEL == 5) { o.rhs[0] = l0 + cam.b[1] * l1;                    o.rhs[1] = l1; }
    else                 { o.rhs[0] = l0;                                    o.rhs[1] = l1; }
    if (!JAC) return;
    // dL/dU = I + (a*(2 drs a') + rs I) + d tang / dU
    double L00 = 1 + rs + 2 * drs * a0 * a0, L01 = 2 * drs * a0 * a1;
    double L10 = 2 * drs * a1 * a0,          L11 = 1 + rs + 2 * drs * a1 * a1;
    if (nP >= 2) {
        const double m = 1 + rs2;
        const double d00 = 2 * (2 * pn0 * a0 + pTu), d01 = 2 * (pn0 * a1 + pn1 * a0);
        const double d11 = 2 * (2 * pn1 * a1 + pTu);
        L00 += d00 * m + ts0 * drs2 * 2 * a0;  L01 += d01 * m + ts0 * drs2 * 2 * a1;
        L10 += d01 * m + ts1 * drs2 * 2 * a0;  L11 += d11 * m + ts1 * drs2 * 2 * a1;
    }
    // post transform matrix T (2x2, second row = [0 1])
    double T00 = 1, T01 = 0;
    if (MODEL == 4) { T00 = 1.0 + cam.b[0]; T01 = cam.b[1]; }
    if (MODEL == 5) { T01 = cam.b[1]; }
    const double TL00 = T00 * L00 + T01 * L10, TL01 = T00 * L01 + T01 * L11;
    const double TL10 = L10, TL11 = L11;
    // dv/du0 = +T*dL*Apre   (res_euler_brown_1.m:167-169)
    if (MODEL == 3) {
        const double A00 = 1.0 + cam.b[0], A01 = cam.b[1];
        o.dU0[0][0] = TL00 * A00; o.dU0[0][1] = TL00 * A01 + TL01;
        o.dU0[1][0] = TL10 * A00; o.dU0[1][1] = TL10 * A01 + TL11;
    } else {
        o.dU0[0][0] = TL00; o.dU0[0][1] = TL01;
        o.dU0[1][0] = TL10; o.dU0[1][1] = TL11;
    }
    // dv/dK = T * a * rho^j   (brown_rad.m:76-79 ; sign: -K passed, v = lhs - l)
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        if (j < nK) {
            const double k0 = a0 * rpow[j + 1], k1 = a1 * rpow[j + 1];
            o.dK[0][j] = T00 * k0 + T01 * k1;
            o.dK[1][j] = k1;
        } else { o.dK[0][j] = 0; o.dK[1][j] = 0; }
    }
    // dv/dP  (tang_scale.m:66-72, brown_tang.m:95-104)
    for (int j = 0; j < MAXP; ++j) { o.dP[0][j] = 0; o.dP[1][j] = 0; }
    if (nP >= 2) {
        const double m = 1 + rs2;
        const double p00 = m * (rho + 2 * a0 * a0), p01 = m * 2 * a0 * a1;
        const double p11 = m * (rho + 2 * a1 * a1);
        o.dP[0][0] = T00 * p00 + T01 * p01; o.dP[0][1] = T00 * p01 + T01 * p11;
        o.dP[1][0] = p01;                   o.dP[1][1] = p11;
        double q = 1.0;
#pragma unroll
        for (int j = 2; j < MAXP; ++j)
            if (j < nP) {
                q *= rho;
                o.dP[0][j] = T00 * ts0 * q + T01 * ts1 * q;
                o.dP[1][j] = ts1 * q;
            }
    }
    // dv/db
    if (MODEL == 2) {
        o.dB[0][0] = o.dB[0][1] = o.dB[1][0] = o.dB[1][1] = 0;
    } else if (MODEL == 3) {      // -dL.dU*dA.dB, dA.dB = [x0 x1; 0 0]  (res_euler_brown_1.m:176-178)
        o.dB[0][0] = -L00 * x0; o.dB[0][1] = -L00 * x1;
        o.dB[1][0] = -L10 * x0; o.dB[1][1] = -L10 * x1;
    } else if (MODEL == 4) {      // -dA.dB, rows [l0 l1; 0 0]
        o.dB[0][0] = -l0; o.dB[0][1] = -l1;
        o.dB[1][0] = 0;   o.dB[1][1] = 0;
    } else {                      // model 5: -[SK*dL*[s0;0], [l1;0]]  (res_euler_brown_3.m:180)
        o.dB[0][0] = -TL00 * s0; o.dB[0][1] = -l1;
        o.dB[1][0] = -TL10 * s0; o.dB[1][1] = 0;
    }
}

// the usual lens (K1-K3, P1-P2: every camera of the synthetic configurations, the roma camera, PhotoModeler's
// default) takes the straight-line instantiation; one wave-uniform test
template <int MODEL, bool JAC>
DBAT_HD void image_side(const CamRec &cam, int nK, int nP, double u, double v, ImgSide &o) {
    if (nK == 3 && nP == 2) image_side_impl<MODEL, JAC, 3, 2>(cam, nK, nP, u, v, o);
    else image_side_impl<MODEL, JAC, -1, -1>(cam, nK, nP, u, v, o);
}

// Full observation: residual r[2] (unweighted, mm) and Jacobian blocks
//   A[2][6]  wrt EO = [centre(3) angles(3)]   (dQ0, dA)
//   B[2][3]  wrt OP                           (dQ)
//   C[2][nIOrows] wrt IO rows [cc px py as sk K.. P..] (only if WITH_IO)
//   PRE: (u, v) already hold rhs -- the corrected image coordinates of a problem whose interior orientation
//   is fixed do not change between iterations and are computed once (k_uv_to_rhs, kernels.hpp)
template <int MODEL, bool JAC, bool WITH_IO, bool PRE = false>
DBAT_HD void obs_eval(const CamRec &cam, int nK, int nP, const double Q[3], double u, double v,
                      double r[2], double A[2][6], double B[2][3], double C[2][MAXIO]) {
    static_assert(!(PRE && WITH_IO), "precomputed image side: fixed interior orientation only");
    const double d0 = Q[0] - cam.c[0], d1 = Q[1] - cam.c[1], d2 = Q[2] - cam.c[2];
    const double X0 = cam.Mt[0] * d0 + cam.Mt[1] * d1 + cam.Mt[2] * d2;
    const double X1 = cam.Mt[3] * d0 + cam.Mt[4] * d1 + cam.Mt[5] * d2;
    const double X2 = cam.Mt[6] * d0 + cam.Mt[7] * d1 + cam.Mt[8] * d2;
    const double iz = recip(X2);
    const double ph0 = X0 * iz, ph1 = X1 * iz;
    ImgSide im;
    if (PRE) { im.rhs[0] = u; im.rhs[1] = v; }
    else image_side<MODEL, JAC && WITH_IO>(cam, nK, nP, u, v, im);
    const double nf = -cam.f;
    r[0] = nf * ph0 - im.rhs[0];
    r[1] = nf * ph1 - im.rhs[1];
    if (!JAC) return;
    // Pi = (1/X2)[1 0 -ph0; 0 1 -ph1]  (pinhole.m:54-66); dv/dQ = -f*Pi*M'
    const double s = nf * iz;
    for (int k = 0; k < 3; ++k) {
        const double m0 = cam.Mt[k], m1 = cam.Mt[3 + k], m2 = cam.Mt[6 + k];
        const double b0 = s * (m0 - ph0 * m2), b1 = s * (m1 - ph1 * m2);
        B[0][k] = b0;  B[1][k] = b1;
        A[0][k] = -b0; A[1][k] = -b1;              // world2cam.m:82  dP0 = -M
    }
    double y[3][3];
    angle_terms(cam, d0, d1, d2, X0, X1, X2, y);   // eulerpinhole2.m:100
    for (int k = 0; k < 3; ++k) {
        A[0][3 + k] = s * (y[k][0] - ph0 * y[k][2]);
        A[1][3 + k] = s * (y[k][1] - ph1 * y[k][2]);
    }
    if (WITH_IO) {
        C[0][0] = -ph0; C[1][0] = -ph1;            // dv/df = -vec(PH)
        C[0][1] = im.dU0[0][0]; C[0][2] = im.dU0[0][1];
        C[1][1] = im.dU0[1][0]; C[1][2] = im.dU0[1][1];
        C[0][3] = im.dB[0][0];  C[0][4] = im.dB[0][1];
        C[1][3] = im.dB[1][0];  C[1][4] = im.dB[1][1];
        // rows 5.. : K1..KnK then P1..PnP.  nK is a run-time value: place the P rows with
        // compile-time indices and selects (a dynamic index would push C into scratch memory).
        // The usual layout (K1-K3, P1-P2: wave-uniform test) needs no selects at all.
        if (nK == 3 && nP == 2) {
            C[0][5] = im.dK[0][0]; C[1][5] = im.dK[1][0]; C[0][6] = im.dK[0][1]; C[1][6] = im.dK[1][1];
            C[0][7] = im.dK[0][2]; C[1][7] = im.dK[1][2];
            C[0][8] = im.dP[0][0]; C[1][8] = im.dP[1][0]; C[0][9] = im.dP[0][1]; C[1][9] = im.dP[1][1];
#pragma unroll
            for (int rr = 10; rr < MAXIO; ++rr) { C[0][rr] = 0.0; C[1][rr] = 0.0; }
        } else
#pragma unroll
        for (int rr = 5; rr < MAXIO; ++rr) {
            double c0 = 0.0, c1 = 0.0;
            const int j = rr - 5;
            if (j < MAXK && j < nK) { c0 = im.dK[0][j < MAXK ? j : 0]; c1 = im.dK[1][j < MAXK ? j : 0]; }
#pragma unroll
            for (int jp = 0; jp < MAXP; ++jp)
                if (jp < nP && rr == 5 + nK + jp) { c0 = im.dP[0][jp]; c1 = im.dP[1][jp]; }
            C[0][rr] = c0; C[1][rr] = c1;
        }
    }
}

// Back-substitution, the usual self-calibration (camera-side columns = 6 EO + cc px py K1 K2 K3 P1 P2,
// CamRec::eo_est bit 8; nK = 3, nP = 2): t = E dc for one observation and its weighted point block B,
// straight from the pieces of the model -- no 2 x 14 block E and no 2 x 15 block of IO derivatives in
// registers (k_backsub_sig<., 14> used 233 of them and ran two waves per SIMD).
// dc[0..5]: step of the camera's EO columns (0 for fixed ones), dc[6..13]: of its eight IO columns.
// IO8 = false: fixed interior orientation, the EO part alone.
template <int MODEL, bool IO8>
DBAT_HD void obs_step_dot(const CamRec &cam, const double Q[3], double u, double v, double w0, double w1,
                           unsigned est, const double *dc, double &t0, double &t1, double B[2][3]) {
    const double d0 = Q[0] - cam.c[0], d1 = Q[1] - cam.c[1], d2 = Q[2] - cam.c[2];
    const double X0 = cam.Mt[0] * d0 + cam.Mt[1] * d1 + cam.Mt[2] * d2;
    const double X1 = cam.Mt[3] * d0 + cam.Mt[4] * d1 + cam.Mt[5] * d2;
    const double X2 = cam.Mt[6] * d0 + cam.Mt[7] * d1 + cam.Mt[8] * d2;
    const double iz = recip(X2);
    const double ph0 = X0 * iz, ph1 = X1 * iz;
    const double s = -cam.f * iz;
    double a0 = 0, a1 = 0;                       // A dc (EO part), unweighted
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double m0 = cam.Mt[k], m1 = cam.Mt[3 + k], m2 = cam.Mt[6 + k];
        const double b0 = s * (m0 - ph0 * m2), b1 = s * (m1 - ph1 * m2);
        const double m = ((est >> k) & 1u) ? 1.0 : 0.0;
        B[0][k] = b0 * w0 * m; B[1][k] = b1 * w1 * m;
        const double mk = ((cam.eo_est >> k) & 1u) ? dc[k] : 0.0;
        a0 -= b0 * mk; a1 -= b1 * mk;             // A(:,k) = -B(:,k)  (world2cam.m:82)
    }
    double y[3][3];
    angle_terms(cam, d0, d1, d2, X0, X1, X2, y);  // eulerpinhole2.m:100
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double mk = ((cam.eo_est >> (3 + k)) & 1u) ? dc[3 + k] : 0.0;
        a0 += s * (y[k][0] - ph0 * y[k][2]) * mk;
        a1 += s * (y[k][1] - ph1 * y[k][2]) * mk;
    }
    if (!IO8) { t0 = a0 * w0; t1 = a1 * w1; return; }      // fixed IO: dc[0..5] only
    ImgSide im;
    image_side<MODEL, true>(cam, 3, 2, u, v, im);
    // IO rows cc | px py | K1 K2 K3 | P1 P2  (obs_eval: C(:,0) = -ph, C(:,1:2) = dU0, C(:,5:7) = dK, C(:,8:9) = dP)
    a0 += -ph0 * dc[6] + im.dU0[0][0] * dc[7] + im.dU0[0][1] * dc[8] + im.dK[0][0] * dc[9] + im.dK[0][1] * dc[10] + im.dK[0][2] * dc[11]
        + im.dP[0][0] * dc[12] + im.dP[0][1] * dc[13];
    a1 += -ph1 * dc[6] + im.dU0[1][0] * dc[7] + im.dU0[1][1] * dc[8] + im.dK[1][0] * dc[9] + im.dK[1][1] * dc[10] + im.dK[1][2] * dc[11]
        + im.dP[1][0] * dc[12] + im.dP[1][1] * dc[13];
    t0 = a0 * w0; t1 = a1 * w1;
}

}  // namespace dbat
