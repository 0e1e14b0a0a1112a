// CPU baseline of the bundle hot path: the reference algorithm AS WRITTEN, in
// C++17 + OpenMP on all host cores.  Timed by bench.py's `cpu_baseline` leg and
// checked against the oracle in tests/test_cpu_ref.py; never on the product path
// (libdbat_hip.so does not link or load it).
//
// What the reference does per iteration (paths relative to /root/reference/code/):
//   bundle/cameramodel/brown_euler_cam4.m:122-183, multi_res.m:20-315
//       residual vector f and an EXPLICIT sparse Jacobian J (here CSC, weighted by
//       R = chol(W) = diag(1/sigma), misc/buildweightmatrix.m:13-43,
//       gauss_newton_armijo.m:104)
//   bundle/lsa/levenberg_marquardt.m:81-82   JTJ = J'*J ; JTr = J'*r   (general sparse product)
//   bundle/lsa/levenberg_marquardt.m:119     p = (JTJ + lambda*I) \ (-JTr)
//       MATLAB's `\` on a sparse SPD matrix = CHOLMOD: fill-reducing ordering +
//       supernodal LL' of the FULL normal matrix (not a hand-made Schur complement)
//   bundle/lsa/levenberg_marquardt.m:160-170 residual at the trial point x+p
//
// The sparse Cholesky below is a supernodal LL' of the full (n x n) matrix with the
// elimination order [object points | EO | IO] (what a minimum-degree ordering gives on
// this matrix: a point column has 2 + 6 k_p + nIO neighbours, a camera column thousands).
// Leaf supernodes = the columns of one object point; the root supernode = every
// camera/IO column, stored as a dense lower triangle whose structurally empty
// 128 x 128 tiles are skipped (tile-level symbolic factorisation).  Left-looking
// updates of the root by the leaves, then a blocked right-looking dense factorisation.
//
// The per-observation model is the closed form of dbat_amd/csrc/model.hpp (also
// compiled for the host); unknown ordering and sharing follow
// misc/buildserialindices.m:57,108-128,162-221 (x = [IO; EO; OP], column-major,
// shared block parameters once).  Prior observations (lsa/prior_obs.m:45-72: rows x(dest) - prior.val(src) with a
// selection matrix as Jacobian, weighted by 1/std, misc/buildweightmatrix.m:25-29) are the rows of J below the image
// rows -- one entry each, kept as (column, weight, value) triplets: [IO priors; EO priors; OP priors], as
// buildserialindices.m:151-159 orders them.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../dbat_amd/csrc/model.hpp"

using namespace dbat;

extern "C" {
struct cpuref_problem {
    int32_t nc, np, no, nIOrows, model, nK, nP, reserved;
    const int32_t *cam, *pt;     // per observation (image-major, ascending OP), 0-based
    const double *uv;            // 2 x no, pixels
    const double *std;           // 2 x no, pixels (IP.std)
    const double *px;            // nc, pixel size (mm)
    const double *IO;            // nIOrows x nc
    const double *EO;            // 6 x nc
    const double *OP;            // 3 x np
    const uint8_t *estIO, *estEO, *estOP;
    const int32_t *IOblock;      // nIOrows x nc
    // prior observations (all three may be null): use flags, values and standard deviations, shaped as IO / EO / OP
    const uint8_t *useIO, *useEO, *useOP;
    const double *priorIO, *priorEO, *priorOP, *stdIO, *stdEO, *stdOP;
};
}

namespace {

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

constexpr int MAXC = 6 + MAXIO + 3;   // columns of one observation's two rows
constexpr int NB = 128;               // tile of the root supernode

struct Ref {
    int nc = 0, np = 0, no = 0, nIOrows = 0, model = 3, nK = 0, nP = 0, nthreads = 1;
    std::vector<int32_t> cam, pt;
    std::vector<double> uv, w, px, IO, EO, OP;
    std::vector<uint8_t> estIO, estEO, estOP;
    std::vector<int32_t> IOblock;
    std::vector<int64_t> IOix, EOix, OPix;      // array entry -> x index or -1
    std::vector<int64_t> IOsrc;                 // x index (IO part) -> array entry it is read from
    int64_t n = 0, nIO = 0, nEO = 0, nOP = 0, NS = 0;
    // explicit J (CSC, weighted), plus the row-wise view (two rows of one observation share their columns)
    std::vector<int64_t> Jp;
    std::vector<int32_t> Ji;
    std::vector<double> Jx;
    std::vector<int8_t> ocnt;
    std::vector<int64_t> oslot;                 // [no][MAXC] slot of row 2o in column ocol
    std::vector<int64_t> ocol;                  // [no][MAXC] x index
    std::vector<double> r;                      // weighted residual (2*no image rows, then the prior rows)
    // prior observation rows: x index, 1/std, prior value (row 2*no + k of J has the single entry pr_w[k] in column pr_col[k])
    std::vector<int64_t> pr_col;
    std::vector<double> pr_w, pr_val;
    std::vector<int64_t> pr_ptr;                // per x index: its prior rows (CSR over the unknowns; empty without priors)
    std::vector<int64_t> pr_rows;
    // elimination order: q[x index] = position; points first, then EO, then IO
    std::vector<int64_t> q, qinv;
    // N = J'J + lambda I, lower triangle in elimination order, CSC
    std::vector<int64_t> Np;
    std::vector<int32_t> Ni;
    std::vector<double> Nx;
    // supernodes of the object points
    int64_t npts_cols = 0;                      // number of point columns (= n - NS)
    std::vector<int64_t> sn_col0;               // first column (elimination order) of each point supernode (+ end)
    std::vector<int64_t> sn_rptr;               // offset of the supernode's off-diagonal row list
    std::vector<int32_t> sn_rows;               // root-local row indices (sorted)
    std::vector<double> LD;                     // 9 per supernode (3x3 lower, column-major, padded)
    std::vector<double> LR;                     // [rows][3] per supernode
    // supernodes with identical off-diagonal rows (object points seen by the same images) are
    // amalgamated for the update of the root: one dense block per group (CHOLMOD merges such
    // columns into one supernode)
    std::vector<int64_t> grp_ptr;
    std::vector<int32_t> grp_sn;
    // root supernode
    std::vector<double> T;                      // NS x NS lower, column-major
    int nt = 0;
    std::vector<uint8_t> tnz;                   // nt x nt tile pattern after symbolic factorisation
    std::vector<double> g, b, xsol;
    std::string err;
};

static void cam_record(const Ref &R, const double *IOv, const double *EOv, int c, CamRec &cr) {
    cr.c[0] = EOv[6 * c]; cr.c[1] = EOv[6 * c + 1]; cr.c[2] = EOv[6 * c + 2];
    cam_rotation(EOv + 6 * c + 3, cr.Mt, cr.sk, cr.ck);
    const double *io = IOv + (size_t)R.nIOrows * c;
    cr.f = io[0]; cr.pp[0] = io[1]; cr.pp[1] = io[2]; cr.b[0] = io[3]; cr.b[1] = io[4];
    for (int k = 0; k < MAXK; ++k) cr.K[k] = k < R.nK ? io[5 + k] : 0.0;
    for (int k = 0; k < MAXP; ++k) cr.P[k] = k < R.nP ? io[5 + R.nK + k] : 0.0;
    cr.sz = R.px[c];
}

template <bool JAC>
static void eval_obs(const Ref &R, const CamRec &cr, const double *Q, double u, double v, double rr[2], double A[2][6],
                     double B[2][3], double C[2][MAXIO]) {
    switch (R.model) {
        case 2: obs_eval<2, JAC, JAC>(cr, R.nK, R.nP, Q, u, v, rr, A, B, C); break;
        case 3: obs_eval<3, JAC, JAC>(cr, R.nK, R.nP, Q, u, v, rr, A, B, C); break;
        case 4: obs_eval<4, JAC, JAC>(cr, R.nK, R.nP, Q, u, v, rr, A, B, C); break;
        default: obs_eval<5, JAC, JAC>(cr, R.nK, R.nP, Q, u, v, rr, A, B, C); break;
    }
}

// x -> full parameter arrays (deserialize.m:28-30: shared block entries fan out)
static void deserialize(const Ref &R, const double *x, std::vector<double> &IOv, std::vector<double> &EOv,
                        std::vector<double> &OPv) {
    IOv = R.IO; EOv = R.EO; OPv = R.OP;
    for (size_t e = 0; e < R.IOix.size(); ++e) if (R.IOix[e] >= 0) IOv[e] = x[R.IOix[e]];
    for (size_t e = 0; e < R.EOix.size(); ++e) if (R.EOix[e] >= 0) EOv[e] = x[R.EOix[e]];
    for (size_t e = 0; e < R.OPix.size(); ++e) if (R.OPix[e] >= 0) OPv[e] = x[R.OPix[e]];
}

// residual (and J values into the CSC slots) at x; returns 0.5 r'r
static double residual_jacobian(Ref &R, const double *x, bool jac, std::vector<double> &rout) {
    std::vector<double> IOv, EOv, OPv;
    deserialize(R, x, IOv, EOv, OPv);
    std::vector<CamRec> cams((size_t)R.nc);
#pragma omp parallel for schedule(static) num_threads(R.nthreads)
    for (int c = 0; c < R.nc; ++c) cam_record(R, IOv.data(), EOv.data(), c, cams[c]);
    const int64_t npr = (int64_t)R.pr_col.size();
    rout.resize((size_t)2 * R.no + npr);
    double f = 0;
    for (int64_t k = 0; k < npr; ++k) {                      // prior_obs.m:45-72, weighted
        const double v = R.pr_w[k] * (x[R.pr_col[k]] - R.pr_val[k]);
        rout[(size_t)2 * R.no + k] = v;
        f += v * v;
    }
#pragma omp parallel for schedule(static) reduction(+ : f) num_threads(R.nthreads)
    for (int64_t o = 0; o < R.no; ++o) {
        const int c = R.cam[o], p = R.pt[o];
        double rr[2], A[2][6] = {}, B[2][3] = {}, C[2][MAXIO] = {};
        if (jac) eval_obs<true>(R, cams[c], &OPv[(size_t)3 * p], R.uv[2 * o], R.uv[2 * o + 1], rr, A, B, C);
        else eval_obs<false>(R, cams[c], &OPv[(size_t)3 * p], R.uv[2 * o], R.uv[2 * o + 1], rr, A, B, C);
        const double w0 = R.w[2 * o], w1 = R.w[2 * o + 1];
        rout[2 * o] = w0 * rr[0]; rout[2 * o + 1] = w1 * rr[1];
        f += rout[2 * o] * rout[2 * o] + rout[2 * o + 1] * rout[2 * o + 1];
        if (!jac) continue;
        // multi_res.m:143-313: pack the blocks of this observation into J
        const int64_t *slot = &R.oslot[(size_t)o * MAXC];
        int kk = 0;
        for (int k = 0; k < R.nIOrows; ++k)
            if (R.IOix[(size_t)R.nIOrows * c + k] >= 0) { R.Jx[slot[kk]] = w0 * C[0][k]; R.Jx[slot[kk] + 1] = w1 * C[1][k]; ++kk; }
        for (int k = 0; k < 6; ++k)
            if (R.EOix[(size_t)6 * c + k] >= 0) { R.Jx[slot[kk]] = w0 * A[0][k]; R.Jx[slot[kk] + 1] = w1 * A[1][k]; ++kk; }
        for (int k = 0; k < 3; ++k)
            if (R.OPix[(size_t)3 * p + k] >= 0) { R.Jx[slot[kk]] = w0 * B[0][k]; R.Jx[slot[kk] + 1] = w1 * B[1][k]; ++kk; }
    }
    return 0.5 * f;
}

// buildserialindices.m:162-221 for one parameter kind: column-major scan, the first
// estimated entry of a (row, block id) pair is the unknown, later ones map to it
static int64_t serial_indices(int rows, int cols, const uint8_t *est, const int32_t *block, int64_t base,
                              std::vector<int64_t> &ix, std::vector<int64_t> *src) {
    ix.assign((size_t)rows * cols, -1);
    int64_t cnt = 0;
    std::map<std::pair<int, int32_t>, int64_t> lead;
    for (int c = 0; c < cols; ++c)
        for (int r = 0; r < rows; ++r) {
            const size_t e = (size_t)rows * c + r;
            if (!est[e]) continue;
            if (block) {
                auto key = std::make_pair(r, block[e]);
                auto it = lead.find(key);
                if (it != lead.end()) { ix[e] = it->second; continue; }
                lead[key] = base + cnt;
            }
            ix[e] = base + cnt;
            if (src) src->push_back((int64_t)e);
            ++cnt;
        }
    return cnt;
}

static bool setup(Ref &R) {
    // ---- unknowns: x = [IO; EO; OP]
    R.nIO = serial_indices(R.nIOrows, R.nc, R.estIO.data(), R.IOblock.data(), 0, R.IOix, &R.IOsrc);
    R.nEO = serial_indices(6, R.nc, R.estEO.data(), nullptr, R.nIO, R.EOix, nullptr);
    R.nOP = serial_indices(3, R.np, R.estOP.data(), nullptr, R.nIO + R.nEO, R.OPix, nullptr);
    R.n = R.nIO + R.nEO + R.nOP;
    R.NS = R.nIO + R.nEO;
    if (R.n >= (int64_t)1 << 31) { R.err = "too many unknowns for 32-bit row indices"; return false; }
    // ---- the unknowns with a prior observation each own one row per use of theirs
    R.pr_ptr.assign((size_t)R.n + 1, 0);
    for (int64_t k = 0; k < (int64_t)R.pr_col.size(); ++k) ++R.pr_ptr[R.pr_col[k] + 1];
    for (int64_t j = 0; j < R.n; ++j) R.pr_ptr[j + 1] += R.pr_ptr[j];
    R.pr_rows.resize(R.pr_col.size());
    {
        std::vector<int64_t> fill(R.pr_ptr.begin(), R.pr_ptr.end() - 1);
        for (int64_t k = 0; k < (int64_t)R.pr_col.size(); ++k) R.pr_rows[fill[R.pr_col[k]]++] = k;
    }
    // ---- weights (buildweightmatrix.m:20): sigma[mm] = IP.std[px] * pxSize
    // (R.w holds IP.std on entry)
    for (int64_t o = 0; o < R.no; ++o) {
        const double s = R.px[R.cam[o]];
        R.w[2 * o] = 1.0 / (R.w[2 * o] * s); R.w[2 * o + 1] = 1.0 / (R.w[2 * o + 1] * s);
    }
    // ---- pattern of J (CSC; rows ascend because observations are visited in row order)
    R.Jp.assign((size_t)R.n + 1, 0);
    R.ocnt.assign((size_t)R.no, 0);
    R.ocol.assign((size_t)R.no * MAXC, -1);
    R.oslot.assign((size_t)R.no * MAXC, -1);
    for (int64_t o = 0; o < R.no; ++o) {
        const int c = R.cam[o], p = R.pt[o];
        int kk = 0;
        int64_t *col = &R.ocol[(size_t)o * MAXC];
        for (int k = 0; k < R.nIOrows; ++k) { const int64_t j = R.IOix[(size_t)R.nIOrows * c + k]; if (j >= 0) col[kk++] = j; }
        for (int k = 0; k < 6; ++k) { const int64_t j = R.EOix[(size_t)6 * c + k]; if (j >= 0) col[kk++] = j; }
        for (int k = 0; k < 3; ++k) { const int64_t j = R.OPix[(size_t)3 * p + k]; if (j >= 0) col[kk++] = j; }
        R.ocnt[o] = (int8_t)kk;
        for (int k = 0; k < kk; ++k) R.Jp[col[k] + 1] += 2;
    }
    for (int64_t j = 0; j < R.n; ++j) R.Jp[j + 1] += R.Jp[j];
    R.Ji.resize((size_t)R.Jp[R.n]);
    R.Jx.assign((size_t)R.Jp[R.n], 0.0);
    {
        std::vector<int64_t> fill(R.Jp.begin(), R.Jp.end() - 1);
        for (int64_t o = 0; o < R.no; ++o)
            for (int k = 0; k < R.ocnt[o]; ++k) {
                const int64_t j = R.ocol[(size_t)o * MAXC + k];
                const int64_t s = fill[j];
                R.oslot[(size_t)o * MAXC + k] = s;
                R.Ji[s] = (int32_t)(2 * o); R.Ji[s + 1] = (int32_t)(2 * o + 1);
                fill[j] += 2;
            }
    }
    // ---- elimination order
    R.q.resize((size_t)R.n); R.qinv.resize((size_t)R.n);
    R.npts_cols = R.nOP;
    for (int64_t j = 0; j < R.n; ++j) {
        int64_t pos;
        if (j >= R.NS) pos = j - R.NS;                          // OP first
        else if (j >= R.nIO) pos = R.nOP + (j - R.nIO);         // then EO
        else pos = R.nOP + R.nEO + j;                           // IO last
        R.q[j] = pos; R.qinv[pos] = j;
    }
    // ---- symbolic J'J (lower triangle, elimination order)
    R.Np.assign((size_t)R.n + 1, 0);
    {
        // two passes with per-thread markers: count, then fill
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                for (int64_t j = 0; j < R.n; ++j) R.Np[j + 1] += R.Np[j];
                R.Ni.resize((size_t)R.Np[R.n]);
            }
#pragma omp parallel num_threads(R.nthreads)
            {
                std::vector<int64_t> mark((size_t)R.n, -1);
                std::vector<int32_t> rows;
#pragma omp for schedule(dynamic, 64)
                for (int64_t jq = 0; jq < R.n; ++jq) {
                    const int64_t j = R.qinv[jq];
                    rows.clear();
                    for (int64_t s = R.Jp[j]; s < R.Jp[j + 1]; s += 2) {
                        const int64_t o = R.Ji[s] >> 1;
                        const int64_t *col = &R.ocol[(size_t)o * MAXC];
                        for (int k = 0; k < R.ocnt[o]; ++k) {
                            const int64_t iq = R.q[col[k]];
                            if (iq >= jq && mark[iq] != jq) { mark[iq] = jq; rows.push_back((int32_t)iq); }
                        }
                    }
                    if (mark[jq] != jq) { mark[jq] = jq; rows.push_back((int32_t)jq); }   // the diagonal (lambda)
                    if (pass == 0) R.Np[jq + 1] = (int64_t)rows.size();
                    else {
                        std::sort(rows.begin(), rows.end());
                        std::copy(rows.begin(), rows.end(), R.Ni.begin() + R.Np[jq]);
                    }
                }
            }
        }
        R.Nx.assign((size_t)R.Np[R.n], 0.0);
    }
    // ---- supernodes of the object points: consecutive point columns with the same off-diagonal rows
    {
        R.sn_col0.clear(); R.sn_rptr.assign(1, 0);
        int64_t jq = 0;
        while (jq < R.npts_cols) {
            int64_t w = 1;
            auto offdiag_begin = [&](int64_t c) {
                int64_t s = R.Np[c];
                while (s < R.Np[c + 1] && R.Ni[s] < R.npts_cols) ++s;
                return s;
            };
            const int64_t s0 = offdiag_begin(jq), len0 = R.Np[jq + 1] - s0;
            while (w < 3 && jq + w < R.npts_cols) {
                const int64_t c = jq + w;
                // same supernode iff column jq has c as a row and the off-diagonal rows agree
                bool linked = false;
                for (int64_t s = R.Np[jq]; s < s0; ++s) if (R.Ni[s] == c) linked = true;
                const int64_t s1 = offdiag_begin(c);
                if (!linked || R.Np[c + 1] - s1 != len0 || !std::equal(R.Ni.begin() + s0, R.Ni.begin() + s0 + len0, R.Ni.begin() + s1)) break;
                ++w;
            }
            R.sn_col0.push_back(jq);
            for (int64_t s = s0; s < s0 + len0; ++s) R.sn_rows.push_back((int32_t)(R.Ni[s] - R.npts_cols));
            R.sn_rptr.push_back((int64_t)R.sn_rows.size());
            jq += w;
        }
        R.sn_col0.push_back(R.npts_cols);
        const int64_t nsn = (int64_t)R.sn_col0.size() - 1;
        R.LD.assign((size_t)9 * nsn, 0.0);
        R.LR.assign((size_t)3 * R.sn_rows.size(), 0.0);
        // groups of supernodes with the same row list
        {
            std::vector<int32_t> order((size_t)nsn);
            for (int64_t i = 0; i < nsn; ++i) order[i] = (int32_t)i;
            auto rows_of = [&](int32_t sidx) { return std::make_pair(&R.sn_rows[R.sn_rptr[sidx]], R.sn_rptr[sidx + 1] - R.sn_rptr[sidx]); };
            std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b2) {
                auto ra = rows_of(a), rb = rows_of(b2);
                if (ra.second != rb.second) return ra.second < rb.second;
                const int c = std::memcmp(ra.first, rb.first, (size_t)ra.second * sizeof(int32_t));
                return c != 0 ? c < 0 : a < b2;
            });
            R.grp_sn = order;
            R.grp_ptr.assign(1, 0);
            for (int64_t i = 1; i <= nsn; ++i) {
                bool same = i < nsn;
                if (same) {
                    auto ra = rows_of(order[i - 1]), rb = rows_of(order[i]);
                    same = ra.second == rb.second && std::memcmp(ra.first, rb.first, (size_t)ra.second * sizeof(int32_t)) == 0;
                    if (same && i - R.grp_ptr.back() >= 4096) same = false;      // bound the work of one task
                }
                if (!same) R.grp_ptr.push_back(i);
            }
        }
    }
    // ---- root supernode: dense lower triangle, tile-level symbolic factorisation
    R.T.assign((size_t)R.NS * R.NS, 0.0);
    R.nt = (int)((R.NS + NB - 1) / NB);
    R.tnz.assign((size_t)R.nt * R.nt, 0);
    for (int64_t jq = R.npts_cols; jq < R.n; ++jq)             // entries of N in the root
        for (int64_t s = R.Np[jq]; s < R.Np[jq + 1]; ++s)
            R.tnz[(size_t)((R.Ni[s] - R.npts_cols) / NB) * R.nt + (jq - R.npts_cols) / NB] = 1;
    {
        const int64_t nsn = (int64_t)R.sn_col0.size() - 1;     // fill from the leaves: all pairs of a leaf's rows
        for (int64_t s = 0; s < nsn; ++s) {
            int last = -1;
            std::vector<int> tl;
            for (int64_t k = R.sn_rptr[s]; k < R.sn_rptr[s + 1]; ++k) { const int t = R.sn_rows[k] / NB; if (t != last) { tl.push_back(t); last = t; } }
            for (size_t a = 0; a < tl.size(); ++a) for (size_t b2 = 0; b2 <= a; ++b2) R.tnz[(size_t)tl[a] * R.nt + tl[b2]] = 1;
        }
        for (int k = 0; k < R.nt; ++k) {
            R.tnz[(size_t)k * R.nt + k] = 1;
            for (int i = k + 1; i < R.nt; ++i) if (R.tnz[(size_t)i * R.nt + k])
                for (int j = k + 1; j <= i; ++j) if (R.tnz[(size_t)j * R.nt + k]) R.tnz[(size_t)i * R.nt + j] = 1;
        }
    }
    R.g.assign((size_t)R.n, 0.0); R.b.assign((size_t)R.n, 0.0); R.xsol.assign((size_t)R.n, 0.0);
    return true;
}

// JTJ = J'*J (levenberg_marquardt.m:81), lower triangle in elimination order, + lambda*I
static double spgemm_JtJ(Ref &R, double lambda) {
    double trace = 0;
#pragma omp parallel num_threads(R.nthreads) reduction(+ : trace)
    {
        std::vector<double> acc((size_t)R.n, 0.0);
#pragma omp for schedule(dynamic, 64)
        for (int64_t jq = 0; jq < R.n; ++jq) {
            const int64_t j = R.qinv[jq];
            for (int64_t s = R.Jp[j]; s < R.Jp[j + 1]; s += 2) {
                const int64_t o = R.Ji[s] >> 1;
                const double v0 = R.Jx[s], v1 = R.Jx[s + 1];
                const int64_t *col = &R.ocol[(size_t)o * MAXC], *slot = &R.oslot[(size_t)o * MAXC];
                for (int k = 0; k < R.ocnt[o]; ++k) {
                    const int64_t iq = R.q[col[k]];
                    if (iq >= jq) acc[iq] += v0 * R.Jx[slot[k]] + v1 * R.Jx[slot[k] + 1];
                }
            }
            for (int64_t e = R.pr_ptr[j]; e < R.pr_ptr[j + 1]; ++e) { const double w = R.pr_w[R.pr_rows[e]]; acc[jq] += w * w; }   // single-entry rows
            trace += acc[jq];
            acc[jq] += lambda;
            for (int64_t s = R.Np[jq]; s < R.Np[jq + 1]; ++s) { R.Nx[s] = acc[R.Ni[s]]; acc[R.Ni[s]] = 0.0; }
        }
    }
    return trace;
}

// C(m x n) -= A(m x k) * B(n x k)', column-major
static void gemm_nt(int m, int n, int k, const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc) {
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        double *c0 = C + (int64_t)j * ldc, *c1 = c0 + ldc, *c2 = c1 + ldc, *c3 = c2 + ldc;
        for (int l = 0; l < k; ++l) {
            const double *a = A + (int64_t)l * lda;
            const double b0 = B[j + (int64_t)l * ldb], b1 = B[j + 1 + (int64_t)l * ldb], b2 = B[j + 2 + (int64_t)l * ldb], b3 = B[j + 3 + (int64_t)l * ldb];
#pragma omp simd
            for (int i = 0; i < m; ++i) { const double ai = a[i]; c0[i] -= ai * b0; c1[i] -= ai * b1; c2[i] -= ai * b2; c3[i] -= ai * b3; }
        }
    }
    for (; j < n; ++j) {
        double *c0 = C + (int64_t)j * ldc;
        for (int l = 0; l < k; ++l) {
            const double *a = A + (int64_t)l * lda;
            const double b0 = B[j + (int64_t)l * ldb];
#pragma omp simd
            for (int i = 0; i < m; ++i) c0[i] -= a[i] * b0;
        }
    }
}

// dense Cholesky of the root supernode in place (lower), skipping empty tiles; returns false if not SPD
static bool root_cholesky(Ref &R) {
    const int64_t n = R.NS, ld = R.NS;
    const int nth = std::min(R.nthreads, 32);      // 47 panels of 128 columns at C3: more threads only wait at the barriers
    double *T = R.T.data();
    bool ok = true;
    for (int k = 0; k < R.nt && ok; ++k) {
        const int64_t k0 = (int64_t)k * NB, kb = std::min<int64_t>(NB, n - k0);
        double *D = T + k0 * ld + k0;
        for (int64_t j = 0; j < kb; ++j) {                      // unblocked potrf of the diagonal tile
            double d = D[j * ld + j];
            for (int64_t l = 0; l < j; ++l) d -= D[l * ld + j] * D[l * ld + j];
            if (!(d > 0)) { ok = false; break; }
            d = std::sqrt(d);
            D[j * ld + j] = d;
            for (int64_t i = j + 1; i < kb; ++i) {
                double s = D[j * ld + i];
                for (int64_t l = 0; l < j; ++l) s -= D[l * ld + i] * D[l * ld + j];
                D[j * ld + i] = s / d;
            }
        }
        if (!ok) break;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nth)
        for (int i = k + 1; i < R.nt; ++i) {                   // panel: X = A L^-T
            if (!R.tnz[(size_t)i * R.nt + k]) continue;
            const int64_t i0 = (int64_t)i * NB, ib = std::min<int64_t>(NB, n - i0);
            double *X = T + k0 * ld + i0;
            for (int64_t j = 0; j < kb; ++j) {
                double *xj = X + j * ld;
                for (int64_t l = 0; l < j; ++l) {
                    const double lj = D[l * ld + j];
                    const double *xl = X + l * ld;
#pragma omp simd
                    for (int64_t r = 0; r < ib; ++r) xj[r] -= xl[r] * lj;
                }
                const double inv = 1.0 / D[j * ld + j];
#pragma omp simd
                for (int64_t r = 0; r < ib; ++r) xj[r] *= inv;
            }
        }
        // trailing update, one task per (i, j) tile with both panels present
        std::vector<std::pair<int, int>> work;
        for (int i = k + 1; i < R.nt; ++i) if (R.tnz[(size_t)i * R.nt + k])
            for (int j = k + 1; j <= i; ++j) if (R.tnz[(size_t)j * R.nt + k]) work.emplace_back(i, j);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nth)
        for (int64_t t = 0; t < (int64_t)work.size(); ++t) {
            const int i = work[t].first, j = work[t].second;
            const int64_t i0 = (int64_t)i * NB, ib = std::min<int64_t>(NB, n - i0);
            const int64_t j0 = (int64_t)j * NB, jb = std::min<int64_t>(NB, n - j0);
            gemm_nt((int)ib, (int)jb, (int)kb, T + k0 * ld + i0, ld, T + k0 * ld + j0, ld, T + j0 * ld + i0, ld);
        }
    }
    return ok;
}

// supernodal LL' of N (elimination order); returns false if a pivot is not positive
static bool factorize(Ref &R, double *ms_leaves, double *ms_update, double *ms_root) {
    const int64_t nsn = (int64_t)R.sn_col0.size() - 1;
    bool ok = true;
    auto t0 = clk::now();
    // leaves: L_D = chol(D), L_R = R L_D^-T
#pragma omp parallel for schedule(static) num_threads(R.nthreads) reduction(&& : ok)
    for (int64_t s = 0; s < nsn; ++s) {
        const int64_t c0 = R.sn_col0[s];
        const int w = (int)(R.sn_col0[s + 1] - c0);
        const int64_t len = R.sn_rptr[s + 1] - R.sn_rptr[s];
        double D[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double *lr = &R.LR[(size_t)3 * R.sn_rptr[s]];
        for (int a = 0; a < w; ++a) {
            const int64_t c = c0 + a;
            int64_t sidx = R.Np[c];
            for (; sidx < R.Np[c + 1] && R.Ni[sidx] < R.npts_cols; ++sidx) D[3 * a + (R.Ni[sidx] - c0)] = R.Nx[sidx];
            for (int64_t k = 0; k < len; ++k) lr[3 * k + a] = R.Nx[sidx + k];
        }
        // 3x3 (w x w) Cholesky, column-major lower: D[3*col + row]
        for (int j = 0; j < w; ++j) {
            double d = D[3 * j + j];
            for (int l = 0; l < j; ++l) d -= D[3 * l + j] * D[3 * l + j];
            if (!(d > 0)) { ok = false; d = 1; }
            d = std::sqrt(d);
            D[3 * j + j] = d;
            for (int i = j + 1; i < w; ++i) {
                double v = D[3 * j + i];
                for (int l = 0; l < j; ++l) v -= D[3 * l + i] * D[3 * l + j];
                D[3 * j + i] = v / d;
            }
        }
        for (int64_t k = 0; k < len; ++k) {
            double *x = lr + 3 * k;
            for (int j = 0; j < w; ++j) {
                double v = x[j];
                for (int l = 0; l < j; ++l) v -= x[l] * D[3 * l + j];
                x[j] = v / D[3 * j + j];
            }
            for (int j = w; j < 3; ++j) x[j] = 0.0;
        }
        std::copy(D, D + 9, &R.LD[(size_t)9 * s]);
    }
    *ms_leaves = ms_since(t0);
    if (!ok) return false;
    t0 = clk::now();
    // root := N(root, root) - sum over leaves L_R L_R': one dense block per group of leaves with
    // the same rows, summed in cache and added to the root once
    const int64_t NS = R.NS;
#pragma omp parallel for schedule(static) num_threads(R.nthreads)
    for (int64_t j = 0; j < NS; ++j) {
        double *tc = &R.T[(size_t)j * NS];
        std::fill(tc + j, tc + NS, 0.0);
        const int64_t jq = R.npts_cols + j;
        for (int64_t s2 = R.Np[jq]; s2 < R.Np[jq + 1]; ++s2) tc[R.Ni[s2] - R.npts_cols] = R.Nx[s2];
    }
    const int64_t ngrp = (int64_t)R.grp_ptr.size() - 1;
#pragma omp parallel num_threads(R.nthreads)
    {
        std::vector<double> blk;
#pragma omp for schedule(dynamic, 4)
        for (int64_t gi = 0; gi < ngrp; ++gi) {
            const int32_t s0 = R.grp_sn[R.grp_ptr[gi]];
            const int64_t len = R.sn_rptr[s0 + 1] - R.sn_rptr[s0];
            const int32_t *rows = &R.sn_rows[R.sn_rptr[s0]];
            blk.assign((size_t)len * len, 0.0);                 // column-major, lower triangle used
            for (int64_t e = R.grp_ptr[gi]; e < R.grp_ptr[gi + 1]; ++e) {
                const double *lr = &R.LR[(size_t)3 * R.sn_rptr[R.grp_sn[e]]];
                for (int64_t c = 0; c < len; ++c) {
                    const double y0 = lr[3 * c], y1 = lr[3 * c + 1], y2 = lr[3 * c + 2];
                    double *bc = &blk[(size_t)c * len];
                    for (int64_t k = c; k < len; ++k) bc[k] += lr[3 * k] * y0 + lr[3 * k + 1] * y1 + lr[3 * k + 2] * y2;
                }
            }
            for (int64_t c = 0; c < len; ++c) {
                double *tc = &R.T[(size_t)rows[c] * NS];
                const double *bc = &blk[(size_t)c * len];
                for (int64_t k = c; k < len; ++k) {
#pragma omp atomic
                    tc[rows[k]] -= bc[k];
                }
            }
        }
    }
    *ms_update = ms_since(t0);
    t0 = clk::now();
    ok = root_cholesky(R);
    *ms_root = ms_since(t0);
    return ok;
}

// solve L L' x = b (elimination order), in place in v
static void solve(Ref &R, double *v) {
    const int64_t nsn = (int64_t)R.sn_col0.size() - 1, NS = R.NS, np_c = R.npts_cols;
    double *vt = v + np_c;
    // forward: leaves
    std::vector<double> priv((size_t)R.nthreads * NS, 0.0);
#pragma omp parallel num_threads(R.nthreads)
    {
        double *mine = &priv[(size_t)omp_get_thread_num() * NS];
#pragma omp for schedule(static)
        for (int64_t s = 0; s < nsn; ++s) {
            const int64_t c0 = R.sn_col0[s];
            const int w = (int)(R.sn_col0[s + 1] - c0);
            const double *D = &R.LD[(size_t)9 * s];
            double y[3] = {0, 0, 0};
            for (int j = 0; j < w; ++j) {
                double t = v[c0 + j];
                for (int l = 0; l < j; ++l) t -= D[3 * l + j] * y[l];
                y[j] = t / D[3 * j + j];
                v[c0 + j] = y[j];
            }
            const int64_t r0 = R.sn_rptr[s], len = R.sn_rptr[s + 1] - r0;
            const double *lr = &R.LR[(size_t)3 * r0];
            const int32_t *rows = &R.sn_rows[r0];
            for (int64_t k = 0; k < len; ++k) mine[rows[k]] += lr[3 * k] * y[0] + lr[3 * k + 1] * y[1] + lr[3 * k + 2] * y[2];
        }
    }
    for (int t = 0; t < R.nthreads; ++t)
        for (int64_t i = 0; i < NS; ++i) vt[i] -= priv[(size_t)t * NS + i];
    // root: forward and backward (column-oriented, contiguous columns)
    const double *T = R.T.data();
    for (int64_t j = 0; j < NS; ++j) {
        const double xj = vt[j] / T[j * NS + j];
        vt[j] = xj;
        const double *c = T + j * NS;
#pragma omp simd
        for (int64_t i = j + 1; i < NS; ++i) vt[i] -= c[i] * xj;
    }
    for (int64_t j = NS - 1; j >= 0; --j) {
        const double *c = T + j * NS;
        double s = 0;
#pragma omp simd reduction(+ : s)
        for (int64_t i = j + 1; i < NS; ++i) s += c[i] * vt[i];
        vt[j] = (vt[j] - s) / c[j];
    }
    // backward: leaves
#pragma omp parallel for schedule(static) num_threads(R.nthreads)
    for (int64_t s = 0; s < nsn; ++s) {
        const int64_t c0 = R.sn_col0[s];
        const int w = (int)(R.sn_col0[s + 1] - c0);
        const double *D = &R.LD[(size_t)9 * s];
        const int64_t r0 = R.sn_rptr[s], len = R.sn_rptr[s + 1] - r0;
        const double *lr = &R.LR[(size_t)3 * r0];
        const int32_t *rows = &R.sn_rows[r0];
        double y[3] = {v[c0], w > 1 ? v[c0 + 1] : 0.0, w > 2 ? v[c0 + 2] : 0.0};
        for (int64_t k = 0; k < len; ++k) {
            const double xt = vt[rows[k]];
            y[0] -= lr[3 * k] * xt; y[1] -= lr[3 * k + 1] * xt; y[2] -= lr[3 * k + 2] * xt;
        }
        for (int j = w - 1; j >= 0; --j) {
            double t = y[j];
            for (int i = j + 1; i < w; ++i) t -= D[3 * j + i] * v[c0 + i];
            v[c0 + j] = t / D[3 * j + j];
        }
    }
}

}  // namespace

extern "C" {

struct cpuref_handle { Ref R; };

const char *cpuref_last_error(const cpuref_handle *h) { return h ? h->R.err.c_str() : "null handle"; }

cpuref_handle *cpuref_create(const cpuref_problem *pb, int32_t nthreads, double *setup_ms) {
    if (!pb || pb->model < 2 || pb->model > 5 || pb->nIOrows != 5 + pb->nK + pb->nP || pb->nIOrows > MAXIO) return nullptr;
    auto h = std::make_unique<cpuref_handle>();
    Ref &R = h->R;
    R.nc = pb->nc; R.np = pb->np; R.no = pb->no; R.nIOrows = pb->nIOrows; R.model = pb->model; R.nK = pb->nK; R.nP = pb->nP;
    R.nthreads = nthreads > 0 ? nthreads : omp_get_max_threads();
    R.cam.assign(pb->cam, pb->cam + R.no); R.pt.assign(pb->pt, pb->pt + R.no);
    R.uv.assign(pb->uv, pb->uv + (size_t)2 * R.no); R.w.assign(pb->std, pb->std + (size_t)2 * R.no);
    R.px.assign(pb->px, pb->px + R.nc);
    R.IO.assign(pb->IO, pb->IO + (size_t)R.nIOrows * R.nc); R.EO.assign(pb->EO, pb->EO + (size_t)6 * R.nc);
    R.OP.assign(pb->OP, pb->OP + (size_t)3 * R.np);
    R.estIO.assign(pb->estIO, pb->estIO + (size_t)R.nIOrows * R.nc); R.estEO.assign(pb->estEO, pb->estEO + (size_t)6 * R.nc);
    R.estOP.assign(pb->estOP, pb->estOP + (size_t)3 * R.np);
    R.IOblock.assign(pb->IOblock, pb->IOblock + (size_t)R.nIOrows * R.nc);
    const auto t0 = clk::now();
    {   // prior observations of estimated parameters (bundle.m:137-154 drops the others); the x index of an entry is known
        // after the index maps exist, so they are built here first (setup() rebuilds the same maps)
        std::vector<int64_t> io, eo, op, src;
        const int64_t nIO = serial_indices(R.nIOrows, R.nc, R.estIO.data(), R.IOblock.data(), 0, io, &src);
        const int64_t nEO = serial_indices(6, R.nc, R.estEO.data(), nullptr, nIO, eo, nullptr);
        serial_indices(3, R.np, R.estOP.data(), nullptr, nIO + nEO, op, nullptr);
        auto collect = [&](const uint8_t *use, const double *val, const double *sd, const std::vector<int64_t> &ix) {
            if (!use || !val || !sd) return;
            for (size_t e = 0; e < ix.size(); ++e)
                if (use[e] && ix[e] >= 0) { R.pr_col.push_back(ix[e]); R.pr_w.push_back(1.0 / sd[e]); R.pr_val.push_back(val[e]); }
        };
        collect(pb->useIO, pb->priorIO, pb->stdIO, io);
        collect(pb->useEO, pb->priorEO, pb->stdEO, eo);
        collect(pb->useOP, pb->priorOP, pb->stdOP, op);
    }
    if (!setup(R)) return nullptr;
    if (setup_ms) *setup_ms = ms_since(t0);
    return h.release();
}

void cpuref_destroy(cpuref_handle *h) { delete h; }
int64_t cpuref_num_params(const cpuref_handle *h) { return h ? h->R.n : -1; }
int32_t cpuref_num_threads(const cpuref_handle *h) { return h ? h->R.nthreads : -1; }
int64_t cpuref_nnz(const cpuref_handle *h, int32_t which) {
    if (!h) return -1;
    const Ref &R = h->R;
    return which == 0 ? (int64_t)(R.Jx.size() + R.pr_col.size()) : which == 1 ? (int64_t)R.Nx.size() : (int64_t)R.LR.size() + R.NS * (R.NS + 1) / 2;
}

// serialize.m:14-18
int32_t cpuref_serialize(const cpuref_handle *h, double *x) {
    if (!h || !x) return -1;
    const Ref &R = h->R;
    for (size_t e = 0; e < R.IOix.size(); ++e) if (R.IOix[e] >= 0) x[R.IOix[e]] = R.IO[e];
    // shared IO entries: the leading one wins (IOsrc), written last
    for (int64_t k = 0; k < R.nIO; ++k) x[k] = R.IO[(size_t)R.IOsrc[k]];
    for (size_t e = 0; e < R.EOix.size(); ++e) if (R.EOix[e] >= 0) x[R.EOix[e]] = R.EO[e];
    for (size_t e = 0; e < R.OPix.size(); ++e) if (R.OPix[e] >= 0) x[R.OPix[e]] = R.OP[e];
    return 0;
}

// One Levenberg-Marquardt iteration as levenberg_marquardt.m:81-82,119,160-170:
//   [r,J] at x ; JTJ = J'*J ; JTr = J'*r ; p = (JTJ + lambda*I) \ (-JTr) ; f(x+p).
// lambda < 0: |lambda| * trace(JTJ)/n  (levenberg_marquardt.m:88-90).
// stats: [0] f=0.5 r'r  [1] f(x+p)  [2] trace(JTJ)  [3] lambda used
//        [4..10] ms: residual+Jacobian, J'J, J'r, leaves, root update, root Cholesky, solve ; [11] ms trial residual
// Returns 0, or -2 if the matrix is not positive definite.
int32_t cpuref_lm_step(cpuref_handle *h, const double *x, double lambda, double *p, double *stats) {
    if (!h || !x) return -1;
    Ref &R = h->R;
    auto t0 = clk::now();
    const double f = residual_jacobian(R, x, true, R.r);
    const double ms_rj = ms_since(t0);
    t0 = clk::now();
    double trace = spgemm_JtJ(R, 0.0);
    if (lambda < 0) lambda = std::fabs(lambda) * trace / (double)R.n;
    if (lambda != 0) {
#pragma omp parallel for schedule(static) num_threads(R.nthreads)
        for (int64_t jq = 0; jq < R.n; ++jq) R.Nx[R.Np[jq]] += lambda;      // the diagonal is the first row of a column
    }
    const double ms_jtj = ms_since(t0);
    t0 = clk::now();
#pragma omp parallel for schedule(dynamic, 256) num_threads(R.nthreads)
    for (int64_t j = 0; j < R.n; ++j) {
        double s = 0;
        for (int64_t k = R.Jp[j]; k < R.Jp[j + 1]; ++k) s += R.Jx[k] * R.r[R.Ji[k]];
        for (int64_t e = R.pr_ptr[j]; e < R.pr_ptr[j + 1]; ++e) s += R.pr_w[R.pr_rows[e]] * R.r[(size_t)2 * R.no + R.pr_rows[e]];
        R.b[R.q[j]] = -s;
    }
    const double ms_jtr = ms_since(t0);
    double ms_leaves = 0, ms_update = 0, ms_root = 0;
    const bool ok = factorize(R, &ms_leaves, &ms_update, &ms_root);
    t0 = clk::now();
    if (ok) solve(R, R.b.data());
    const double ms_solve = ms_since(t0);
    std::vector<double> xt((size_t)R.n);
    for (int64_t j = 0; j < R.n; ++j) { const double pj = ok ? R.b[R.q[j]] : 0.0; if (p) p[j] = pj; xt[j] = x[j] + pj; }
    t0 = clk::now();
    std::vector<double> rt;
    const double ft = residual_jacobian(R, xt.data(), false, rt);
    const double ms_res = ms_since(t0);
    if (stats) {
        stats[0] = f; stats[1] = ft; stats[2] = trace; stats[3] = lambda;
        stats[4] = ms_rj; stats[5] = ms_jtj; stats[6] = ms_jtr; stats[7] = ms_leaves; stats[8] = ms_update;
        stats[9] = ms_root; stats[10] = ms_solve; stats[11] = ms_res;
    }
    return ok ? 0 : -2;
}

// |J v|^2, r'J v and v'v with the explicit Jacobian and residual of the last cpuref_lm_step (the
// termination test's norm(J*p), levenberg_marquardt.m:162; the slope r'Jp of the line search).  Not part
// of the timed step: the parity tests compare the GPU's scalars of one linearise+solve with it.
int32_t cpuref_step_norms(const cpuref_handle *h, const double *v, double *out3) {
    if (!h || !v || !out3) return -1;
    const Ref &R = h->R;
    std::vector<double> Jv((size_t)2 * R.no + R.pr_col.size(), 0.0);
    for (size_t k = 0; k < R.pr_col.size(); ++k) Jv[(size_t)2 * R.no + k] = R.pr_w[k] * v[R.pr_col[k]];
    double vv = 0;
    for (int64_t j = 0; j < R.n; ++j) {              // column sweep (serial: columns share rows)
        vv += v[j] * v[j];
        for (int64_t k = R.Jp[j]; k < R.Jp[j + 1]; ++k) Jv[R.Ji[k]] += R.Jx[k] * v[j];
    }
    double a = 0, b = 0;
    for (size_t i = 0; i < Jv.size(); ++i) { a += Jv[i] * Jv[i]; b += Jv[i] * R.r[i]; }
    out3[0] = a; out3[1] = b; out3[2] = vv;
    return 0;
}

}  // extern "C"
