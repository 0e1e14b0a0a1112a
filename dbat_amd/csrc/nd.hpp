// Nested dissection of the camera co-visibility graph (host only).
//
// One ordering serves two purposes:
//   * the elimination order of the reduced camera system in the dataflow Cholesky
//     (chol_df.hpp): independent parts of the camera network factor concurrently, only the
//     separators form a dependent chain;
//   * multi-GPU: the first levels of the same tree cut the network into one DOMAIN per rank
//     and a few TOP separators.  A separator is the set of cameras of one half that see a
//     camera of the other half, so an object point never sees interior cameras of two
//     domains (two such cameras would be co-visible and one of them in the separator):
//     every point belongs to the one domain whose interior cameras it sees, its block of
//     the reduced system touches only that domain's rows and the top separators' rows, and
//     a rank can build AND factor its domain without a word from the others.  Only the
//     Schur complement on the top separators is summed over the ranks.
//
// No counterpart in the reference (MATLAB's `\` orders and factors J'J internally,
// gauss_newton_armijo.m:172); the closest structure is the permuted block factor
// [OP; EO; IO] of bundle_cov.m:82-99.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <vector>

namespace dbat {

struct NdTree {
    int nparts = 1;
    std::vector<int> order;         // position -> camera
    std::vector<int> block_end;     // order.size() after every block (leaf or separator), in elimination order
    std::vector<int> block_owner;   // per block: the rank whose domain it belongs to, -1: a top separator
    std::vector<uint8_t> block_sep; // per block: a separator -- it directly follows the last block of the part it was cut from
    std::vector<int> cam_owner;     // per camera: rank, -1: top separator
    int n_top_cams = 0;
};

// adj: symmetric co-visibility bitsets (adj_words 64-bit words per camera); xyz: 3 coordinates per
// camera (projection centres) for the geometric bisection; weight (may be null): observations per
// camera, balances the domains; nparts: ranks; leaf: cameras of a block that is not cut further.
inline void nd_build(int nc, const uint64_t *adj, int adj_words, const double *xyz, const double *weight,
                     int nparts, int leaf, bool nd_off, NdTree &T) {
    T = NdTree();
    T.nparts = std::max(1, nparts);
    T.order.reserve(nc);
    T.cam_owner.assign(nc, 0);
    auto adjacent = [&](int a, int b) { return (adj[(size_t)a * adj_words + (b >> 6)] >> (b & 63)) & 1ull; };
    auto emit = [&](std::vector<int> &cams, int owner, bool sep = false) {
        std::sort(cams.begin(), cams.end());
        for (int c : cams) { T.order.push_back(c); T.cam_owner[c] = owner; if (owner < 0) ++T.n_top_cams; }
        if (!cams.empty()) { T.block_end.push_back((int)T.order.size()); T.block_owner.push_back(owner); T.block_sep.push_back(sep ? 1 : 0); }
    };
    // part0 .. part0+np-1: the ranks this set of cameras is divided among
    std::function<void(std::vector<int> &, int, int)> nd = [&](std::vector<int> &cams, int part0, int np) {
        if (cams.empty()) return;
        if (np == 1 && (int)cams.size() <= leaf) { emit(cams, part0); return; }
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int c : cams) for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], xyz[3 * c + d]); hi[d] = std::max(hi[d], xyz[3 * c + d]); }
        int ax = 0;
        for (int d = 1; d < 3; ++d) if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
        std::vector<int> s(cams);
        std::stable_sort(s.begin(), s.end(), [&](int a, int b) { return xyz[3 * a + ax] < xyz[3 * b + ax]; });
        size_t half = s.size() / 2;
        const int npa = np / 2, npb = np - npa;
        std::vector<int> A, B, Sep, A2;
        auto cut = [&](size_t h) {                      // lower part A = s[0, h), separator = the cameras of A that see B
            A.assign(s.begin(), s.begin() + h); B.assign(s.begin() + h, s.end()); Sep.clear(); A2.clear();
            for (int a : A) {
                bool touch = false;
                for (int b : B) if (adjacent(a, b)) { touch = true; break; }
                (touch ? Sep : A2).push_back(a);
            }
        };
        if (np > 1) {
            // The cut between the ranks.  The separator comes out of the lower part, so the cut is moved until
            // what REMAINS of it carries npa / np of the two interiors' weight (observations): bisection on the
            // position (the interior of A grows with it).
            auto wsum = [&](const std::vector<int> &v) { double t = 0; for (int c : v) t += weight ? weight[c] : 1.0; return t; };
            size_t lo = 1, hi = s.size() - 1;
            half = std::min(std::max<size_t>(s.size() * npa / np, lo), hi);
            for (int it = 0; it < 16 && lo < hi; ++it) {
                cut(half);
                const double wa = wsum(A2) / npa, wb = wsum(B) / npb;
                if (wa < wb) lo = half + 1; else hi = half;
                const size_t next = (lo + hi) / 2;
                if (next == half) break;
                half = next;
            }
            half = std::min(std::max<size_t>(half, 1), s.size() - 1);
        }
        cut(half);
        if (Sep.size() * 2 >= cams.size() || A2.empty()) {     // no useful separator: one dense block
            emit(cams, np > 1 ? -1 : part0);                   // (between ranks: the whole set is shared)
            return;
        }
        nd(A2, part0, np > 1 ? npa : 1);
        nd(B, np > 1 ? part0 + npa : part0, np > 1 ? npb : 1);
        emit(Sep, np > 1 ? -1 : part0, true);
    };
    std::vector<int> all(nc);
    for (int c = 0; c < nc; ++c) all[c] = c;
    if (nd_off) emit(all, T.nparts > 1 ? -1 : 0);
    else nd(all, 0, T.nparts);
}

}  // namespace dbat
