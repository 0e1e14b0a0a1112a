// Threads of the host plan (plan.hpp): contiguous ranges over std::thread, vectors that are not zero-filled on
// resize (the plan's per-observation arrays are written once, in parallel: a zero fill by one thread would cost
// more than the pass that fills them), a parallel sort of totally ordered records.
//
// Every use keeps the RESULT independent of the number of threads: ranges only decide who writes an element, never
// what is written; reductions are exact (integer counts, min / max, bit-or) or done by one thread.
#pragma once
#include <algorithm>
#include <cstdint>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include "env.hpp"

namespace dbat {

// allocator whose construct() default-initialises: resize() of a vector of doubles does not touch the memory
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using uvec = std::vector<T, NoInitAlloc<T>>;

// Jobs 1 .. k-1 on threads of their own, job 0 on the caller.  A thread that cannot be created (std::system_error: a
// pids limit of the container) leaves its job to the caller; an exception inside a job (bad_alloc of a per-thread
// vector) is carried over the joins and rethrown -- a destroyed joinable std::thread or an exception escaping a thread
// would be std::terminate where dbat_hip_create should return DBAT_HIP_ENOMEM.
template <class Job>
inline void par_jobs(int k, Job &&job) {
    std::vector<std::thread> th;
    std::vector<int> inline_jobs;
    std::exception_ptr first;
    std::mutex mu;
    auto guarded = [&](int i) {
        try { job(i); }
        catch (...) { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); }
    };
    th.reserve(k > 1 ? k - 1 : 0);
    for (int i = 1; i < k; ++i) {
        try { th.emplace_back(guarded, i); }
        catch (const std::system_error &) { inline_jobs.push_back(i); }
    }
    guarded(0);
    for (int i : inline_jobs) guarded(i);
    for (auto &x : th) x.join();
    if (first) std::rethrow_exception(first);
}

struct Par {
    int nt = 1;
    // DBAT_HIP_PLAN_GRAIN=n: cut ranges down to n elements per thread (default: each pass has its own minimum, a few
    // thousand) -- the sanitizer and determinism tests use it to run the threaded paths on tiny scenes
    static int64_t grain(int64_t dflt) { const int g = env_int("DBAT_HIP_PLAN_GRAIN", 0); return g > 0 ? g : dflt; }
    static int default_threads() {
        const int e = env_int("DBAT_HIP_PLAN_THREADS", 0);
        if (e > 0) return std::min(e, 256);
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(std::max<unsigned>(hw, 1), 32);
    }
    // f(lo, hi, tid) over [0, n) cut into nt contiguous ranges (tid = index of the range)
    template <class F>
    void run(int64_t n, F &&f, int64_t min_per_thread = 4096) const {
        const int k = ranges(n, min_per_thread);
        if (k <= 1) { f((int64_t)0, n, 0); return; }
        par_jobs(k, [&f, n, k](int t) { f(n * t / k, n * (t + 1) / k, t); });
    }
    int ranges(int64_t n, int64_t min_per_thread = 4096) const {
        return (int)std::max<int64_t>(1, std::min<int64_t>(nt, n / std::max<int64_t>(grain(min_per_thread), 1)));
    }
    template <class V, class T>
    void fill(V &v, const T &x) const {
        run((int64_t)v.size(), [&](int64_t lo, int64_t hi, int) { std::fill(v.begin() + lo, v.begin() + hi, x); }, 1 << 16);
    }
    // sort under a strict total order (no two records compare equal): the result is the unique sorted sequence,
    // whatever the number of threads.  Pieces are sorted independently and merged pairwise, level by level.
    template <class R, class Less>
    void sort(std::vector<R> &v, Less less) const {
        const int64_t n = (int64_t)v.size();
        int k = 1;
        while (2 * k <= nt && n / (2 * k) >= grain(1 << 14)) k *= 2;
        if (k == 1) { std::sort(v.begin(), v.end(), less); return; }
        std::vector<int64_t> cut(k + 1);
        for (int i = 0; i <= k; ++i) cut[i] = n * i / k;
        par_jobs(k, [&](int i) { std::sort(v.begin() + cut[i], v.begin() + cut[i + 1], less); });
        std::vector<R> tmp(v.size());
        std::vector<R> *src = &v, *dst = &tmp;
        for (int w = 1; w < k; w *= 2) {
            const int nm = (k + 2 * w - 1) / (2 * w);          // merges of this level
            par_jobs(nm, [&, w](int q) {
                const int i = 2 * w * q;
                const int64_t a = cut[i], b = cut[std::min(i + w, k)], c = cut[std::min(i + 2 * w, k)];
                std::merge(src->begin() + a, src->begin() + b, src->begin() + b, src->begin() + c, dst->begin() + a, less);
            });
            std::swap(src, dst);
        }
        if (src != &v) v.swap(tmp);
    }
};

}  // namespace dbat
