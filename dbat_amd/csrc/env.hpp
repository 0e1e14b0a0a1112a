// Environment switches of libdbat_hip.so -- the complete list.  Two classes:
//   product      selects a path that some class of scenes needs (and that a test forces), or prints statistics;
//                results are those of the reference whatever the value
//   measurement  ablation bits, per-task clocks, schedule overrides: compiled in only with -DDBAT_HIP_PROFILING
//                (make prof -> libdbat_hip_prof.so); some of them make the results WRONG by design
// A DBAT_HIP_* variable that is not in the table -- a typo, a knob of an older round -- or a measurement switch
// given to the product build makes dbat_hip_create / dbat_hip_plan* fail loudly (DBAT_HIP_EINVAL) instead of
// being ignored: a stray variable must not change or silently not change a production solve.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>

extern char **environ;

namespace dbat {

#ifdef DBAT_HIP_PROFILING
constexpr bool kProfilingBuild = true;
#else
constexpr bool kProfilingBuild = false;
#endif

// values: nullptr = any value; otherwise the '|'-separated list of the values the library honours (anything else is refused)
struct EnvSwitch { const char *name; bool measurement; const char *what; const char *values = nullptr; };
static const EnvSwitch kEnvSwitches[] = {
    {"DBAT_HIP_SIG", false, "0: signature kernels off, 2: on wherever a chunk's rows fit (default: by group length)", "0|1|2"},
    {"DBAT_HIP_SIG_IOS_OFF", false, "self-calibration: IO rows by LDS atomics in every tile"},
    {"DBAT_HIP_CMAX", false, "cameras per tile (0: column-list kernel only)"},
    {"DBAT_HIP_BT", false, "observations per batch (128 | 256)", "128|256"},
    {"DBAT_HIP_TILE_BMIN", false, "fewest batches a tile may be capped at (default 2)"},
    {"DBAT_HIP_HEAVY", false, "0: heavy / giant points by the column-list kernels (pair terms by global atomics) instead of the matrix-core path", "0|1"},
    {"DBAT_HIP_HEAVY_KS", false, "heavy / giant points: k-steps per task of k_heavy_syrk at most (default: by the amount of work, 12 ... 96)"},
    {"DBAT_HIP_GIANT_THREADS", false, "threads of the giant-point kernels (64 | 128 | 256)", "64|128|256"},
    {"DBAT_HIP_MG_REPLICATED", false, "several ranks: envelope summed, replicated factorisation"},
    {"DBAT_HIP_ND_OFF", false, "no nested dissection"},
    {"DBAT_HIP_ND_LEAF", false, "leaf size of the dissection"},
    {"DBAT_HIP_ND_PAD_ALL", false, "every block of the dissection on a tile boundary"},
    {"DBAT_HIP_ND_JOIN_SMALL", false, "separators up to that many rows join their child's last tile"},
    {"DBAT_HIP_DF_SPLIT", false, "factorisation: helper tasks for sums longer than this many products (default 96)"},
    {"DBAT_HIP_DF_CHUNK", false, "factorisation: products per helper task (default 32)"},
    {"DBAT_HIP_DF_CHAIN", false, "factorisation: 0 = diagonal tiles as ordinary tasks (no chain role), 1 = chain role whatever the pattern", "0|1"},
    {"DBAT_HIP_DF_CHAIN_WG", false, "factorisation: workgroups of the chain role at most (default 32)"},
    {"DBAT_HIP_DF_L2", false, "factorisation: 1 = finished tiles of the compact layout are read through the L2", "0|1"},
    {"DBAT_HIP_COV_DENSE", false, "posterior covariance: the dense inverse of the reduced system (rocsolver_dpotri) instead of the selected inversion"},
    {"DBAT_HIP_SPRANK_OFF", false, "structural rank from the counting conditions only"},
    {"DBAT_HIP_PLAN_THREADS", false, "threads of the host plan (default: hardware concurrency, at most 32)"},
    {"DBAT_HIP_PLAN_GRAIN", false, "elements per thread below which a pass of the host plan is not split (tests: 1)"},
    {"DBAT_HIP_PLAN_STATS", false, "1: print layout statistics, 2: also the wall time of every section of the plan"},
    {"DBAT_HIP_PIVOT_STATS", false, "print the pivot extremes and the rcond estimate of every solve"},
    {"DBAT_HIP_ABLATE", true, "k_build_sig / tile kernels: switch phases off (results are wrong), phase clocks"},
    {"DBAT_HIP_DF_TRACE", true, "per-task clocks of the factorisation, written to this file"},
    {"DBAT_HIP_DF_TRACE_POTF2", true, "with DF_TRACE: clocks inside the chain role's factorisations as well"},
    {"DBAT_HIP_DF_ABLATE", true, "factorisation: operand tiles of the products not fetched (1: L(i,j), 2: both, 3: and no flag looks) -- timing only, wrong results", "0|1|2|3"},
    {"DBAT_HIP_DF_ORDER", true, "one of the candidate task orders instead of the simulated best"},
    {"DBAT_HIP_DF_GRID", true, "workgroups of the factorisation"},
    {"DBAT_HIP_GRID_OBS", true, "launch size of the observation-parallel kernels"},
};

// nullptr when the variable is unset -- or is a measurement switch and this is the product build (env_validate has
// refused such an environment before any value is read)
inline const char *env_get(const char *name) {
    for (const EnvSwitch &e : kEnvSwitches)
        if (!strcmp(e.name, name)) return (e.measurement && !kProfilingBuild) ? nullptr : getenv(name);
    return nullptr;                                  // not in the table: a programming error, never a silent getenv
}
inline int env_int(const char *name, int dflt) { const char *e = env_get(name); return e ? atoi(e) : dflt; }
inline bool env_on(const char *name) { return env_get(name) != nullptr; }

inline bool env_validate(std::string &err) {
    for (char **p = environ; p && *p; ++p) {
        if (strncmp(*p, "DBAT_HIP_", 9) != 0) continue;
        const char *eq = strchr(*p, '=');
        const std::string name(*p, eq ? (size_t)(eq - *p) : strlen(*p));
        const EnvSwitch *hit = nullptr;
        for (const EnvSwitch &e : kEnvSwitches) if (name == e.name) hit = &e;
        if (!hit) { err = "unknown environment variable " + name + " (the switches of libdbat_hip.so: csrc/env.hpp, DESIGN.md 9)"; return false; }
        if (hit->measurement && !kProfilingBuild) {
            err = name + " is a measurement switch (" + hit->what + "): this library was built without -DDBAT_HIP_PROFILING (make -C dbat_amd/csrc prof)";
            return false;
        }
        if (hit->values && eq) {                         // a value the library would silently ignore is as bad as a typo in the name
            const std::string val(eq + 1), list = std::string("|") + hit->values + "|";
            if (val.empty() || val.find('|') != std::string::npos || list.find("|" + val + "|") == std::string::npos) {
                err = name + "=" + val + ": not one of " + hit->values + " (" + hit->what + ")";
                return false;
            }
        }
    }
    return true;
}

}  // namespace dbat
