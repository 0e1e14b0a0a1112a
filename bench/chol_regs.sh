# k_chol_df alone in seven seconds: registers / scratch of both instantiations (the library takes three minutes).
#   bash bench/chol_regs.sh            (prints VGPRs / AGPRs / scratch bytes per lane; -S output in /tmp/chol_regs/t.s)
R=$(cd $(dirname $0)/.. && pwd); T=/tmp/chol_regs; rm -rf $T; mkdir -p $T/a/b; cp -r $R/dbat_amd/csrc $T/a/b/csrc; cp -r $R/include $T/a/include
cat > $T/a/b/csrc/t.hip <<'EOF'
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "../../include/dbat_hip.h"
#include "chol_df.hpp"
#define DF_ARGS dbat::DfView, int, int, const dbat::DfJob *, int, int *, int *, int, double *, int *, long long *, const int *, const int *, double *, double *, double *, const double *, double *, double *, int, const int *, int, int, dbat::DfChain
template __global__ void dbat::k_chol_df<false>(DF_ARGS);
template __global__ void dbat::k_chol_df<true>(DF_ARGS);
EOF
cd $T/a/b/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only -S t.hip -o $T/t.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 "Function Name: _ZN4dbat9k_chol_df" | grep -E "Name|VGPRs:|AGPRs:|Scratch" | sed 's/.*remark: *//; s/ \[-R.*//; s/Function Name: _ZN4dbat9k_chol_dfILb\(.\).*/k_chol_df<\1>/' | tr '\n' ' '; echo
grep -c "scratch_" $T/t.s | sed 's/^/scratch instructions: /'
