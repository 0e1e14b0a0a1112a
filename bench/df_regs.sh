# k_chol_df<false> / <true> alone in half a minute: registers / scratch / LDS (the library takes three minutes).
# EXTRA="-DDBAT_DF_OCC2=1" etc. are passed to the compiler.
R=$(cd $(dirname $0)/.. && pwd); T=/tmp/df_regs; rm -rf $T; mkdir -p $T/a/b; cp -r $R/dbat_amd/csrc $T/a/b/csrc; cp -r $R/include $T/a/include
cat > $T/a/b/csrc/t.hip <<'EOF'
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "../../include/dbat_hip.h"
#include "kernels.hpp"
#include "chol_df.hpp"
#define DF_ARGS dbat::DfView, int, int, const dbat::DfJob *, int, int *, int *, int, double *, int *, long long *, const int *, const int *, double *, double *, double *, const double *, double *, double *, int, const int *, int, int, dbat::DfChain
template __global__ void dbat::k_chol_df<false>(DF_ARGS);
template __global__ void dbat::k_chol_df<true>(DF_ARGS);
#ifdef DF_EXTRA_INST
DF_EXTRA_INST
#endif
EOF
cd $T/a/b/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only $EXTRA -S t.hip -o $T/t.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A10 "Function Name: _ZN4dbat9k_chol_df" | grep -E "Name|VGPRs:|AGPRs:|Scratch|LDS|Occupancy|error" | sed 's/.*remark: *//; s/ \[-R.*//'
