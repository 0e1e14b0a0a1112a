# k_build_sig<3,4,6,false> and <3,5,14,false> alone in eight seconds: registers / scratch (the library takes three minutes).
R=$(cd $(dirname $0)/.. && pwd); T=/tmp/sig_regs; rm -rf $T; mkdir -p $T/a/b; cp -r $R/dbat_amd/csrc $T/a/b/csrc; cp -r $R/include $T/a/include
cat > $T/a/b/csrc/t.hip <<'EOF'
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "../../include/dbat_hip.h"
#include "kernels.hpp"
#include "sig.hpp"
#define SIG_ARGS dbat::DevProblem, const double *, const dbat::CamRec *, double, int, double *, double *, double *, double *, double *, double *, unsigned long long *, const int32_t *, const int32_t *, const uint8_t *, const double *, const double *, unsigned *
template __global__ void dbat::k_build_sig<3, 4, 6, false>(SIG_ARGS);
template __global__ void dbat::k_build_sig<3, 5, 14, false>(SIG_ARGS);
EOF
cd $T/a/b/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only -S t.hip -o $T/t.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "Function Name: _ZN4dbat11k_build_sig" | grep -E "Name|VGPRs:|AGPRs:|Scratch|error" | sed 's/.*remark: *//; s/ \[-R.*//; s/Function Name: _ZN4dbat11k_build_sigILi3ELi\(.\)ELi\([0-9]*\).*/k_build_sig<3,\1,\2,false>/' | tr '\n' ' '; echo
