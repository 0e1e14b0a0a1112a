# registers / LDS / scratch of the kernels in a built library: bench/kernel_regs.sh [lib] [name filter]
L=${1:-dbat_amd/libdbat_hip.so}; T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $L && /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co && /opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | grep -E "\.name:|vgpr_count|agpr_count|vgpr_spill|private_segment_fixed|group_segment_fixed" | paste - - - - - - | sed "s/ \+/ /g" | grep -E "${2:-.}" | cut -c1-230
rm -rf $T
