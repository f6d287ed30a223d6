#!/bin/bash
# registers, scratch (spill) bytes and LDS of every kernel in the built library, from the code-object metadata
# usage: tools/kernel_resources.sh [path/to/libunevenhip.so]
LIB=${1:-uneven_planner_amd/libunevenhip.so}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $LIB || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$TMP/fat.bin 2>/dev/null | grep gfx950 | head -1 > $TMP/target
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$TMP/fat.bin --targets=$(cat $TMP/target) --output=$TMP/dev.co 2>/dev/null || { echo "no gfx950 code object in $LIB"; exit 1; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co | awk '
/\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {l=$2}
/\.vgpr_spill_count:/ {vs=$2} /\.sgpr_spill_count:/ {ss=$2} /\.wavefront_size:/ {printf "%-90s vgpr %3s sgpr %3s scratch %5s B lds %6s B spills v %s s %s\n", name, v, s, p, l, vs, ss}'
rm -rf $TMP
