#!/bin/bash
# disassemble the gfx950 code object of a library build: tools/isa_dump.sh [lib] [out.s]
LIB=${1:-uneven_planner_amd/libunevenhip.so}; OUT=${2:-build/isa/all.s}
mkdir -p $(dirname $OUT); TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $LIB || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$TMP/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/dev.co 2>/dev/null || exit 1
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $TMP/dev.co > $OUT 2>/dev/null
rm -rf $TMP; wc -l $OUT
