#!/bin/bash
# device-only build of ONE instantiation of uph_solver_kernel: registers, spills, scratch and the ISA in ~15 s
# usage: [TAG=name] tools/one_kernel.sh <NT> <WPS> <MODE> [extra flags...]      -> build/isa/k_<NT>_<WPS>_<MODE>[_tag].s
NT=$1; WPS=$2; MODE=$3; shift 3
cd "$(dirname "$0")/.."
mkdir -p build/isa
OUT=build/isa/k_${NT}_${WPS}_${MODE}${TAG:+_$TAG}.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-device-only -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -DUPH_ONE_KERNEL=1 -DUPH_OK_NT=$NT -DUPH_OK_WPS=$WPS -DUPH_OK_MODE=$MODE "$@" \
  -x hip uneven_planner_amd/csrc/unevenhip.hip -S -o $OUT 2>&1 | grep -v "hip-link" 
awk '/^_Z17uph_solver_kernel/ {on=1} on && /\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):/ {printf "%s %s  ", $1, $2} END {print ""}' $OUT | sed 's/^/'"$(basename $OUT)"': /'
