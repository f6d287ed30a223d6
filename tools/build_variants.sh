#!/bin/bash
# Builds named variants of libunevenhip.so for A/B measurements on the GPU box: build/variants/libunevenhip_<name>.so
# usage: tools/build_variants.sh name1="-DFLAG=1 ..." name2="..."      (run in the repo root; select with UNEVENHIP_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
SRC=uneven_planner_amd/csrc
pids=()
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $flags -shared $SRC/unevenhip.hip $SRC/map_build.hip $SRC/kino_search.hip $SRC/resample_host.cpp $SRC/map_io_host.cpp -ldl -lpthread -o build/variants/libunevenhip_$name.so && echo "built $name ($flags)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
