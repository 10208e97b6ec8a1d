#!/bin/bash
# Builds timing variants of the library (ablations of the fused pass; their rasters are garbage) as sarpro_amd/lib_<name>.so
# usage: tools/build_variants.sh NAME:-DFLAG[,-DFLAG2] ...
set -e
cd "$(dirname "$0")/../sarpro_amd/csrc"
OTHERS="kernels.o f32_kernels.o chain_kernels.o resize_kernels.o resize_path.o batch.o api.o f32_path.o comm.o host_logic.o tiff_io.o"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"; flags="${flags//,/ }"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wno-pass-failed $flags -c fused_kernels.hip -o /tmp/fused_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$name.so $OTHERS /tmp/fused_$name.o -ldl -lpthread
done
