#!/bin/bash
# Builds a timing / ablation variant of the library as sarpro_amd/lib_<name>.so: kernels.hip recompiled with extra flags, the
# other objects as they stand.  usage: tools/build_variant.sh <name> "<flags>" [file.hip]   (A/B them with tools/time_variants.py)
set -e
cd "$(dirname "$0")/../sarpro_amd/csrc"
name=$1; flags=$2; src=${3:-kernels.hip}; obj=${src%.hip}.o
make -s -j8
OTHERS=$(ls *.o | grep -v "^$obj$")
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function $flags -c $src -o /tmp/variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$name.so $OTHERS /tmp/variant_$name.o -ldl -lpthread
echo built sarpro_amd/lib_$name.so
