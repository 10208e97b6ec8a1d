#!/bin/bash
# Builds sarpro_amd/lib_co768.so: the co-residency experiment of round 5 -- the fused CLAHE -> RGB pass as
# 768-thread workgroups WITHOUT its 65-KB blue table (SARPRO_RGB_LITE: blue = rne(Pv[level1] * Qv[level2])) and the piece histogram as
# 512-thread workgroups with 4096 LDS bins, so that one workgroup of each fits on a compute unit (93 + 54 KB of LDS, 12 + 8 waves).
# Measure with:  SARPRO_HIP_LIB=$PWD/sarpro_amd/lib_co768.so CONFIGS=2:0:0:0,2:2:0:0 python tools/pipe_sweep.py   (order 2 pairs scene
# i + 1's histogram pass with scene i's fused pass).
set -e
cd "$(dirname "$0")/../sarpro_amd/csrc"
make -s -j8
F="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -DSARPRO_RGB_LITE -DSARPRO_RGB_BLOCK=768 -DSARPRO_RGB_POOL=3008 -DSARPRO_PIECE_BLOCK=512 -DSARPRO_PIECE_BINS=4096 -DSARPRO_PIECE_LOWBINS=32 -DSARPRO_PIECE_AHEAD=3"
/opt/rocm/bin/hipcc $F -c piece_kernels.hip -o /tmp/co_piece.o
/opt/rocm/bin/hipcc $F -c kernels.hip -o /tmp/co_kernels.o
/opt/rocm/bin/hipcc $F -x hip -c api.cpp -o /tmp/co_api.o
OTHERS=$(ls *.o | grep -v "^piece_kernels.o$" | grep -v "^api.o$" | grep -v "^kernels.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_co768.so $OTHERS /tmp/co_piece.o /tmp/co_kernels.o /tmp/co_api.o -ldl -lpthread
echo built sarpro_amd/lib_co768.so
