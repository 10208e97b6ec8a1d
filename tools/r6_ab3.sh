#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6d; mkdir -p $O
for rep in 1 2 3; do
for l in lib_wgtimes.so lib_pipe_wgtimes.so; do
  echo "== $l"; SARPRO_HIP_LIB=$PWD/sarpro_amd/$l timeout 300 python tools/rgb_wg_times.py 2>&1 | tail -2
done; done | tee $O/wg_ab.txt
N=40 python tools/time_variants.py - lib_pipe.so - lib_pipe.so 2>&1 | cut -c1-200 | tee $O/ab_pipe.txt
