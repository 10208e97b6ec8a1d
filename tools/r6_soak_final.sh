#!/bin/bash
# soaks at the committed tree: random rasters through every route (sizes where the fused route is taken by attribute), the GRD-like rasters
# of 36 MP and more through the default route against the exact kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s; mkdir -p $O
timeout 1200 python tools/soak_random_rasters.py 250 12 31 > $O/soak_random_rasters.txt 2>&1; echo "exit $?"; tail -3 $O/soak_random_rasters.txt | cut -c1-400
timeout 1200 python tools/soak_grd_like.py 100 1 8 > $O/soak_grd_like.txt 2>&1; echo "exit $?"; tail -3 $O/soak_grd_like.txt | cut -c1-400
