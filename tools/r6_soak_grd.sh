#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c; mkdir -p $O
timeout 1500 python tools/soak_grd_like.py ${1:-200} 1 8 > $O/soak_grd_like.txt 2>&1; echo "exit $?"; tail -3 $O/soak_grd_like.txt | cut -c1-600; grep -c accepted $O/soak_grd_like.txt
