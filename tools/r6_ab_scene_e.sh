#!/bin/bash
# the fused pass on the nine scenes (tools/time_scenes.py) for a build of HEAD (sarpro_amd/lib_head.so) and the working tree's, alternating; the route tests first
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_spec_chain.py tests/test_gpu_resident_batch.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do for l in lib_head.so -; do
  if [ "$l" = "-" ]; then unset SARPRO_HIP_LIB; else export SARPRO_HIP_LIB=$PWD/sarpro_amd/$l; fi
  echo "$l $(python tools/time_scenes.py 2>/dev/null | grep -o '^[A-Z]-[a-zA-Z-]* \|^A \|"clahe_rgb_fused": [0-9.]*' | paste - - | awk '{printf "%s=%s ", $1, $NF}')"
done; done 2>&1 | tee gpurun_out/ab_scene_e.txt
