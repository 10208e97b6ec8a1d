#!/bin/bash
# A/B of the working tree's build against a build of HEAD (sarpro_amd/lib_head.so: git stash; make; cp; git stash pop; make) on one box; the route tests of the working tree first
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_spec_chain.py tests/test_gpu_resident_batch.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/ab_head_tests.txt
N=${N:-30} python tools/time_variants.py ${LIBS:-lib_head.so - lib_head.so - lib_head.so -} > gpurun_out/ab_head.txt 2>&1
cat gpurun_out/ab_head_tests.txt; cut -c1-14 gpurun_out/ab_head.txt | paste -d' ' - <(grep -o '"clahe_rgb_fused": [0-9.]*' gpurun_out/ab_head.txt) <(grep -o '"fused_min": [0-9.]*' gpurun_out/ab_head.txt)
