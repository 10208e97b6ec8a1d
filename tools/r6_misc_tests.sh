#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stripes_resized.py tests/test_gpu_stripes_f32.py -x -q -m gpu > $O/tests_stripes.txt 2>&1; tail -5 $O/tests_stripes.txt
