#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f32.py -x -q -m gpu -k "resident_batch_of_f32 or batch_driver" > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 900 python tools/batch_rate_f32.py 2048 64 2>&1 | tee $O/batch_rate_f32.txt | cut -c1-600
timeout 600 python tools/batch_rate_f32.py 1024 64 2>&1 | tee -a $O/batch_rate_f32.txt | cut -c1-600
