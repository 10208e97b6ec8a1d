#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_spec_chain.py tests/test_gpu_resident_batch.py tests/test_gpu_multirank_local.py tests/test_gpu_attrs.py -x -q -m gpu > $O/tests.txt 2>&1; tail -15 $O/tests.txt
grep -q "passed" $O/tests.txt && ! grep -q failed $O/tests.txt || exit 1
timeout 300 python tools/time_routes.py 2>&1 | tee $O/time_routes.txt | cut -c1-700
