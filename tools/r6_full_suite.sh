#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.txt 2>&1; tail -5 $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
