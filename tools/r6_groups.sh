#!/bin/bash
# fused pass: hand-out groups of 1 / 2 / 3 items (RGB_GROUP, planner attribute), per-kernel times per scene, configurations interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_spec_chain.py tests/test_gpu_resident_batch.py tests/test_gpu_multirank_local.py -x -q -m gpu > $O/groups_tests.txt 2>&1; tail -3 $O/groups_tests.txt
SARPRO_HIP_RGB_GROUP=3 timeout 600 python -m pytest tests/test_gpu_spec_chain.py -x -q -m gpu > $O/groups3_tests.txt 2>&1; tail -2 $O/groups3_tests.txt
for rep in 1 2; do for g in 1 2 3 4; do
  echo "== RGB_GROUP=$g rep $rep"
  SARPRO_HIP_RGB_GROUP=$g timeout 300 python tools/time_scenes.py 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    p = l.split(' ', 2)
    if len(p) == 3 and p[2].startswith('{'):
        d = json.loads(p[2]); print(p[0], p[1], 'fused', d.get('clahe_rgb_fused'), 'hist', d.get('dn_hist_u16'))
"
done; done > $O/groups_times.txt 2>&1
cat $O/groups_times.txt
