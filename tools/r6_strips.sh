#!/bin/bash
# fused pass, strips + chunks (round 6): parity first (with a short timeout: a lost claim would hang the waves that wait for it), then times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spec_chain.py tests/test_gpu_resident_batch.py tests/test_gpu_multirank_local.py -x -q -m gpu > $O/strips_tests.txt 2>&1; tail -3 $O/strips_tests.txt
grep -q passed $O/strips_tests.txt || exit 1
for sh in 6 5 7; do
  echo "== RGB_CHUNK_SHIFT=$sh"
  SARPRO_HIP_RGB_CHUNK_SHIFT=$sh timeout 300 python tools/time_scenes.py 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    p = l.split(' ', 2)
    if len(p) == 3 and p[2].startswith('{'):
        d = json.loads(p[2]); print(p[0], p[1], 'fused', d.get('clahe_rgb_fused'), 'hist', d.get('dn_hist_u16'))
"
done > $O/strips_times.txt 2>&1
cat $O/strips_times.txt
echo "== wg times"; SARPRO_HIP_LIB=$PWD/sarpro_amd/lib_wgtimes.so timeout 300 python tools/rgb_wg_times.py 2>&1 | tail -4 | tee $O/strips_wgtimes.txt
