#!/bin/bash
# fused pass with the software-pipelined prologue: parity, A/B against round 5's kernel (lib_old.so) on this box, workgroup stamps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spec_chain.py tests/test_gpu_resident_batch.py tests/test_gpu_multirank_local.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
grep -q passed $O/tests.txt || exit 1
python tools/time_variants.py lib_old.so - lib_old.so - lib_old.so - 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    p = l.split(' ', 1)
    try:
        d = json.loads(p[1]); print(p[0], 'fused', d.get('clahe_rgb_fused'), 'hist', d.get('dn_hist_u16'))
    except Exception: print(l.strip()[:300])
" | tee $O/ab.txt
SARPRO_HIP_LIB=$PWD/sarpro_amd/lib_wgtimes.so timeout 300 python tools/rgb_wg_times.py 2>&1 | tail -4 | tee $O/wgtimes.txt
