#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6b; mkdir -p $O
N=40 python tools/time_variants.py lib_old.so - lib_old.so - lib_old.so - lib_old.so - 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    p = l.split(' ', 1)
    try:
        d = json.loads(p[1]); print(p[0], 'fused median', d.get('clahe_rgb_fused'), 'min', d.get('fused_min'), 'mean', d.get('fused_mean'), 'hist', d.get('dn_hist_u16'))
    except Exception: print(l.strip()[:300])
" | tee $O/ab2.txt
