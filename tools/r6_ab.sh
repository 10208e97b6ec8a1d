#!/bin/bash
# A/B on one box, interleaved: round 5's item hand-out (lib_old.so) against the strips + chunks build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6a; mkdir -p $O
python tools/time_variants.py lib_old.so - lib_old.so - "-:SARPRO_HIP_RGB_CHUNK_SHIFT=5" lib_old.so - 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    p = l.split(' ', 1)
    try:
        d = json.loads(p[1]); print(p[0], 'fused', d.get('clahe_rgb_fused'), 'hist', d.get('dn_hist_u16'))
    except Exception: print(l.strip()[:300])
" | tee $O/ab_strips.txt
