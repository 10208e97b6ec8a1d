#!/bin/bash
# step time and the three big kernels of the headline chain against the row pitch of the resident rasters
for pa in 64 128 256 1024 2048 4096; do
  python3 bench.py --no-secondary --no-cpu-baseline --pitch-align $pa 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels_ms_per_step']
print('pitch-align', sys.argv[1], 'ms/step', d['ms_per_step'], 'dn_hist', k['dn_hist_u16'], 'apply', k['clahe_apply_u8_spec'], 'compose', k['compose_u8'])" $pa
done
