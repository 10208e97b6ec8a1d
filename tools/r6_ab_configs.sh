#!/bin/bash
# A/B of the secondary configurations (tools/time_configs.py) between a build of HEAD (sarpro_amd/lib_head.so) and the working tree's
mkdir -p gpurun_out
for i in 1 2 3; do
  for l in lib_head.so -; do
    if [ "$l" = "-" ]; then unset SARPRO_HIP_LIB; else export SARPRO_HIP_LIB=$PWD/sarpro_amd/$l; fi
    python tools/time_configs.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', {k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('cfg2_robust_dualpol_synrgb_ms','cfg3_clahe_u16_per_band_ms','cfg3_clahe_u16_kernels','cfg3_ratio_f32_clahe_u16_ms','cfg3_ratio_f32_kernels','f32_robust_u8_ms','f32_standard_u8_ms','f32_robust_u16_ms','cfg1_2048_f32_standard_u8_ms')})"
  done
done 2>&1 | tee gpurun_out/ab_configs.txt | cut -c1-900
