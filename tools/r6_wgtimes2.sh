#!/bin/bash
# fused pass: what taller items cost or save in SUMMED workgroup time (mean_busy_us: the imbalance of few large items aside) -- is the
# sweep order's DRAM locality real, or was it the tail?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6a; mkdir -p $O
for a in "" "RGB_ITEM_ROWS=1250,RGB_TAIL_ROWS=0" "RGB_ITEM_ROWS=2500,RGB_TAIL_ROWS=0" "RGB_ITEM_ROWS=256" "RGB_ITEM_ROWS=128,RGB_TAIL_ROWS=0" "NO_SWEEP_ORDER=1"; do
  echo "== $a"; SARPRO_HIP_LIB=$PWD/sarpro_amd/lib_wgtimes.so ATTRS="$a" timeout 300 python tools/rgb_wg_times.py 2>&1 | tail -4
done > $O/wgtimes_rows.txt 2>&1
cat $O/wgtimes_rows.txt
