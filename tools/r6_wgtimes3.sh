#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6a; mkdir -p $O
for a in "RGB_CHUNK_SHIFT=12" "RGB_CHUNK_SHIFT=6" "RGB_CHUNK_SHIFT=8"; do
  echo "== $a"; SARPRO_HIP_LIB=$PWD/sarpro_amd/lib_wgtimes.so ATTRS="$a" timeout 300 python tools/rgb_wg_times.py 2>&1 | tail -4
done > $O/wgtimes_chunks.txt 2>&1
cat $O/wgtimes_chunks.txt
