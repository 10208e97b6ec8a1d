#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6a; mkdir -p $O
for g in 1 2 3; do echo "== RGB_GROUP=$g"; SARPRO_HIP_LIB=$PWD/sarpro_amd/lib_wgtimes.so SARPRO_HIP_RGB_GROUP=$g timeout 300 python tools/rgb_wg_times.py 2>&1 | tail -4; done > $O/wgtimes_groups.txt 2>&1
cat $O/wgtimes_groups.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SARPRO_HIP_PIPE_ORDER=4 SARPRO_HIP_RGB_GROUP=1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr4 -- python3 $R/tools/pipe_trace.py 9 3 > $R/$O/trace_ord4.log 2>&1
MIN_US=0 python3 $R/tools/trace_overlap.py /tmp/tr4 9 > $R/$O/trace_ord4.txt 2>&1
tail -1 $R/$O/trace_ord4.txt
