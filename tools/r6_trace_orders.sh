cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a; mkdir -p $O
for ord in 3 0; do
  export SARPRO_HIP_PIPE_ORDER=$ord
  rm -rf /tmp/tr$ord
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$ord -- python3 $R/tools/pipe_trace.py 9 3 > $O/trace_ord$ord.log 2>&1
  MIN_US=0 python3 $R/tools/trace_overlap.py /tmp/tr$ord 9 > $O/trace_ord$ord.txt 2>&1
  tail -1 $O/trace_ord$ord.txt
done
