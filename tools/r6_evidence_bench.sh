#!/bin/bash
# the bench part of tools/r6_evidence.sh alone (after a change to bench.py): the driver's command under rocprofv3 --kernel-trace --stats,
# the one-stream loop, and the full line
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e_evidence; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bench_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-traffic > $O/bench_line_under_rocprof.json 2> $O/bench_stats.log
rocprofv3 --kernel-trace --stats -d $O/bench_one_stream_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-traffic --lanes 0 > $O/bench_one_stream_line_under_rocprof.json 2> $O/bench_one_stream_stats.log
cd $R
( time python3 bench.py > $O/bench_full.json 2> $O/bench_full.log ) 2> $O/bench_full_time.txt
for d in bench_stats bench_one_stream_stats; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
rm -rf $O/bench_stats $O/bench_one_stream_stats
head -3 $O/bench_stats_kernel_stats.csv | cut -c1-160; head -3 $O/bench_one_stream_stats_kernel_stats.csv | cut -c1-160; cat $O/bench_full_time.txt
