#!/bin/bash
# Everything DESIGN.md / README quote for round 6, measured in one go on the GPU box; outputs under gpurun_out/r6e_evidence/
# (tools/copy_evidence.sh copies the summaries to profiles/r6/).  usage (from the repo root on the box): bash tools/r6_evidence.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e_evidence; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the driver's command under rocprofv3 --kernel-trace --stats (resident batch over the nine-scene cycle), and the one-stream loop (--lanes 0)
rocprofv3 --kernel-trace --stats -d $O/bench_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-traffic > $O/bench_line_under_rocprof.json 2> $O/bench_stats.log
rocprofv3 --kernel-trace --stats -d $O/bench_one_stream_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-traffic --lanes 0 > $O/bench_one_stream_line_under_rocprof.json 2> $O/bench_one_stream_stats.log
# 2. the lanes' timeline: kernel trace of one resident batch, overlap per sweep
rocprofv3 --kernel-trace -d $O/pipe_trace -o pipe --output-format csv -- python3 $R/tools/pipe_trace.py 6 > $O/pipe_trace.log 2>&1
python3 $R/tools/trace_overlap.py $O/pipe_trace 6 > $O/pipe_trace_overlap.txt 2>&1
# 3. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes); SQ counters of the fused pass, of the piece histogram, of the resize passes, of the f32 kernels
bash $R/tools/pmc_traffic.sh $O/pmc_traffic > $O/pmc_traffic.txt 2>&1
PMC_KERNEL='k_clahe_rgb_fused(' bash $R/tools/pmc_apply.sh $O/pmc_rgb_fused > $O/pmc_rgb_fused.txt 2>&1
PMC_KERNEL=k_dn_hist_pieces bash $R/tools/pmc_apply.sh $O/pmc_dn_hist_pieces > $O/pmc_dn_hist_pieces.txt 2>&1
PMC_KERNEL=k_resize_h PROFILE_SCRIPT=$R/tools/profile_resize.py bash $R/tools/pmc_apply.sh $O/pmc_resize_h > $O/pmc_resize_h.txt 2>&1
bash $R/tools/pmc_f32.sh $O/pmc_f32 > $O/pmc_f32.txt 2>&1
# 4. kernel-stats of the config-2 and config-3 flows
rocprofv3 --kernel-trace --stats -d $O/config2_stats -o c2 --output-format csv -- python3 $R/tools/profile_resize.py 5 > $O/config2_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/config3_stats -o c3 --output-format csv -- python3 $R/tools/trace_f32.py > $O/config3_stats.log 2>&1
cd $R
# 5. the bench line as the driver runs it (secondary records, full-size CPU baseline, live PMC traffic)
python3 bench.py > $O/bench_full.json 2> $O/bench_full.log
# 6. the tools behind the prose figures
python3 tools/time_scenes.py > $O/time_scenes.txt 2>&1
REPS=4 CONFIGS=3:0:0:0,3:3:0:0,2:3:0:0,3:1:0:0,3:0:0:0,3:3:0:0 python3 tools/pipe_sweep.py > $O/pipe_sweep.txt 2>&1
python3 tools/time_routes.py > $O/time_routes.txt 2>&1
python3 tools/batch_rate_f32.py 2048 64 > $O/batch_rate_f32.txt 2>&1
python3 tools/batch_rate_f32.py 1024 64 >> $O/batch_rate_f32.txt 2>&1
timeout 900 python3 tools/soak_grd_like.py 200 1 8 > $O/soak_grd_like.txt 2>&1
python3 tools/time_resize_flow.py > $O/time_resize_flow.txt 2>&1
python3 tools/time_configs.py > $O/time_configs.txt 2>&1
timeout 600 python3 tools/soak_spec_vs_exact.py 6 > $O/soak_spec_vs_exact.txt 2>&1
timeout 900 python3 tools/soak_routes.py 2 > $O/soak_routes.txt 2>&1
timeout 600 python3 tools/soak_random_rasters.py 90 30 1 > $O/soak_random_rasters.txt 2>&1
python3 -m pytest tests -q -m gpu > $O/gpu_suite.txt 2>&1
for d in bench_stats bench_one_stream_stats config2_stats config3_stats; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
rm -rf $O/pmc_traffic/*/pmc_* $O/pmc_rgb_fused $O/pmc_dn_hist_pieces $O/pmc_resize_h $O/pmc_f32/*/ $O/bench_stats $O/bench_one_stream_stats $O/config2_stats $O/config3_stats 2>/dev/null
find $O/pipe_trace -name "*.csv" -size +2M -delete 2>/dev/null
ls -la $O
