#!/bin/bash
# copies the summaries of a tools/r6_evidence.sh run (gpurun_out/r6e_evidence) into profiles/r6  (usage: tools/copy_evidence.sh [run dir] [profiles dir])
E=${1:-gpurun_out/r6e_evidence}; P=${2:-profiles/r6}; mkdir -p $P
for f in bench_full.json bench_line_under_rocprof.json bench_one_stream_line_under_rocprof.json bench_stats_kernel_stats.csv bench_one_stream_stats_kernel_stats.csv \
         config2_stats_kernel_stats.csv config3_stats_kernel_stats.csv pipe_trace_overlap.txt pmc_traffic.txt pmc_rgb_fused.txt pmc_dn_hist_pieces.txt \
         pmc_resize_h.txt pmc_resize_h_before.txt pmc_f32.txt time_scenes.txt pipe_sweep.txt time_resize_flow.txt time_configs.txt soak_routes.txt soak_random_rasters.txt \
         soak_spec_vs_exact.txt spec_margin.txt resize_variants.txt time_routes.txt batch_rate_f32.txt soak_grd_like.txt; do [ -f $E/$f ] && cp $E/$f $P/; done
grep -E "passed|failed" $E/gpu_suite.txt | tail -1 > $P/gpu_suite.txt
[ -f $E/pmc_traffic/FETCH_SIZE.json ] && cp $E/pmc_traffic/FETCH_SIZE.json $P/pmc_FETCH_SIZE.json
[ -f $E/pmc_traffic/WRITE_SIZE.json ] && cp $E/pmc_traffic/WRITE_SIZE.json $P/pmc_WRITE_SIZE.json
cat $P/gpu_suite.txt
