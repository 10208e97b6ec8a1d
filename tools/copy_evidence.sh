#!/bin/bash
# copies the summaries of a tools/r4_evidence.sh run (gpurun_out/r4e) into profiles/r4
E=gpurun_out/r4e; P=profiles/r4
cp $E/bench_full.json $E/bench_line_under_rocprof.json $E/bench_scene_a_line_under_rocprof.json $P/
cp $E/bench_stats/bench_kernel_stats.csv $P/bench_kernel_stats.csv
cp $E/bench_scene_a_stats/bench_kernel_stats.csv $P/bench_scene_a_kernel_stats.csv
cp $E/pmc_traffic.txt $E/pmc_rgb_fused.txt $E/pmc_apply_u16.txt $E/pmc_apply_u16_cf.txt $E/time_scenes.txt $E/time_routes.txt $E/time_configs.txt $E/time_clahe_u16.txt $E/soak_routes.txt $E/soak_spec_vs_exact.txt $P/
grep -E "passed|failed" $E/gpu_suite.txt | tail -1 > $P/gpu_suite.txt
cp $E/pmc_traffic/FETCH_SIZE.json $P/pmc_FETCH_SIZE.json; cp $E/pmc_traffic/WRITE_SIZE.json $P/pmc_WRITE_SIZE.json
cat $P/gpu_suite.txt
