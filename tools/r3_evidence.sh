#!/bin/bash
# Everything DESIGN.md / README quote for round 3, measured in one go on the GPU box; outputs under gpurun_out/r3/
# (the summaries are then copied to profiles/r3/).  usage (from the repo root on the box): bash tools/r3_evidence.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the driver's command under rocprofv3 --kernel-trace --stats: default route (fused CLAHE -> RGB pass), then the apply + compose route
rocprofv3 --kernel-trace --stats -d $O/bench_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline > $O/bench_line_under_rocprof.json 2> $O/bench_stats.log
export SARPRO_HIP_NO_FUSED_RGB=1
rocprofv3 --kernel-trace --stats -d $O/bench_apply_compose_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline > $O/bench_apply_compose_line_under_rocprof.json 2> $O/bench_apply_compose_stats.log
unset SARPRO_HIP_NO_FUSED_RGB
# 2. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of both routes' kernels; SQ counters of the fused pass and of the apply pass
bash $R/tools/pmc_traffic.sh $O/pmc_traffic > $O/pmc_traffic.txt 2>&1
SARPRO_HIP_NO_FUSED_RGB=1 bash $R/tools/pmc_traffic.sh $O/pmc_traffic_apply_compose > $O/pmc_traffic_apply_compose.txt 2>&1
PMC_KERNEL=k_clahe_rgb_fused bash $R/tools/pmc_apply.sh $O/pmc_rgb_fused > $O/pmc_rgb_fused.txt 2>&1
SARPRO_HIP_NO_FUSED_RGB=1 PMC_KERNEL=k_clahe_apply_u8_spec bash $R/tools/pmc_apply.sh $O/pmc_apply > $O/pmc_apply.txt 2>&1
SARPRO_HIP_NO_FUSED_RGB=1 SARPRO_HIP_STRIP_ALIGN=8 SARPRO_HIP_NO_SAMPLED_HIST=1 SARPRO_HIP_CHUNK_ROWS=234 PMC_KERNEL=k_clahe_apply_u8_spec bash $R/tools/pmc_apply.sh $O/pmc_apply_r2_form > $O/pmc_apply_r2_form.txt 2>&1
cd $R
# 3. the bench line as the driver runs it (secondary records, full-size CPU baseline)
python3 bench.py > $O/bench_full.json 2> $O/bench_full.log
# 4. the tools behind the prose figures
python3 tools/time_variants.py - -:SARPRO_HIP_NO_FUSED_RGB=1 -:SARPRO_HIP_NO_FUSED_RGB=1,SARPRO_HIP_NO_SAMPLED_HIST=1 -:SARPRO_HIP_NO_FUSED_RGB=1,SARPRO_HIP_NO_SAMPLED_HIST=1,SARPRO_HIP_STRIP_ALIGN=8,SARPRO_HIP_CHUNK_ROWS=234 - > $O/time_routes.txt 2>&1
python3 tools/time_resize_flow.py > $O/time_resize_flow.txt 2>&1
python3 tools/spec_accuracy.py 4 > $O/spec_accuracy.txt 2>&1
python3 tools/size_sweep.py > $O/size_sweep.txt 2>&1
python3 tools/time_configs.py > $O/time_configs.json 2>&1
python3 tools/time_f32_routes.py > $O/time_f32_routes.txt 2>&1
python3 tools/time_dualpol_f32.py 1024 2048 4096 > $O/time_dualpol_f32.txt 2>&1
python3 tools/time_strategies.py > $O/time_strategies.txt 2>&1
python3 tools/resident_batch_rate.py > $O/resident_batch_rate.txt 2>&1
timeout 600 python3 tools/soak_spec_vs_exact.py 6 > $O/soak_spec_vs_exact.txt 2>&1
timeout 600 python3 tools/soak_routes.py 2 > $O/soak_routes.txt 2>&1
timeout 600 python3 tools/soak_zones.py 300 > $O/soak_zones.txt 2>&1
timeout 600 python3 tools/soak_f32_switches.py 2000 > $O/soak_f32_switches.txt 2>&1
# 5. f32 flavour (config 3(ii)): SQ counters, and the kernel timeline of one call with its gaps
bash tools/pmc_f32.sh $O/pmc_f32 > $O/pmc_f32.txt 2>&1; rm -rf $O/pmc_f32
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace_f32 -- python3 $R/tools/trace_f32.py > $O/trace_f32.log 2>&1)
python3 tools/trace_gaps.py $O/trace_f32 --last 40 > $O/trace_f32_timeline.txt 2>&1; rm -rf $O/trace_f32
ls -la $O
