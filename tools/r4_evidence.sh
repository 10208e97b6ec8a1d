#!/bin/bash
# Everything DESIGN.md / README quote for round 4, measured in one go on the GPU box; outputs under gpurun_out/r4e/
# (the summaries are then copied to profiles/r4/).  usage (from the repo root on the box): bash tools/r4_evidence.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the driver's command under rocprofv3 --kernel-trace --stats (its nine-scene cycle), and on scene A alone
rocprofv3 --kernel-trace --stats -d $O/bench_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-traffic > $O/bench_line_under_rocprof.json 2> $O/bench_stats.log
rocprofv3 --kernel-trace --stats -d $O/bench_scene_a_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-traffic --scenes 1 > $O/bench_scene_a_line_under_rocprof.json 2> $O/bench_scene_a_stats.log
# 2. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes); SQ counters of the fused pass, of the u16-output CLAHE apply (config 3(i))
bash $R/tools/pmc_traffic.sh $O/pmc_traffic > $O/pmc_traffic.txt 2>&1
PMC_KERNEL=k_clahe_rgb_fused bash $R/tools/pmc_apply.sh $O/pmc_rgb_fused > $O/pmc_rgb_fused.txt 2>&1
PMC_KERNEL=k_clahe_apply_u16 PROFILE_SCRIPT=$R/tools/profile_clahe_u16.py bash $R/tools/pmc_apply.sh $O/pmc_apply_u16_cf > $O/pmc_apply_u16_cf.txt 2>&1   # kernel 4a (the default)
SARPRO_HIP_NO_U16_CF=1 PMC_KERNEL=k_clahe_apply_u16 PROFILE_SCRIPT=$R/tools/profile_clahe_u16.py bash $R/tools/pmc_apply.sh $O/pmc_apply_u16 > $O/pmc_apply_u16.txt 2>&1   # kernel 4
cd $R
# 3. the bench line as the driver runs it (secondary records, full-size CPU baseline, live PMC traffic)
python3 bench.py > $O/bench_full.json 2> $O/bench_full.log
# 4. the tools behind the prose figures
python3 tools/time_scenes.py > $O/time_scenes.txt 2>&1
python3 tools/time_variants.py - -:SARPRO_HIP_NO_FUSED_RGB=1 - > $O/time_routes.txt 2>&1
python3 tools/time_configs.py > $O/time_configs.txt 2>&1
(python3 tools/time_clahe_u16.py -; SARPRO_HIP_NO_U16_CF=1 python3 tools/time_clahe_u16.py -; python3 tools/time_clahe_u16.py -) > $O/time_clahe_u16.txt 2>&1
timeout 600 python3 tools/soak_spec_vs_exact.py 6 > $O/soak_spec_vs_exact.txt 2>&1
timeout 900 python3 tools/soak_routes.py 2 > $O/soak_routes.txt 2>&1
python3 -m pytest tests -q -m gpu > $O/gpu_suite.txt 2>&1
rm -rf $O/pmc_traffic/*/pmc_* $O/pmc_rgb_fused/* $O/pmc_apply_u16/* $O/pmc_apply_u16_cf/* 2>/dev/null
ls -la $O
