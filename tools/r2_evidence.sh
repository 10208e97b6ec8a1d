#!/bin/bash
# Everything DESIGN.md / README quote for round 2, measured in one go on the GPU box; outputs under gpurun_out/r2/
# (the summaries are then copied to profiles/r2_*).  usage (from the repo root on the box): bash tools/r2_evidence.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the driver's command under rocprofv3 --kernel-trace --stats (default route, then the opt-in fused pass)
rocprofv3 --kernel-trace --stats -d $O/bench_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary > $O/bench_line_under_rocprof.json 2> $O/bench_stats.log
rocprofv3 --kernel-trace --stats -d $O/bench_fused_stats -o bench --output-format csv -- python3 $R/bench.py --no-secondary --fused > $O/bench_fused_line_under_rocprof.json 2> $O/bench_fused_stats.log
# 2. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and the SQ counters of the apply kernel
bash $R/tools/pmc_traffic.sh $O/pmc_traffic > $O/pmc_traffic.txt 2>&1
bash $R/tools/pmc_apply.sh $O/pmc_apply > $O/pmc_apply.txt 2>&1
cd $R
# 3. the bench line as the driver runs it (with the secondary records)
python3 bench.py > $O/bench_full.json 2> $O/bench_full.log
# 4. the tools behind the prose figures
python3 tools/time_configs.py > $O/time_configs.json 2>&1
python3 tools/e2e_host_rate.py > $O/e2e_host_rate.txt 2>&1
python3 tools/e2e_tiff_rate.py > $O/e2e_tiff_rate.json 2>&1
python3 tools/resident_batch_rate.py > $O/resident_batch_rate.txt 2>&1
python3 tools/predict_accuracy.py > $O/predict_accuracy.txt 2>&1
python3 tools/time_strategies.py > $O/time_strategies.txt 2>&1
timeout 600 python3 tools/soak_spec_vs_exact.py 6 > $O/soak_spec_vs_exact.txt 2>&1
timeout 600 python3 tools/soak_routes.py 2 > $O/soak_routes.txt 2>&1
python3 tools/time_f32_routes.py > $O/time_f32_routes.txt 2>&1
timeout 900 python3 tools/soak_zones.py 300 > $O/soak_zones_300.txt 2>&1
bash tools/pitch_sweep.sh > $O/pitch_sweep.txt 2>&1
bash tools/pmc_f32.sh $O/pmc_f32 > $O/pmc_f32.txt 2>&1
cd $R
[ -x build/stream_bench ] && ./build/stream_bench > $O/stream_bench.txt 2>&1
[ -x build/instr_bench ] && ./build/instr_bench > $O/instr_bench.txt 2>&1
# 5. timeline of the f32 flavour (config 3(ii))
cd /tmp
rocprofv3 --kernel-trace -d $O/trace_f32 -- python3 $R/tools/trace_f32.py > $O/trace_f32.log 2>&1
cd $R
python3 tools/trace_gaps.py $O/trace_f32 --last 40 > $O/trace_f32_timeline.txt 2>&1
ls -la $O
