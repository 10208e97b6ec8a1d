#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_spec_chain.py tests/test_gpu_resident_batch.py -x -q -m gpu > $O/tests2.txt 2>&1; tail -2 $O/tests2.txt
for ord in 3 0 3 0; do
  SARPRO_HIP_PIPE_ORDER=$ord timeout 900 python bench.py --no-cpu-full --no-secondary --no-traffic > $O/bench_ord$ord.json 2> $O/bench_ord$ord.err
  python3 - <<PY
import json
d=json.load(open("$O/bench_ord$ord.json"))
r=d["roofline"]
print("order $ord", "ms_per_step", d["ms_per_step"], "one_stream", d.get("ms_per_step_one_stream"), "frac", r["frac"], "frac_one_stream", r.get("frac_one_stream"), "worst", r.get("frac_worst_scene"), r.get("worst_scene"), r.get("ms_per_launch_by_scene"))
PY
done | tee $O/bench_orders.txt
