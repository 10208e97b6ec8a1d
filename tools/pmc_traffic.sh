#!/bin/bash
# HBM traffic counters per kernel of the headline pass: two SEPARATE rocprofv3 passes (FETCH_SIZE, WRITE_SIZE),
# kernel-trace only.  usage: tools/pmc_traffic.sh <outdir> ; prints per-kernel means (raw counter units: KiB... see guide)
cd /tmp && export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/profile_one.py 3 4 > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" "$OUT/$c.json" <<'PY'
import csv, sys, collections, json
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1]
    acc[k] += float(r["Counter_Value"]); n[k] += 1
out = {k: {"mean": acc[k] / n[k], "dispatches": n[k]} for k in acc}
json.dump({"counter": sys.argv[2], "per_kernel": out}, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mean"])[:8]: print(sys.argv[2], k, f"{v['mean']:.6g}", v["dispatches"])
PY
done
