#!/bin/bash
# SQ counters of the f32-flavour kernels (config 3(ii)), separate rocprofv3 passes.  usage: tools/pmc_f32.sh <outdir>
cd /tmp && export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/$tag -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/trace_f32.py > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1]
        if k.startswith("k_f32"):
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
except Exception as e:
    print("ERR", e)
for (k, c), v in sorted(acc.items()): print(f"{k:40s} {c:24s} {v / max(n[(k, c)], 1):14.4g} per dispatch ({n[(k, c)]})")
PY
done
