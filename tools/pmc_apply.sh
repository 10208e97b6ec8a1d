#!/bin/bash
# PMC counters of the CLAHE apply kernel (separate rocprofv3 passes, no trace domains besides kernel-trace).
# usage: tools/pmc_apply.sh <outdir> ; prints per-counter sums for kernels whose name contains clahe_apply
cd /tmp && export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/$tag -o pmc --output-format csv -- python3 ${PROFILE_SCRIPT:-$GRAFT_REPO_ROOT/tools/profile_one.py} 2 4 > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections, os
acc = collections.defaultdict(float); n = collections.defaultdict(int)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if os.environ.get("PMC_KERNEL", "clahe_apply") in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
except Exception as e:
    print("ERR", e)
for k, v in acc.items(): print(f"{k} {v / max(n[k],1):.4g} per dispatch ({n[k]} dispatches)")
PY
done
