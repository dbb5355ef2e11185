#!/bin/bash
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; O=$R/gpurun_out/ldspmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p -- python3 $R/tools/run_frame.py 16 1 3 > $O/log.txt 2>&1
ls $O/p/*/ | head
python3 - <<EOF
import csv, glob, collections
f = glob.glob("$O/p/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:16]
print("%-60s %12s %12s %8s %12s" % ("kernel", "lds_active", "bank_confl", "frac", "insts_lds"))
for k, v in rows:
    a, c = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)
    print("%-60s %12.3e %12.3e %8.3f %12.3e" % (k, a, c, c / a if a else 0, v.get("SQ_INSTS_LDS", 0)))
EOF
