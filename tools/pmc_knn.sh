#!/bin/bash
# usage (GPU box): bash tools/pmc_knn.sh <tag> <shape>   -> FETCH_SIZE pass and SQ wave-state pass over the frame's kNN searches in one shape
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmck_$1_fetch -- python3 $R/tools/run_knn.py $2 > $R/gpurun_out/pmck_$1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmck_$1_sq -- python3 $R/tools/run_knn.py $2 >> $R/gpurun_out/pmck_$1.log 2>&1
python3 - <<PY
import csv, glob, collections
for kind in ("fetch", "sq"):
    f = glob.glob("$R/gpurun_out/pmck_$1_" + kind + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "knn_f16x3" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); acc[k]["n"] += 1
    for k, v in acc.items():
        if kind == "fetch":
            print("$1", k, "launches", int(v["n"]), "FETCH GB/launch (x2 corrected)", round(2 * 1024 * v["FETCH_SIZE"] / v["n"] / 1e9, 2))
        else:
            wc = max(v["SQ_WAVE_CYCLES"], 1)
            print("$1", k, "wait_any %.2f wait_inst %.2f active %.2f valu %.2f mfma_busy %.3f" % (v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_ANY"] / wc, v["SQ_ACTIVE_INST_VALU"] / wc, v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["SQ_BUSY_CYCLES"], 1)))
PY
