#!/bin/bash
# the two counters VERDICT r5 item 5 asks for, on the frame's two feature-space kNN searches (one rocprofv3 --pmc pass, no tracing options beside it)
R=${GRAFT_REPO_ROOT:?}; cd $R && python scp_amd/build.py > /dev/null 2>&1; mkdir -p $R/gpurun_out/r6knn; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/r6knn/pmc -- python3 $R/tools/run_knn.py 256 > $R/gpurun_out/r6knn/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/r6knn/pmc/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    if "knn_f16x3" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); acc[k]["n"] += 1
out = open("$R/gpurun_out/r6knn/r6_knn_counters.txt", "w")
for k, v in acc.items():
    line = "%s launches %d  COEXEC %.4g  MFMA_BUSY %.4g  BUSY %.4g  WAIT_ANY %.4g  WAVE_CYCLES %.4g  ACTIVE_VALU %.4g  | coexec/mfma_busy %.3f  mfma_busy/busy %.2f  wait_any/wave %.3f  valu/wave %.3f" % (
        k, int(v["n"] / 6), v["SQ_VALU_MFMA_COEXEC_CYCLES"], v["SQ_VALU_MFMA_BUSY_CYCLES"], v["SQ_BUSY_CYCLES"], v["SQ_WAIT_ANY"], v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_VALU"],
        v["SQ_VALU_MFMA_COEXEC_CYCLES"] / max(v["SQ_VALU_MFMA_BUSY_CYCLES"], 1), v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["SQ_BUSY_CYCLES"], 1), v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_WAVE_CYCLES"], 1))
    print(line); out.write(line + "\n")
PY
tail -3 $R/gpurun_out/r6knn/pmc.log
