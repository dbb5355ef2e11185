#!/bin/bash
# usage (on the GPU box): bash tools/pmc_rowchain.sh   -- SQ counter passes over tools/run_rowchain.py (no tracing options next to --pmc)
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_rc1 -- python3 $R/tools/${RC_SCRIPT:-run_rowchain.py} > $R/gpurun_out/pmc_rc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_rc2 -- python3 $R/tools/${RC_SCRIPT:-run_rowchain.py} > $R/gpurun_out/pmc_rc2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_IFETCH SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_rc3 -- python3 $R/tools/${RC_SCRIPT:-run_rowchain.py} > $R/gpurun_out/pmc_rc3.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for tag in ("rc1", "rc2", "rc3"):
    fs = glob.glob(f"{root}/gpurun_out/pmc_{tag}/*/*counter_collection.csv")
    if not fs:
        print(tag, "no output:", open(f"{root}/gpurun_out/pmc_{tag}.log").read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); acc[k]["n"] += 1
    for k, v in acc.items():
        if "rc_" in k:
            print(tag, k, {a: f"{b:.4g}" for a, b in v.items()})
PY
