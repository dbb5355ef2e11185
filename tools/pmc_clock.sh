#!/bin/bash
# usage (on the GPU box): bash tools/pmc_clock.sh   -> the clock every hot kernel of the L16-m frame runs at: GRBM_GUI_ACTIVE / (End - Start) per dispatch
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; O=$R/gpurun_out/clockpmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p -- python3 $R/tools/run_frame.py 16 1 3 > $O/log.txt 2>&1
python3 - <<EOF
import csv, glob, collections
f = glob.glob("$O/p/*/*counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if dur <= 0: continue
    a = acc[r["Kernel_Name"][:56]]
    a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
for k, (c, d, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-56s launches %4d  total %8.2f ms  cycles/ns %.3f" % (k, n, d / 1e6, c / d))
EOF
