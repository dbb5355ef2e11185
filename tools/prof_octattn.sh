#!/bin/bash
# usage (on the GPU box): bash tools/prof_octattn.sh <tag> [level] [cylin]  -> gpurun_out/prof_<tag>/ kernel stats of 3 OctAttention frames
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -- python3 $R/tools/run_octattn.py ${2:-12} ${3:-0} > $R/gpurun_out/prof_$1.log 2>&1
grep frame $R/gpurun_out/prof_$1.log
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_$1/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
nf = 17   # run_octattn.py: 3 single frames + 2 warm-up + 12 pipelined
print(f"total kernel time {tot/nf/1e6:.1f} ms per frame ({nf} frames)")
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls'])/nf:6.1f} {float(r['TotalDurationNs'])/nf/1e6:8.2f} ms")
PY
