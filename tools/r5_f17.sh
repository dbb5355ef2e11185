#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_model.py -q -k "f17m or L14cylin" > $O/pytest_f17.txt 2>&1; tail -15 $O/pytest_f17.txt | cut -c1-300
bash tools/r5_launch_count.sh
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_decode -- python3 $GRAFT_REPO_ROOT/bench.py --decode --steps 2 --warmup 1 > $O/prof_decode.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/prof_decode/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
    print("decode run: kernel time total %.1f ms over %d launches (1 encode + 4 decodes: 1 warm-up, 1 stage-stamped, 2 timed)" % (tot / 1e6, calls))
    for r in rows[:16]: print("%-60s %7d calls %8.2f ms avg %7.1f us" % (r["Name"][:60], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
tail -1 $O/prof_decode.log | cut -c1-400
