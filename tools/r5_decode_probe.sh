#!/bin/bash
# round 5: is the decoder host-bound or GPU-latency-bound?  cProfile of one decode, kernel-time sum (rocprofv3 --stats) of a 2-frame decode bench
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1
timeout 600 python tools/decode_cprofile.py > $O/decode_cprofile.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_bench.py -x -q > $O/pytest_dist_bench.txt 2>&1; tail -3 $O/pytest_dist_bench.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_e2e.py -x -q -k "f17m or L14cylin or c8192 or host_transform or batched" > $O/pytest_new.txt 2>&1; tail -3 $O/pytest_new.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_decode -- python3 $GRAFT_REPO_ROOT/bench.py --decode --steps 2 --warmup 1 > $O/prof_decode.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/prof_decode/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
    print("decode run: kernel time total %.1f ms over %d launches (3 decodes + 1 encode + stage decode)" % (tot / 1e6, calls))
    for r in rows[:14]: print("%-60s %7d calls %8.2f ms avg %7.1f us" % (r["Name"][:60], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
head -45 $O/decode_cprofile.txt
