#!/bin/bash
# generic round-5 GPU job: build, a pytest selection, a list of bench configurations.   bash tools/r5_job.sh <tag> "<pytest -k expr>" "<config> ..."
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=$1; K=$2; CFGS=$3; O=gpurun_out/$T; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
if [ -n "$K" ]; then timeout 2400 python -m pytest tests -m gpu -q -k "$K" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -12 | cut -c1-250; fi
for c in $CFGS; do
  timeout 900 python bench.py --config $c --no-legs --cpu-baseline none > $O/bench_$c.json 2> $O/bench_$c.err
  python - <<PY
import json
try:
    z=json.loads(open("$O/bench_$c.json").read().strip().splitlines()[-1])
    print("$c", "fps %.2f ms %.2f frac %.3f valid %s device %s" % (z["value"], z["ms_per_step"], z["roofline"]["frac"], z["roofline"]["valid"], z.get("device_transform_fps")))
    for k,v in list(z["roofline_kernels"].items())[:9]: print("   %-44s %7.2f ms %6.3f" % (k[:44], v["total_ms_per_frame"], v.get("frac",0)))
except Exception as e: print("$c: no line", e)
PY
done
