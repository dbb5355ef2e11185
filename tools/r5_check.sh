#!/bin/bash
# full GPU test suite + the default bench line (+ optional extra command), outputs under gpurun_out/<tag>
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-r5chk}; O=gpurun_out/$T; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests.txt
t0=$(date +%s); timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -2 $O/bench.err
python - <<PY
import json
try:
    z=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("value", z["value"], "ms", z["ms_per_step"], "frac", z["roofline"]["frac"], "valid", z["roofline"]["valid"], "device", z.get("device_transform_fps"), "verified", z.get("strict_identity_verified"))
    for k,v in z.get("configs",{}).items(): print(k, {a:v.get(a) for a in ("fps","ms_per_step","error")}, v.get("roofline"))
    print("decode", z.get("decode")); print("decode procs", z.get("decode_2_procs"), z.get("decode_4_procs")); print("cli", z.get("cli"), z.get("cli_over_bench")); print(z.get("legs_error"))
except Exception as e: print("no bench line", e)
PY
