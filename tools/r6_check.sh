#!/bin/bash
# full GPU test suite + the bench line as the driver runs it (--steps 20 --warmup 5), outputs under gpurun_out/<tag>
#   tools/r6_check.sh <tag> [pytest -k expression | all | none] [bench args...]
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-r6chk}; K=${2:-all}; shift 2
O=gpurun_out/$T; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
if [ "$K" = all ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests.txt
elif [ "$K" != none ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q -k "$K" > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/gpu_tests.txt
fi
t0=$(date +%s); timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -2 $O/bench.err
python - <<PY
import json
try:
    z=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("value", z["value"], "ms", z["ms_per_step"], "frac", z["roofline"]["frac"], "valid", z["roofline"]["valid"], "kernel_ms_sum", z["roofline"].get("kernel_ms_sum"), "device", z.get("device_transform_fps"), "verified", z.get("strict_identity_verified"))
    for k,v in sorted(z.get("roofline_kernels",{}).items(), key=lambda kv:-kv[1]["total_ms_per_frame"]): print("   %-44s %7.3f ms  %3d launches  frac %s" % (k, v["total_ms_per_frame"], v["launches_per_frame"], v.get("frac")))
    for k,v in z.get("configs",{}).items(): print(k, {a:v.get(a) for a in ("fps","ms_per_step","error","strict_identity_verified","transform_parity")}, (v.get("roofline") or {}).get("frac"))
    d=z.get("decode") or {}
    print("decode", d.get("fps"), d.get("stage_ms")); print("decode procs", (z.get("decode_2_procs") or {}).get("fps"), (z.get("decode_4_procs") or {}).get("fps")); print("cli", (z.get("cli") or {}).get("fps"), z.get("cli_over_bench")); print(z.get("legs_error"))
    c=z.get("cpu_baseline") or {}
    print("cpu", c.get("seconds_per_frame"), c.get("full_window_s"), c.get("full_window_cost_s"), c.get("sampled_over_full_frame_recorded"))
except Exception as e: print("no bench line", e)
PY
