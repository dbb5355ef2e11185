#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6e; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 python -m pytest tests -m gpu -x -q -k "mlp3 or one_launch_heads or hier2 or fused_two_stage or packed_forward or roundtrip or phase2 or tiefree or logits_vs_reference or eight_ranks or octattn" > $O/tests.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/tests.txt; grep -n "max err\|max|dlogit|\|8 ranks" $O/tests.txt | head -20
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
z=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", z["value"], "ms", z["ms_per_step"], "frac", z["roofline"]["frac"], "kernel_ms_sum", z["roofline"].get("kernel_ms_sum"), "device", z.get("device_transform_fps"))
for k,v in sorted(z.get("roofline_kernels",{}).items(), key=lambda kv:-kv[1]["total_ms_per_frame"]): print("   %-44s %7.3f ms  %3d launches  frac %s" % (k, v["total_ms_per_frame"], v["launches_per_frame"], v.get("frac")))
PY
timeout 600 python bench.py --decode --steps 2 --warmup 1 > $O/bench_decode.json 2> $O/bench_decode.err; echo "decode rc=$?"
python - <<PY
import json
z=json.loads(open("$O/bench_decode.json").read().strip().splitlines()[-1])
print("decode", z["value"], z.get("stage_ms"), z.get("decoded_occupancy_equals_encoded"))
PY
