#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6b; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -x -q -k "hier2 or fused_two_stage or hierarchical or packed_forward_equals or one_leaf or ford_like or roundtrip" > $O/tests.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/tests.txt; grep -n "max err\|max|dlogit|" $O/tests.txt | head
for c in 0 2 3; do timeout 600 python tools/gemm_split_shapes.py $c > $O/shapes_cfg$c.txt 2>&1; echo "cfg $c: $(tail -1 $O/shapes_cfg$c.txt)"; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
z=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", z["value"], "ms", z["ms_per_step"], "frac", z["roofline"]["frac"], "kernel_ms_sum", z["roofline"].get("kernel_ms_sum"), "device", z.get("device_transform_fps"))
for k,v in sorted(z.get("roofline_kernels",{}).items(), key=lambda kv:-kv[1]["total_ms_per_frame"]): print("   %-44s %7.3f ms  %3d launches  frac %s" % (k, v["total_ms_per_frame"], v["launches_per_frame"], v.get("frac")))
PY
