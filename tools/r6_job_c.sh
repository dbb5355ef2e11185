#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6c; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for c in 0 3; do timeout 600 python tools/gemm_split_shapes.py $c > $O/shapes_cfg$c.txt 2>&1; echo "cfg $c: $(tail -1 $O/shapes_cfg$c.txt)"; done
cat $O/shapes_cfg0.txt | tail -24
for cfg in octattn-L14-cylin octattn-L12-spher; do
  timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --no-legs --no-cpu-baseline --config $cfg > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?"
  python - <<PY
import json
z=json.loads(open("$O/bench_$cfg.json").read().strip().splitlines()[-1])
print("$cfg value", z["value"], "ms", z["ms_per_step"])
for k,v in sorted(z.get("roofline_kernels",{}).items(), key=lambda kv:-kv[1]["total_ms_per_frame"]): print("   %-44s %7.3f ms  %3d launches  frac %s" % (k, v["total_ms_per_frame"], v["launches_per_frame"], v.get("frac")))
PY
done
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/gpu_tests.txt
