#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r5n; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1
timeout 600 python tools/gemm_split_shapes.py > $O/gemm_split_shapes.txt 2>&1; tail -32 $O/gemm_split_shapes.txt
for b in 4 8 16; do
  timeout 600 python bench.py --config ehem-L12-s --batch $b --steps 32 --warmup 4 --no-legs --cpu-baseline none > $O/l12_b$b.json 2> $O/l12_b$b.err
  python - <<PY
import json
try:
    z=json.loads(open("$O/l12_b$b.json").read().strip().splitlines()[-1])
    print("L12-s batch $b: strict %.2f fps, device %.2f fps, host cpu %.1f ms" % (z["value"], z.get("device_transform_fps") or 0, z["host_cpu_ms_per_frame"]))
except Exception as e: print("no line", e)
PY
done
