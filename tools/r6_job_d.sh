#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6d; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for s in 0 1 2 4; do SCP_GEMM_STAGGER=$s timeout 600 python tools/gemm_split_shapes.py > $O/shapes_stagger$s.txt 2>&1; echo "stagger $s: $(tail -1 $O/shapes_stagger$s.txt)"; done
paste <(cut -c1-78 $O/shapes_stagger0.txt | tail -21) <(cut -c60-78 $O/shapes_stagger1.txt | tail -21) <(cut -c60-78 $O/shapes_stagger2.txt | tail -21) <(cut -c60-78 $O/shapes_stagger4.txt | tail -21)
for s in 0 1 2; do
for cfg in ehem-L16-m octattn-L14-cylin; do
  SCP_GEMM_STAGGER=$s timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline --config $cfg > $O/bench_${cfg}_s$s.json 2> $O/bench_${cfg}_s$s.err
  python - <<PY
import json
z=json.loads(open("$O/bench_${cfg}_s$s.json").read().strip().splitlines()[-1])
print("stagger $s $cfg value", z["value"], "ms", z["ms_per_step"], "gemm_split", z["roofline_kernels"]["gemm_split_kernel"]["total_ms_per_frame"], "kernel_ms_sum", z["roofline"].get("kernel_ms_sum"))
PY
done; done
