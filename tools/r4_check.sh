#!/bin/bash
# usage (on the GPU box): bash tools/r4_check.sh <tag> : GPU tests, the default bench line, the decoder line, rocprofv3 kernel stats of 4 frames
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests -q -m gpu > $O/${T}_gpu_tests.txt 2>&1; tail -15 $O/${T}_gpu_tests.txt
timeout 600 python3 bench.py > $O/${T}_bench.log 2>&1; grep '^{' $O/${T}_bench.log | tail -1 > $O/${T}_bench.json; cut -c1-400 $O/${T}_bench.json; tail -5 $O/${T}_bench.log | grep -v '^{' | cut -c1-300
timeout 300 python3 bench.py --decode > $O/${T}_decode.log 2>&1; grep '^{' $O/${T}_decode.log | tail -1 > $O/${T}_bench_decode.json; cut -c1-600 $O/${T}_bench_decode.json
bash tools/prof_frame.sh $T
cp $R/gpurun_out/prof_$T/*/*_kernel_stats.csv $O/${T}_frame_kernel_stats.csv 2>/dev/null
python3 - <<PY
import json
b = json.load(open("$O/${T}_bench.json"))
print("fps", b["value"], "valid", b["roofline"]["valid"], b["roofline"]["kernel_ms_sum"], b["roofline"]["kernel_ms_bound"], "strict", b.get("strict_identity_fps"), b.get("transform_parity"))
for k, v in b["roofline_kernels"].items():
    print(f"{k[:40]:40s} {v['launches_per_frame']:4d} {v['avg_launch_us']:9.1f} us  {v['total_ms_per_frame']:7.2f} ms  frac {v.get('frac', 0):.3f}")
print(b["roofline_frame"])
PY
