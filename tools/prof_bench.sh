#!/bin/bash
# usage (GPU box): bash tools/prof_bench.sh <tag> : kernel trace of bench.py (pipelined mode), 6 frames
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profb_$1 -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline > $R/gpurun_out/profb_$1.log 2>&1
tail -1 $R/gpurun_out/profb_$1.log | cut -c1-200
