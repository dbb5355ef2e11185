#!/bin/bash
# usage (on the GPU box): bash tools/prof_frame.sh <tag>   -> gpurun_out/prof_<tag>/ kernel stats of 4 frames of tools/run_frame.py
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1 -- python3 $R/tools/run_frame.py 16 1 4 > $R/gpurun_out/prof_$1.log 2>&1
tail -2 $R/gpurun_out/prof_$1.log | cut -c1-200
