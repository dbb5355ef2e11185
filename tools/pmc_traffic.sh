#!/bin/bash
# usage (on the GPU box): bash tools/pmc_traffic.sh <tag>
# Two separate counter passes (FETCH_SIZE, WRITE_SIZE) of 2 frames of tools/run_frame.py; no tracing options next to --pmc.
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$1_$c -- python3 $R/tools/run_frame.py 16 1 2 > $R/gpurun_out/pmc_$1_$c.log 2>&1
  ls $R/gpurun_out/pmc_$1_$c/*/ | head -3
done
