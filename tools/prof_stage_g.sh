#!/bin/bash
# usage (on the GPU box): bash tools/prof_stage_g.sh <tag> : kernel stats of stage G alone at L12 same-level, 1 and 4 frames per scp_geom_build
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; cd /tmp && export TMPDIR=/tmp
for b in 1 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profg_$1_b$b -- python3 $R/tools/run_geom_batch.py $b 20 > $R/gpurun_out/profg_$1_b$b.log 2>&1
  tail -1 $R/gpurun_out/profg_$1_b$b.log | cut -c1-200
  cp $R/gpurun_out/profg_$1_b$b/*/*_kernel_stats.csv $R/gpurun_out/profg_$1_b$b.csv
done
