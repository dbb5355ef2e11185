#!/bin/bash
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; cd /tmp && export TMPDIR=/tmp
for d in 0 1 2; do
  SCP_KNN_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_knn$d -- python3 $R/tools/run_frame.py 16 1 3 > $R/gpurun_out/prof_knn$d.log 2>&1
  echo DBG=$d; grep -h "knn_" $R/gpurun_out/prof_knn$d/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-
done
