#!/bin/bash
# usage (on the GPU box): bash tools/round_profiles.sh <tag>
# Everything the round's profiles/ entries come from, into gpurun_out/<tag>/: the GPU test log, the default bench line (with the CPU
# baseline), rocprofv3 --kernel-trace --stats of the same bench command, and the five configuration lines of `bench.py --all-configs`.
R=${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repository root)}; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -q -m gpu -x > $O/${T}_gpu_tests.txt 2>&1; tail -2 $O/${T}_gpu_tests.txt
timeout 600 python3 bench.py > $O/${T}_bench.log 2>&1; grep '^{' $O/${T}_bench.log | tail -1 > $O/${T}_bench.json; cut -c1-300 $O/${T}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profb -- python3 $R/bench.py --no-cpu-baseline > $O/${T}_profb.log 2>&1
cp $O/profb/*/*_kernel_stats.csv $O/${T}_bench_kernel_stats.csv 2>/dev/null; grep '^{' $O/${T}_profb.log | tail -1 | cut -c1-200
rm -rf $O/profb
cd $R
timeout 1500 python3 bench.py --all-configs --out-dir $O --tag $T > $O/${T}_all_configs.log 2>&1; cut -c1-160 $O/${T}_all_configs.log
