#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5f; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
for n in 8 24; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps $n --warmup 4 --no-legs --no-strict-leg --cpu-baseline none > $O/bench_$n.json 2> $O/bench_$n.err
done
A=$(find $O/prof_8 -name "*kernel_stats.csv" | head -1); B=$(find $O/prof_24 -name "*kernel_stats.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/launch_count.py $A $B 8 24 > $O/launches_per_frame.md; head -60 $O/launches_per_frame.md
cp $B $O/bench24_kernel_stats.csv
