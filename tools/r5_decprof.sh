#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5q; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_decode -- python3 $GRAFT_REPO_ROOT/bench.py --decode --steps 2 --warmup 1 > $O/prof_decode.log 2>&1
cp $(find $O/prof_decode -name "*kernel_stats.csv" | head -1) $O/decode_after_kernel_stats.csv
tail -1 $O/prof_decode.log | cut -c1-300
