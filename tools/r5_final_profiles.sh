#!/bin/bash
# end of round 5: the OctAttention L14 frame (kernel stats, PMC traffic, SQ wave states) and the decoder's kernel stats with the final code
cd "${GRAFT_REPO_ROOT:?}" || exit 1
bash tools/r5_profiles.sh r5f octattn
O=$GRAFT_REPO_ROOT/gpurun_out/r5f
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_decode -- python3 $GRAFT_REPO_ROOT/bench.py --decode --steps 2 --warmup 1 > $O/prof_decode.log 2>&1
cp $(find $O/prof_decode -name "*kernel_stats.csv" | head -1) $O/r5f_decode_final_kernel_stats.csv; rm -rf $O/prof_decode
tail -1 $O/prof_decode.log | cut -c1-200
ls $O
