#!/bin/bash
# round 5: do the four waves of a row-chain workgroup collide on the CU's address path when they issue their LDS-DMA pieces in lockstep?
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r5b; mkdir -p $O
run() {   # tag, defs
    SCP_RC_DEFS="$2" python scp_amd/build.py > $O/build_$1.log 2>&1 || { echo "build $1 failed"; tail -5 $O/build_$1.log; return; }
    RC_STAMPS=1 timeout 300 python tools/mb_postattn.py > $O/postattn_$1.txt 2>&1
    timeout 300 python tools/mb_rowchain.py > $O/rowchain_$1.txt 2>&1
    echo "== $1 ($2)"; grep -E "^rowchain|cycles per tile|max err" $O/postattn_$1.txt; grep -E "ms" $O/rowchain_$1.txt | head -8
}
run base ""
run skew1 "-DRC_SKEW=1"
run skew2 "-DRC_SKEW=2"
run spread "-DRC_DMA_SPREAD"
run spread_skew1 "-DRC_DMA_SPREAD -DRC_SKEW=1"
run skew1_nogelu "-DRC_SKEW=1 -DRC_NOGELU"
python scp_amd/build.py > $O/build_final.log 2>&1
