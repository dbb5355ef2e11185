#!/bin/bash
# round 5: what a step of rc_post_attn_kernel's MLP phase spends beyond its 96 MFMAs (3 072 cycles): probe builds with parts compiled out
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r5c; mkdir -p $O
run() {   # tag, defs
    SCP_RC_DEFS="$2" python scp_amd/build.py > $O/build_$1.log 2>&1 || { echo "build $1 failed"; tail -5 $O/build_$1.log; return; }
    RC_STAMPS=1 timeout 300 python tools/mb_postattn.py > $O/postattn_$1.txt 2>&1
    echo "== $1 ($2)"; grep -E "^rowchain|cycles per tile" $O/postattn_$1.txt
}
run base ""
run nogelu "-DRC_NOGELU"
run nogelu_nodma "-DRC_NOGELU -DRC_PROBE_NODMA"
run nogelu_nofrag "-DRC_NOGELU -DRC_PROBE_NOFRAG"
run nogelu_nodma_nofrag "-DRC_NOGELU -DRC_PROBE_NODMA -DRC_PROBE_NOFRAG"
run nogelu_nodma_nofrag_nobar "-DRC_NOGELU -DRC_PROBE_NODMA -DRC_PROBE_NOFRAG -DRC_PROBE_NOBAR"
run nodma "-DRC_PROBE_NODMA"
run nofrag "-DRC_PROBE_NOFRAG"
run nobar_nodma "-DRC_PROBE_NODMA -DRC_PROBE_NOBAR"
python scp_amd/build.py > $O/build_final.log 2>&1
