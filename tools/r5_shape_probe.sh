#!/bin/bash
# round 5: rc_post_attn_kernel with every 32x32x16 product issued as two v_mfma_f32_16x16x32_bf16 (RC_PROBE_SHAPE16: WRONG results, the real step's operand
# traffic, LDS-DMA, fragment reads and vector work) against the product build: wall time per launch, cycles per tile, in-kernel clock
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r5s16; mkdir -p $O
run() {   # tag, defs
    SCP_RC_DEFS="$2" python scp_amd/build.py > $O/build_$1.log 2>&1 || { echo "build $1 failed"; tail -5 $O/build_$1.log; return; }
    RC_STAMPS=1 timeout 300 python tools/mb_postattn.py > $O/postattn_$1.txt 2>&1
    echo "== $1 ($2)"; grep -E "^rowchain|cycles per tile|clock" $O/postattn_$1.txt
}
run base ""
run shape16 "-DRC_PROBE_SHAPE16"
run base2 ""
run shape16_nodma "-DRC_PROBE_SHAPE16 -DRC_PROBE_NODMA"
run nodma "-DRC_PROBE_NODMA"
python scp_amd/build.py > $O/build_final.log 2>&1
