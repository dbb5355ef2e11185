#!/bin/bash
# round 5: the post-attention kernel with the 10-instruction GELU: accuracy, time, cycle stamps; probe build without the activation
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r5a; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1
RC_STAMPS=1 timeout 300 python tools/mb_postattn.py > $O/mb_postattn.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "post_attn or swin or logits or tiefree or packed" > $O/pytest_model.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
SCP_RC_DEFS="-DRC_NOGELU" python scp_amd/build.py > $O/build_nogelu.log 2>&1
RC_STAMPS=1 timeout 300 python tools/mb_postattn.py > $O/mb_postattn_nogelu.txt 2>&1
python scp_amd/build.py >> $O/build.log 2>&1
tail -5 $O/mb_postattn.txt $O/mb_postattn_nogelu.txt $O/pytest_model.txt; cat $O/bench.json | head -c 600
