#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-r5j}; O=gpurun_out/$T; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 2400 python -m pytest tests -m gpu -q -x -k "ln_linear or ln_qkv or linear_split or roundtrip or decode or packed_forward or batch_invariant" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -8 | cut -c1-250
for v in "1 1" "0 1" "1 0" "0 0"; do set -- $v
  SCP_RC_GROUPS=$1 SCP_GEMM_SMALL=$2 timeout 900 python bench.py --decode --steps 3 --warmup 1 > $O/decode_$1$2.json 2> $O/decode_$1$2.err
  python - <<PY
import json
try:
    z=json.loads(open("$O/decode_$1$2.json").read().strip().splitlines()[-1])
    print("groups=$1 small=$2: decode fps %.3f ms %.1f ok %s stage %s" % (z["value"], z["ms_per_step"], z["decoded_occupancy_equals_encoded"], z["stage_ms"]))
except Exception as e: print("no line", e)
PY
done
