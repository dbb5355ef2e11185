#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r5t; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 python -m pytest tests -m gpu -q -x -k "knn or roundtrip or decode_mode or tiefree" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|assert|Error" $O/pytest.txt | tail -8 | cut -c1-250
for v in 0 1; do
  SCP_KNN_SPLIT=$v timeout 900 python bench.py --decode --steps 3 --warmup 1 > $O/decode_$v.json 2> $O/decode_$v.err
  python - <<PY
import json
try:
    z=json.loads(open("$O/decode_$v.json").read().strip().splitlines()[-1])
    print("knn split=$v: decode fps %.3f ms %.1f ok %s stage %s" % (z["value"], z["ms_per_step"], z["decoded_occupancy_equals_encoded"], z["stage_ms"]))
except Exception as e: print("no line", e); print(open("$O/decode_$v.err").read()[-600:])
PY
done
