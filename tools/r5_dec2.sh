#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r5s; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests -m gpu -q -x -k "weights or roundtrip or decode_mode or cli_encode_then_decode" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -5 | cut -c1-250
timeout 900 python bench.py --decode --steps 3 --warmup 1 > $O/decode.json 2> $O/decode.err
python - <<PY
import json
z=json.loads(open("$O/decode.json").read().strip().splitlines()[-1])
print("decode fps %.3f ms %.1f ok %s host cpu %.0f stage %s" % (z["value"], z["ms_per_step"], z["decoded_occupancy_equals_encoded"], z["host_cpu_ms_per_frame"], z["stage_ms"]))
PY
timeout 600 python tools/decode_cprofile.py 2>&1 | head -34
