#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-r5l}; O=gpurun_out/$T; mkdir -p $O
python scp_amd/build.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_model.py -q -x -k "wide_post_attn or swin_post_attn" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|assert|Error" $O/pytest.txt | tail -8 | cut -c1-250
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from scp_amd import native
L = native.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
rn = lambda *sh, s=1.0: (torch.randn(sh, generator=g) * s).to(dev)
pw = native.PostAttnWeights(rn(256, 256, s=0.05), rn(256, s=0.1), 1 + rn(256, s=0.1), rn(256, s=0.1), rn(1024, 256, s=0.05), rn(1024, s=0.1), rn(256, 1024, s=0.03), rn(256, s=0.1))
for M in (512, 2048, 4096, 8192, 16384, 32768):
    x, o = rn(M, 256), native.split_rows(rn(M, 256)); y = torch.empty_like(x)
    res = []
    for mode in (0, 1):
        L.scp_rc_set_wide(mode)
        for _ in range(5): native.swin_post_attn(o, x, pw, out=y)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): native.swin_post_attn(o, x, pw, out=y)
        e.record(); torch.cuda.synchronize()
        res.append(1e3 * s.elapsed_time(e) / 50)
    print(f"M={M}: chain {res[0]:.1f} us, wide {res[1]:.1f} us per launch")
L.scp_rc_set_wide(-1)
PY
timeout 900 python -m pytest tests -m gpu -q -x -k "roundtrip or decode_mode or packed_forward or batch_invariant" > $O/pytest2.txt 2>&1; echo "pytest2 rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest2.txt | tail -5 | cut -c1-250
for v in 0 -1; do
  SCP_RC_WIDE=$v timeout 900 python bench.py --decode --steps 3 --warmup 1 > $O/decode_w$v.json 2> $O/decode_w$v.err
  python - <<PY
import json
try:
    z=json.loads(open("$O/decode_w$v.json").read().strip().splitlines()[-1])
    print("wide=$v: decode fps %.3f ms %.1f ok %s stage %s" % (z["value"], z["ms_per_step"], z["decoded_occupancy_equals_encoded"], z["stage_ms"]))
except Exception as e: print("no line", e)
PY
done
