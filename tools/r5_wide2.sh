#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r5m; mkdir -p $O
cat > /tmp/tw.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from scp_amd import native
L = native.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
rn = lambda *sh, s=1.0: (torch.randn(sh, generator=g) * s).to(dev)
pw = native.PostAttnWeights(rn(256, 256, s=0.05), rn(256, s=0.1), 1 + rn(256, s=0.1), rn(256, s=0.1), rn(1024, 256, s=0.05), rn(1024, s=0.1), rn(256, 1024, s=0.03), rn(256, s=0.1))
out = []
for M in (512, 4096, 8192):
    x, o = rn(M, 256), native.split_rows(rn(M, 256)); y = torch.empty_like(x)
    L.scp_rc_set_wide(0); want = native.swin_post_attn(o, x, pw)
    L.scp_rc_set_wide(1); ok = torch.equal(native.swin_post_attn(o, x, pw), want)
    for _ in range(5): native.swin_post_attn(o, x, pw, out=y)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): native.swin_post_attn(o, x, pw, out=y)
    e.record(); torch.cuda.synchronize()
    out.append(f"M={M}: {1e3 * s.elapsed_time(e) / 50:.1f} us ok={ok}")
print("; ".join(out))
PY
for d in "8 4" "4 2" "12 4" "8 8" "16 4"; do set -- $d
  SCP_RC_DEFS="-DRCW_D1=$1 -DRCW_D2=$2" python scp_amd/build.py > $O/build_$1_$2.log 2>&1 || { echo "build failed $d"; continue; }
  echo "D1=$1 D2=$2: $(python /tmp/tw.py 2>&1 | tail -1)"
done
python scp_amd/build.py > $O/build.log 2>&1
