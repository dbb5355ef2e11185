"""OctAttention attention timing: python tools/mb_octattn.py [B] [c]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 113
c = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.Generator().manual_seed(0)
t = [torch.randn((B, c, 600), generator=g).to(dev) for _ in range(5)]
for mode in ("f16x3", "f32"):
    native.OCTATTN_MODE = mode
    for _ in range(2): native.octattn_attention(*t, 4)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): native.octattn_attention(*t, 4)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    fl = B * 4 * (c * c / 2) * 150 * 2 * 2
    print(f"{mode}: B={B} c={c}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s (causal half)", flush=True)
