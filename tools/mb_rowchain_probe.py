#!/usr/bin/env python3
"""Where a step of the row-chain LN + linear kernel spends its time: the same launch with parts compiled out at run time
(SCP_RC_PROBE bits: 1 no stores, 2 no DMA, 4 no MFMAs, 8 no epilogue; results are wrong with any of them)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SCP_RC_PROBE"] = "16"      # any non-zero value at first use enables re-reading the variable on every call
from scp_amd import native
M = 590848
dev = torch.device("cuda:0")
native.lib()
x = torch.randn((M, 256), device=dev)
W = torch.randn((768, 256), device=dev) * 0.05
fw = native.LnFoldedWeight(W, torch.ones(256, device=dev), torch.zeros(256, device=dev))
out = torch.empty((M, 768), device=dev)
def t(n=10):
    native.swin_ln_linear(x, fw, None, out=out); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): native.swin_ln_linear(x, fw, None, out=out)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
import ctypes
L = native.lib()
L.scp_rc_debug_buffer.argtypes = [ctypes.c_void_p]
buf = torch.zeros((256 * 4 * 8,), dtype=torch.int64, device=dev)
print("probe  ms      cycles per tile and wave: barrier waits / steps (12 x 3072 MFMA floor) / LayerNorm / drain")
for p in (16, 1, 2, 8, 10):
    os.environ["SCP_RC_PROBE"] = str(p)
    ms = t()
    L.scp_rc_debug_buffer(buf.data_ptr())
    buf.zero_()
    native.swin_ln_linear(x, fw, None, out=out); torch.cuda.synchronize()
    L.scp_rc_debug_buffer(None)
    b = buf.cpu().view(256, 4, 8).double()
    tiles = b[:, :, 4].clamp(min=1)
    print(f"{p:3d}  {ms:.3f}   " + " / ".join(f"{(b[:, :, i] / tiles).mean():8.0f}" for i in range(4)) + "   per step: slice 0 / 1-7 / 8-15 (without the barrier wait = first column / 12): " + " / ".join(f"{(b[:, :, i] / tiles / 12).mean():6.0f}" for i in (5, 6, 7)), flush=True)
