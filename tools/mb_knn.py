"""kNN kernel timing on frame-like data: python tools/mb_knn.py  (SCP_KNN_DBG=1: MFMA only, 2: + pass 1)"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from scp_amd import native
dev = torch.device('cuda:0')
def timeit(f, reps=3, warm=1):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
nw = 70
g = torch.Generator().manual_seed(0)
ktab = torch.tensor([[w * 8192, 8192] for w in range(nw) for _ in range(16)], dtype=torch.int32, device=dev)
for C in (3, 144, 192):
    # locally correlated features (a random walk along the token order), like octree siblings
    base = torch.cumsum(torch.randn((nw * 8192, C), generator=g) * 0.05, 0) + torch.randn((nw * 8192, C), generator=g) * 0.3
    x = base.to(dev).contiguous()
    ms = timeit(lambda: native.knn_topk_packed(x, ktab))
    fl = 2.0 * nw * 8192 * 8192 * max(C, 4)
    print(f"knn packed {nw} x 8192 C={C}: {ms:8.2f} ms  {fl/ms/1e9:7.2f} TFLOP/s  ({ms/nw*100:.2f} ms per 100 windows)", flush=True)
