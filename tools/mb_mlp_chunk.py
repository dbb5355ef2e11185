"""Does chunking fc1 -> fc2 by rows keep the 1024-wide hidden activation in the memory-side cache?  (timing experiment)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device('cuda:0')
M = 590848
h = torch.randn((M, 256), device=dev); x = torch.randn((M, 256), device=dev)
w1 = torch.randn((1024, 256), device=dev) / 16; b1 = torch.randn(1024, device=dev)
w2 = torch.randn((256, 1024), device=dev) / 32; b2 = torch.randn(256, device=dev)
s1, s2 = native.SplitWeight(w1), native.SplitWeight(w2)
hs = native.split_rows(h)
hid = native.SplitAct.empty(M, 1024, dev)
out = torch.empty((M, 256), device=dev)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
def full():
    native.linear_split(hs, s1, b1, 2, None, want="split", out_split=hid)
    native.linear_split(hid, s2, b2, 0, x, out=out)
ref = None
def chunked(C, cfg2=0):
    def f():
        for r0 in range(0, M, C):
            r1 = min(M, r0 + C)
            a = native.SplitAct(hs.t[:, r0:r1], 256)
            hc = native.SplitAct(hid.t[:, :r1 - r0], 1024)          # the SAME scratch rows for every chunk: stays cache resident
            native.linear_split(a, s1, b1, 2, None, want="split", out_split=hc)
            native.linear_split(hc, s2, b2, 0, x[r0:r1], out=out[r0:r1], cfg=cfg2)
    return f
t = timeit(full); print(f"full fc1+fc2: {t*1e3:.0f} us")
full(); ref = out.clone()
for C in (16384, 32768, 65536, 131072):
    for cfg2 in (1, 2):
        t = timeit(chunked(C, cfg2)); chunked(C, cfg2)()
        print(f"chunk {C:6d} rows, fc2 cfg {cfg2}: {t*1e3:.0f} us  same={torch.equal(out, ref)}")
