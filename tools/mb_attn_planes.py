#!/usr/bin/env python3
"""Plane-fed window attention (K / V as bf16 hi / lo planes staged by LDS-DMA) against the fp32-fed kernel: identical bits? time?
    python tools/mb_attn_planes.py [windows]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device("cuda:0")
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1152
g = torch.Generator().manual_seed(1)
T = nw * 512
qkv = (torch.randn((T, 768), generator=g) * 2.0).to(dev)
table = (torch.randn((1023, 4), generator=g) * 0.5).to(dev)
# sequences of 1, 2, 3 ... windows
lens, left = [], nw
while left > 0:
    n = min(left, 1 + len(lens) % 5); lens.append(n); left -= n
rows = []
base = 0
for n in lens:
    rows += [[base, n * 512]] * n; base += n * 512
wtab = torch.tensor(rows, dtype=torch.int32, device=dev)
q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for shift in (0, 256):
    a = native.swin_attention_packed(q, k, v, table, wtab, shift)
    kv = native.KvPlanes(k, v)
    b = native.swin_attention_packed_planes(q, kv, table, wtab, shift)
    print(f"shift {shift}: identical {torch.equal(a, b)}  max |d| {(a - b).abs().max().item():.3e}", flush=True)
    t0 = timeit(lambda: native.swin_attention_packed(q, k, v, table, wtab, shift, split=True))
    t1 = timeit(lambda: native.swin_attention_packed_planes(q, kv, table, wtab, shift, split=True))
    t2 = timeit(lambda: native.KvPlanes(k, v))
    print(f"   {nw} windows: fp32-fed {t0:.0f} us, plane-fed {t1:.0f} us (+ standalone plane pass {t2:.0f} us)", flush=True)
    L = native.lib()
    for rep in range(2):
        for var, nm in ((0, "standard online softmax"), (1, "fixed reference 0 first (default)")):
            L.scp_set_attention_variant(var)
            c = native.swin_attention_packed_planes(q, kv, table, wtab, shift)
            tv = timeit(lambda: native.swin_attention_packed_planes(q, kv, table, wtab, shift, split=True), 10)
            print(f"      variant {var} ({nm}): {tv:.0f} us, max |d| to the default {(c - b).abs().max().item():.3e} of {b.abs().max().item():.2f}", flush=True)
    L.scp_set_attention_variant(1)
