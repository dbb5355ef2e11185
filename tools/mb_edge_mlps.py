#!/usr/bin/env python3
"""scp_geo_edge_mlps (both edge MLPs of the geometry feature generator in one row-chain launch) against float64 and against the six split GEMMs."""
import os, sys, torch
import torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
from scp_amd.ops import leaky_mlp3_s, split_cat
dev = torch.device("cuda:0")
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
torch.manual_seed(0)
mk = lambda a, b, c, d: nn.Sequential(nn.Linear(a, b), nn.LeakyReLU(0.01), nn.Linear(b, c), nn.LeakyReLU(0.01), nn.Linear(c, d)).to(dev)
m1, m2 = mk(448, 256, 256, 256), mk(512, 256, 256, 128)
for M in (1000, 590848):
    g = torch.Generator().manual_seed(M)
    p1, p2, p3 = (torch.randn((M, c), generator=g).to(dev) for c in (64, 128, 256))
    ew = native.EdgeMlpWeights(m1, m2)
    out = torch.empty((M, 256), device=dev)
    native.geo_edge_mlps(p1, p2, p3, ew, out[:, 128:])
    def old():
        e_in = native.SplitAct.empty(M, 512, dev)
        native.split_rows(p3, out=e_in.cols(0, 256))
        leaky_mlp3_s(m1, split_cat((p1, p2, p3)), want="split", out_split=e_in.cols(256, 512))
        o = torch.empty((M, 256), device=dev)
        leaky_mlp3_s(m2, e_in, out=o[:, 128:])
        return o
    o0 = old()
    idx = torch.cat((torch.arange(0, min(M, 300)), torch.randint(0, M, (1000,), generator=g), torch.arange(max(0, M - 300), M))).to(dev)
    with torch.no_grad():
        d1, d2 = m1.double(), m2.double()
        ref = d2(torch.cat((p3[idx].double(), d1(torch.cat((p1[idx], p2[idx], p3[idx]), 1).double())), 1))
        m1.float(); m2.float()
    print(f"M={M}: max err vs float64: one launch {(out[idx, 128:].double() - ref).abs().max().item():.2e}, six launches {(o0[idx, 128:].double() - ref).abs().max().item():.2e}"
          f" (|ref| max {ref.abs().max().item():.2f});  {timeit(lambda: native.geo_edge_mlps(p1, p2, p3, ew, out[:, 128:])):.3f} ms against {timeit(old):.3f} ms", flush=True)
