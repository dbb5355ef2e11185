#!/usr/bin/env python3
"""scp_swin_merge (patch merging in one row-chain launch) against float64 and against layernorm_rows(gather) + gemm_split."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
from scp_amd.ops import linear_s
dev = torch.device("cuda:0")
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for n_src in (1000, 303616 * 2):
    g = torch.Generator().manual_seed(n_src)
    x = (torch.randn((n_src, 256), generator=g) * 1.3 + 0.2).to(dev)
    M = (n_src + 1) // 2
    ev = torch.arange(0, 2 * M, 2); od = ev + 1
    od[od >= n_src] = n_src
    ev[5 % M] = n_src                                   # a zero row in the even slot too
    ev, od = ev.to(dev), od.to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(512, generator=g)).to(dev), (0.1 * torch.randn(512, generator=g)).to(dev)
    W = (torch.randn((256, 512), generator=g) * 0.04).to(dev)
    mw = native.MergeWeights(W, gamma, beta)
    y = native.swin_merge(x, ev, od, mw)
    old = lambda: linear_s(native.layernorm_rows(x, gamma, beta, 1e-5, ia=ev, ib=od, split=True), W, None)
    y0 = old()
    idx = torch.cat((torch.arange(0, min(M, 300)), torch.randint(0, M, (1000,), generator=g), torch.arange(max(0, M - 300), M))).to(dev)
    xz = torch.cat((x, torch.zeros((1, 256), device=dev))).double()
    cat = torch.cat((xz[ev[idx]], xz[od[idx]]), 1)
    ref = torch.nn.functional.layer_norm(cat, (512,), gamma.double(), beta.double(), 1e-5) @ W.double().T
    print(f"n_src={n_src} M={M}: max err vs float64: merge kernel {(y[idx].double() - ref).abs().max().item():.2e}, two launches {(y0[idx].double() - ref).abs().max().item():.2e};"
          f"  {timeit(lambda: native.swin_merge(x, ev, od, mw)):.3f} ms against {timeit(old):.3f} ms", flush=True)
