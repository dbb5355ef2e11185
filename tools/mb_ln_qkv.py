#!/usr/bin/env python3
"""scp_swin_ln_qkv against scp_swin_ln_linear + scp_swin_kv_planes: q identical? planes identical? time per 590 848 rows?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device("cuda:0")
for M, N in ((512, 768), (512, 512), (1024, 768), (70144, 768), (70144, 512)):
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn((M, 256), generator=g) * 1.5 + 0.3).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(256, generator=g)).to(dev), (0.1 * torch.randn(256, generator=g)).to(dev)
    valid = (torch.rand(M, generator=g) > 0.1).float().to(dev)
    W = (torch.randn((N, 256), generator=g) * 0.05).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    fw = native.LnFoldedWeight(W, gamma, beta)
    ref = native.swin_ln_linear(x, fw, b, 1e-5, valid)
    nq = N - 512
    want = native.KvPlanes(ref[:, nq:nq + 256], ref[:, nq + 256:])
    q, kv = native.swin_ln_qkv(x, fw, b, 1e-5, valid)
    torch.cuda.synchronize()
    okq = True if q is None else torch.equal(q, ref[:, :256].contiguous())
    a, w = kv.t.view(torch.int16), want.t.view(torch.int16)
    print(f"M={M} N={N}: q identical {okq}; planes identical: K hi {torch.equal(a[0], w[0])} K lo {torch.equal(a[1], w[1])} Vt hi {torch.equal(a[2], w[2])} Vt lo {torch.equal(a[3], w[3])}", flush=True)
    for i in range(4):
        if not torch.equal(a[i], w[i]):
            d = (a[i] != w[i]).nonzero()
            print("   plane", i, "first diffs", d[:6].tolist(), "count", d.shape[0])

def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
M = 590848
g = torch.Generator().manual_seed(1)
x = (torch.randn((M, 256), generator=g)).to(dev)
for N in (768, 512):
    W = (torch.randn((N, 256), generator=g) * 0.05).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    fw = native.LnFoldedWeight(W, torch.ones(256, device=dev), torch.zeros(256, device=dev))
    out = torch.empty((M, N), device=dev)
    print(f"N={N} M={M}: swin_ln_linear {timeit(lambda: native.swin_ln_linear(x, fw, b, out=out)):.3f} ms, swin_ln_qkv {timeit(lambda: native.swin_ln_qkv(x, fw, b)):.3f} ms", flush=True)
