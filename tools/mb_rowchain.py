#!/usr/bin/env python3
"""Row-chain kernels (csrc/rowchain.hip) against the launches they replace, on one MI355X: correctness against float64 and time.
    python tools/mb_rowchain.py [rows]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native  # noqa: E402
from scp_amd.ops import linear_s, _split  # noqa: E402


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 590848
    dev = torch.device("cuda:0")
    native.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn((M, 256), generator=g) * 1.5 + 0.3).to(dev)
    gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(dev)
    beta = (0.1 * torch.randn(256, generator=g)).to(dev)
    valid = (torch.rand(M, generator=g) > 0.1).float().to(dev)
    for N in (768, 512, 256):
        W = (torch.randn((N, 256), generator=g) * 0.05).to(dev)
        b = (torch.randn(N, generator=g) * 0.1).to(dev)
        fw = native.LnFoldedWeight(W, gamma, beta)
        out = native.swin_ln_linear(x, fw, b, 1e-5, valid)
        torch.cuda.synchronize()
        # float64 reference on a sample of rows
        idx = torch.cat((torch.arange(0, min(M, 300)), torch.randint(0, M, (2000,), generator=g), torch.arange(max(0, M - 300), M))).to(dev)
        xs = x[idx].double()
        ln = torch.nn.functional.layer_norm(xs, (256,), gamma.double(), beta.double(), 1e-5) * valid[idx].double()[:, None]
        ref = ln @ W.double().T + b.double()
        err = (out[idx].double() - ref).abs().max().item()
        # the launches it replaces
        def old():
            h = native.layernorm_rows(x, gamma, beta, 1e-5, valid=valid[:, None].contiguous(), split=True)
            return linear_s(h, W, b)
        o2 = old()
        err_old = (o2[idx].double() - ref).abs().max().item()
        t_new = timeit(lambda: native.swin_ln_linear(x, fw, b, 1e-5, valid, out=out))
        t_old = timeit(old)
        t_ln = timeit(lambda: native.layernorm_rows(x, gamma, beta, 1e-5, valid=valid[:, None].contiguous(), split=True))
        fl = 2.0 * M * N * 256
        print(f"N={N:4d} M={M}: rowchain {t_new:.3f} ms ({fl / t_new / 1e9:.0f} TF/s alg)  |  LN {t_ln:.3f} + gemm_split {t_old - t_ln:.3f} = {t_old:.3f} ms"
              f"   max err vs f64: new {err:.2e} old {err_old:.2e}", flush=True)
    # ragged M and invalid rows
    for Mr in (1, 31, 129, 1000):
        xr = x[:Mr].contiguous()
        W = (torch.randn((256, 256), generator=g) * 0.05).to(dev)
        fw = native.LnFoldedWeight(W, gamma, beta)
        o = native.swin_ln_linear(xr, fw, None, 1e-5, None)
        ref = torch.nn.functional.layer_norm(xr.double(), (256,), gamma.double(), beta.double(), 1e-5) @ W.double().T
        print(f"M={Mr}: max err {(o.double() - ref).abs().max().item():.2e}")


if __name__ == "__main__":
    main()
