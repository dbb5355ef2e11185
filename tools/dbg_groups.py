import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for N in (768, 512, 256):
    W, b = (torch.randn((N, 256), generator=g) * 0.05).to(dev), (torch.randn(N, generator=g) * 0.1).to(dev)
    fw = native.LnFoldedWeight(W, torch.ones(256, device=dev), torch.zeros(256, device=dev))
    for M in (512, 4096, 70144):
        x = torch.randn((M, 256), generator=g).to(dev)
        try:
            o = native.swin_ln_linear(x, fw, b, 1e-5, None); torch.cuda.synchronize()
            print("ln_linear", N, M, "ok", float(o.abs().max()))
        except Exception as e:
            print("ln_linear", N, M, "FAILED", e, native.lib().scp_last_hip_error())
        if N != 256:
            try:
                q, kv = native.swin_ln_qkv(x, fw, b, 1e-5, None); torch.cuda.synchronize()
                print("ln_qkv", N, M, "ok")
            except Exception as e:
                print("ln_qkv", N, M, "FAILED", e, native.lib().scp_last_hip_error())
