import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device("cuda:0"); native.lib()
torch.manual_seed(0)
for M, N in ((128, 256), (31, 256), (128, 768), (300, 512)):
    x = torch.randn((M, 256), device=dev)
    W = torch.randn((N, 256), device=dev) * 0.05
    b = torch.randn(N, device=dev)
    fw = native.LnFoldedWeight(W, torch.ones(256, device=dev), torch.zeros(256, device=dev))
    o = native.swin_ln_linear(x, fw, b)
    ref = (torch.nn.functional.layer_norm(x.double(), (256,)) @ W.double().T + b.double()).float()
    bad = (o - ref).abs() > 1e-3
    print(M, N, "bad elements", int(bad.sum()), "of", bad.numel())
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
        print("  bad rows", rows[:40], "...", len(rows)); print("  bad cols", cols[:80], "...", len(cols))
        r, c = bad.nonzero()[0].tolist(); print("  first", r, c, o[r, c].item(), ref[r, c].item())
        # is a bad value equal to some other reference element?
        rr = rows[0]; m = (ref - o[rr, cols[0]]).abs() < 1e-4
        print("  value of", rr, cols[0], "matches ref at", m.nonzero()[:5].tolist())
