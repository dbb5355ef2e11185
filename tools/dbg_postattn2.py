import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device("cuda:0"); native.lib()
g = torch.Generator().manual_seed(1)
rn = lambda *sh, s=1.0: (torch.randn(sh, generator=g) * s).to(dev)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x, ofull = rn(M, 256), rn(M, 256)
wp, bp = rn(256, 256, s=0.05), rn(256, s=0.1)
gamma, beta = 1 + rn(256, s=0.1), rn(256, s=0.1)
w1, b1, w2, b2 = rn(1024, 256, s=0.05), rn(1024, s=0.1), rn(256, 1024, s=0.03), rn(256, s=0.1)
pw = native.PostAttnWeights(wp, bp, gamma, beta, w1, b1, w2, b2)
y = native.swin_post_attn(native.split_rows(ofull), x, pw)
torch.cuda.synchronize()
print("ran M =", M, "dump =", os.environ.get("SCP_RC_DUMP"), "finite:", bool(torch.isfinite(y).all()))
