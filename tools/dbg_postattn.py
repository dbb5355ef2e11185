import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device("cuda:0"); native.lib()
g = torch.Generator().manual_seed(1)
rn = lambda *sh, s=1.0: (torch.randn(sh, generator=g) * s).to(dev)
M = 256
x, ofull = rn(M, 256), rn(M, 256)
wp, bp = rn(256, 256, s=0.05), rn(256, s=0.1)
gamma, beta = 1 + rn(256, s=0.1), rn(256, s=0.1)
w1, b1, w2, b2 = rn(1024, 256, s=0.05), rn(1024, s=0.1), rn(256, 1024, s=0.03), rn(256, s=0.1)
o = native.split_rows(ofull)
def ref(wp, bp, gamma, beta, w1, b1, w2, b2):
    x1 = x.double() + ofull.double() @ wp.double().T + bp.double()
    h = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x1, (256,), gamma.double(), beta.double(), 1e-5) @ w1.double().T + b1.double())
    return x1 + h @ w2.double().T + b2.double()
def run(name, **kw):
    p = dict(wp=wp, bp=bp, gamma=gamma, beta=beta, w1=w1, b1=b1, w2=w2, b2=b2); p.update(kw)
    pw = native.PostAttnWeights(p["wp"], p["bp"], p["gamma"], p["beta"], p["w1"], p["b1"], p["w2"], p["b2"])
    y = native.swin_post_attn(o, x, pw).double()
    r = ref(**p)
    e = (y - r).abs()
    bad = ~(e < 1e-3)
    print(f"{name:28s} max err {e[~bad].max().item() if (~bad).any() else float('nan'):.2e}  bad {int(bad.sum())}/{bad.numel()}  nan {int(torch.isnan(y).sum())}", end="")
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
        print(f"  rows {rows[:8]}..({len(rows)}) cols {cols[:12]}..({len(cols)})", end="")
    print()
z = torch.zeros_like
run("w2=0 (x1 + b2 only)", w2=z(w2))
run("w2=0, wp=0", w2=z(w2), wp=z(wp))
run("w1=0 (H = gelu(b1) const)", w1=z(w1))
run("gamma=1,beta=0", gamma=torch.ones_like(gamma), beta=z(beta))
run("full")

# dump the normalised rows / statistics
M1 = 128
pw = native.PostAttnWeights(wp, bp, gamma, beta, w1, b1, w2, b2)
x1 = x[:M1].double() + ofull[:M1].double() @ wp.double().T + bp.double()
nrm = (x1 - x1.mean(1, keepdim=True)) / torch.sqrt(x1.var(1, unbiased=False, keepdim=True) + 1e-5)
os.environ["SCP_RC_DUMP"] = "1"
d = native.swin_post_attn(native.split_rows(ofull[:M1].contiguous()), x[:M1].contiguous(), pw).double()
perm = native.rc_perm16(256, dev)
print("dump 1: nan", int(torch.isnan(d).sum()), "max |d - normalised| (natural order)", (d - nrm).abs().nan_to_num(9).max().item())
os.environ["SCP_RC_DUMP"] = "2"
d2 = native.swin_post_attn(native.split_rows(ofull[:M1].contiguous()), x[:M1].contiguous(), pw).double()
print("dump 2: mean col0 err", (d2[:, 0] - x1.mean(1)).abs().max().item(), " rstd col1 err", (d2[:, 1] - 1 / torch.sqrt(x1.var(1, unbiased=False) + 1e-5)).abs().max().item(), "nan", int(torch.isnan(d2).sum()))
print(d2[:6, :2], x1.mean(1)[:6])
os.environ.pop("SCP_RC_DUMP")
