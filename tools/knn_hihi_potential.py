"""How often would a certified one-product (hi.hi) first pass of the f16x3 feature search have to fall back to the three-product tile?
For the frame's own searches: fraction of (32-query tile, 32-candidate tile) pairs holding at least one pair whose distance lies within
the error band of the one-product value (band = 2 * 1.5 * 2^-10 * |x| |y| on the squared distance) of the query's bound - with the FINAL
20th-best distance as the bound (optimistic) and with the 40th-best (roughly what a half-list holds mid-sweep)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import ehem_cfg
from scp_amd import native
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.synth import synth_frame
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
enc = FrameEncoder(model, 'kitti', 16, spher=True, mullevel=True, device=dev)
xyz = torch.from_numpy(synth_frame(0)).to(dev)
calls = []
orig = native.knn_topk_packed
def rec(x, ktab, thr0=None):
    calls.append((x.clone(), ktab.clone()))
    return orig(x, ktab)
native.knn_topk_packed = rec
enc.encode(xyz)
native.knn_topk_packed = orig
for x, ktab in calls:
    C = x.shape[1]
    if C <= 4: continue
    tab = ktab.cpu().numpy()
    full = [int(b) for b, n in tab[::16] if n == 8192][:8:2]
    for rank in (19, 39, 79):
        rates = []
        for base in full:
            w = x[base:base + 8192].double()
            d = torch.cdist(w, w) ** 2
            thr = torch.sort(d, 1)[0][:, rank]
            r = w.norm(dim=1)
            band = 2 * 1.5 * 2.0 ** -10 * r[:, None] * r[None, :]
            possible = d < thr[:, None] + band
            t = possible.reshape(256, 32, 256, 32).any(3).any(1)
            t0 = (d < thr[:, None]).reshape(256, 32, 256, 32).any(3).any(1)
            rates.append((round(t.float().mean().item(), 3), round(t0.float().mean().item(), 3)))
        print(f"C={C} bound = {rank + 1}th best: tile pairs needing the three-product tile (with band, without band) {rates};  median |x|^2 {float((r*r).median()):.1f}, median bound {float(thr.median()):.3f}", flush=True)
