"""Upper bound on tile-level early exit in the feature-space kNN: for the frame's own searches, the fraction of (32-query tile, 32-candidate
tile) pairs in which EVERY pair is already farther than the query's final 20th-neighbour distance on the first A features alone."""
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
    full = [int(b) for b, n in tab[::16] if n == 8192][:6:2]        # three full windows
    for A in (C // 4, C // 3, C // 2, 2 * C // 3):
        A = A // 16 * 16
        rates = []
        for base in full:
            w = x[base:base + 8192].double()
            d = torch.cdist(w, w) ** 2
            thr = torch.sort(d, 1)[0][:, 19]                      # final 20th-best squared distance per query
            dA = torch.cdist(w[:, :A], w[:, :A]) ** 2
            far = dA > thr[:, None] * 1.0001                      # pair certainly outside the final top 20
            t = far.reshape(256, 32, 256, 32).all(3).all(1)       # [query tile, candidate tile]
            rates.append(t.float().mean().item())
        print(f"C={C} A={A}: tile pairs prunable after {A} features (final bounds) {[round(r, 3) for r in rates]}", flush=True)
