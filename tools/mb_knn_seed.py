"""Best-case experiment for seeded kNN: capture the frame's three kNN inputs, run each (a) plain, (b) with the exact 20th-best
distance (minus a margin) as a-priori bound, (c) with the bound from the +-10 index neighbours / previous call's neighbours."""
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
import scp_amd.models.ehem as E
enc.encode(xyz)
native.knn_topk_packed = orig
def timeit(f, reps=3):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
def dist_to(x, idx):            # [T,k] distances in the kernel's convention, fp32
    xx = (x * x).sum(1)
    out = torch.empty(idx.shape, dtype=torch.float32, device=dev)
    for a in range(0, x.shape[0], 65536):
        nb = x[idx[a:a + 65536].long()]                     # [b,k,C]
        out[a:a + 65536] = 2 * (nb * x[a:a + 65536, None]).sum(2) - xx[idx[a:a + 65536].long()] - xx[a:a + 65536, None]
    return out
prev = None
for x, ktab in calls:
    T, C = x.shape
    idx = orig(x, ktab)
    t0 = timeit(lambda: orig(x, ktab))
    d = dist_to(x, idx)
    scale = (x * x).sum(1).max().item()
    thr_best = d.min(1)[0] - 1e-4 * scale
    t1 = timeit(lambda: orig(x, ktab, thr_best))
    ok1 = torch.equal(orig(x, ktab, thr_best), idx)
    # realistic seeds: previous call's neighbours (or +-10 index neighbours for the first call)
    if prev is None:
        base = torch.arange(T, device=dev)[:, None] + torch.tensor([i for i in range(-10, 11) if i != 0], device=dev)[None]
        seq0 = ktab[:, 0].long().repeat_interleave(512); seqn = ktab[:, 1].long().repeat_interleave(512)
        seeds = torch.minimum(torch.maximum(base, seq0[:, None]), (seq0 + seqn - 1)[:, None]).int()
    else:
        seeds = prev
    ds = dist_to(x, seeds)
    # distinct seeds only count once: take the 20th best over the de-duplicated set conservatively = min over seeds if all distinct
    thr_seed = ds.min(1)[0] - 1e-4 * scale
    dup = (torch.sort(seeds, 1)[0].diff(dim=1) == 0).any(1)
    thr_seed[dup] = float('-inf')
    t2 = timeit(lambda: orig(x, ktab, thr_seed))
    ok2 = torch.equal(orig(x, ktab, thr_seed), idx)
    tight = ((thr_seed > float('-inf')).float().mean().item())
    print(f"C={C}: plain {t0:6.2f} ms | exact bound {t1:6.2f} ms same={ok1} | seeded {t2:6.2f} ms same={ok2} (rows with a bound {100*tight:.1f}%)", flush=True)
    prev = idx
