"""Workgroup shapes of the packed f16x3 kNN on the frame's own three searches: 128-query kernel vs 256-query XCD-scheduled kernel
(with / without the half-step stagger).  Interleaved rounds in one process; neighbour lists must be identical."""
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
def timeit(f, reps=3):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
shapes = (128, 257, 256)
for x, ktab in calls:
    if x.shape[1] <= 4:
        continue
    ref = None
    for sh in shapes:
        native.set_knn_workgroup(sh)
        got = orig(x, ktab)
        if ref is None: ref = got
        else: assert torch.equal(ref, got), f"shape {sh} differs"
    times = {sh: [] for sh in shapes}
    for rnd in range(4):
        for sh in shapes:
            native.set_knn_workgroup(sh)
            orig(x, ktab)
            times[sh].append(timeit(lambda: orig(x, ktab)))
    print(f"C={x.shape[1]} rows={x.shape[0]}: " + "  ".join(f"{sh}: min {min(t):.2f} med {sorted(t)[len(t)//2]:.2f} ms" for sh, t in times.items()), flush=True)
native.set_knn_workgroup(256)
