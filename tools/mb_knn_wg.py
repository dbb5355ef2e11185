"""Workgroup shapes of the packed f16x3 kNN on the frame's own three searches: 128-query kernel vs 256-query XCD-scheduled kernel
(MB_SHAPES=128,256,...).  Interleaved rounds in one process; neighbour lists must be identical."""
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
shapes = tuple(int(v) for v in os.environ.get('MB_SHAPES', '128,256').split(','))
for x, ktab in calls:
    if x.shape[1] <= 4:      # position search (exact fp32 kernel, one shape): time only
        orig(x, ktab)
        print(f"C={x.shape[1]} rows={x.shape[0]}: " + "  ".join(f"{timeit(lambda: orig(x, ktab)):.2f}" for _ in range(4)) + " ms", flush=True)
        continue
    ref = None
    for sh in shapes:
        native.set_knn_workgroup(sh)
        got = orig(x, ktab)
        if ref is None: ref = got
        elif not torch.equal(ref, got): print(f"shape {sh}: {(ref != got).any(1).float().mean().item() * 100:.4f} % of the rows differ", flush=True)
    times = {sh: [] for sh in shapes}
    for rnd in range(4):
        for sh in shapes:
            native.set_knn_workgroup(sh)
            orig(x, ktab)
            times[sh].append(timeit(lambda: orig(x, ktab)))
    print(f"C={x.shape[1]} rows={x.shape[0]}: " + "  ".join(f"{sh}: min {min(t):.2f} med {sorted(t)[len(t)//2]:.2f} ms" for sh, t in times.items()), flush=True)
native.set_knn_workgroup(256)
# cycle stamps of the K = 192 search (diagnostic build)
for x, ktab in calls:
    if x.shape[1] != 192:
        continue
    nb = (x.shape[0] // 256 // 8 + 64) * 8 + x.shape[0] // 256          # blocks of the launch (knn_launch: nslots * 8 + nspill)
    buf = torch.zeros((nb * 8 * 2, 4), dtype=torch.int64, device=dev)   # 8 u64 per wave: rows 2 j hold the stamps
    for shp in (256, 256 + 32, 256 + 32 + 64, 256 + 64):     # +32: only waves 0-3 compute (one wave per SIMD); +64: no fragment reads
        buf.zero_()
        native.set_knn_workgroup(shp)
        native.lib().scp_knn_debug_buffer(buf.data_ptr())
        orig(x, ktab); torch.cuda.synchronize()
        native.lib().scp_knn_debug_buffer(None)
        native.set_knn_workgroup(256)
        b = buf[(buf[:, 3] > 0) & (buf[:, 1] > 0)].double()
        f = b[b[:, 3] == 256]
        print(f"shape {shp} stamps over {b.shape[0]} waves: cycles per tile  sync+issue {float((b[:,0]/b[:,3]).mean()):.0f}  mfma {float((b[:,1]/b[:,3]).mean()):.0f}  "
              f"select {float((b[:,2]/b[:,3]).mean()):.0f}   (full windows only: sync {float((f[:,0]/256).mean()):.0f} mfma {float((f[:,1]/256).mean()):.0f} "
              f"select {float((f[:,2]/256).mean()):.0f})", flush=True)
