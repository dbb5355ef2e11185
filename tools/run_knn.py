"""The frame's two feature-space kNN searches under each workgroup shape given on the command line (for rocprofv3 passes):
python tools/run_knn.py 128 256 272 ...   (every shape is launched `REPS` times; kernels of different shapes have different names
except the order bit: shape s > 255 runs knn_f16x3_wg256_kernel<K, G>)."""
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
torch.cuda.synchronize()
REPS = 2
for sh in [int(a) for a in sys.argv[1:]]:
    native.set_knn_workgroup(sh)
    for x, ktab in calls:
        if x.shape[1] > 4:
            for _ in range(REPS):
                orig(x, ktab)
    torch.cuda.synchronize()
print("done")
