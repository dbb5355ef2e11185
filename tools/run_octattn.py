"""OctAttention frame encode timing: python tools/run_octattn.py [level] [cylin 0/1]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import octattn_cfg
from scp_amd.models import OctAttention
from scp_amd.weights import fill_weights
from scp_amd.encoder import OctAttnFrameEncoder
from scp_amd.synth import synth_frame
level = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cyl = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
dev = torch.device('cuda:0')
model = fill_weights(OctAttention(octattn_cfg()), 0).to(dev)
enc = OctAttnFrameEncoder(model, 'kitti', level, spher=not cyl, cylin=cyl, device=dev)
for i in range(3):
    xyz = synth_frame(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    res = enc.encode(xyz)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"frame {i}: {dt*1e3:.1f} ms nodes {res['n_nodes']} bpp {res['bpp']:.3f}", flush=True)
# pipelined throughput: the host range coder of frame k runs under the GPU work of frame k + 1
frames = [synth_frame(i) for i in range(12)]
for f in frames[:2]: enc.finish(enc.encode_async(f))
torch.cuda.synchronize(); t = time.perf_counter()
prev = None
outs = []
for f in frames:
    h = enc.encode_async(f)
    if prev is not None: outs.append(enc.finish(prev))
    prev = h
outs.append(enc.finish(prev))
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"pipelined: {dt / len(frames) * 1e3:.1f} ms per frame ({len(frames) / dt:.2f} frames/s), bpp {outs[1]['bpp']:.3f}", flush=True)
