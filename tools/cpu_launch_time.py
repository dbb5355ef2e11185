import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.synth import synth_frame
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
enc = FrameEncoder(model, 'kitti', 16, spher=True, mullevel=True, device=dev)
frames = [torch.from_numpy(synth_frame(i)).to(dev) for i in range(6)]
enc.finish(enc.encode_async(frames[0]))
torch.cuda.synchronize()
t0 = time.perf_counter(); hs = []; ts = []
for f in frames:
    a = time.perf_counter(); hs.append(enc.encode_async(f)); ts.append(time.perf_counter() - a)
t1 = time.perf_counter()
for h in hs: enc.finish(h)
torch.cuda.synchronize(); t2 = time.perf_counter()
print("encode_async CPU time per frame (ms):", [round(x * 1e3, 1) for x in ts])
print(f"all enqueued after {1e3*(t1-t0):.1f} ms, all done after {1e3*(t2-t0):.1f} ms ({1e3*(t2-t0)/len(frames):.1f} ms/frame)")
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
h = enc.encode_async(frames[0])
pr.disable(); enc.finish(h)
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
print("=== second call while the GPU is busy ===")
torch.cuda.synchronize()
h0 = enc.encode_async(frames[1])
pr = cProfile.Profile(); pr.enable()
h1 = enc.encode_async(frames[2])
pr.disable(); enc.finish(h0); enc.finish(h1)
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
