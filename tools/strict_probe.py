import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.synth import synth_frame
dev = torch.device("cuda:0")
enc = FrameEncoder(fill_weights(EHEM(ehem_cfg()), 0).to(dev), "kitti", 16, spher=True, mullevel=True, device=dev)
xyz_h = synth_frame(0); xyz = torch.from_numpy(xyz_h).to(dev)
for _ in range(3): enc.preprocess(xyz); torch.cuda.synchronize()
def t(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("preprocess (device transform): %.3f ms" % t(lambda: enc.preprocess(xyz)))
ints = enc.host_ints(xyz_h)
for _ in range(3): enc.preprocess(xyz, ints); torch.cuda.synchronize()
print("preprocess (host ints given):  %.3f ms" % t(lambda: enc.preprocess(xyz, ints)))
print("host_ints:                      %.3f ms" % t(lambda: enc.host_ints(xyz_h)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): enc.preprocess(xyz, ints); torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
