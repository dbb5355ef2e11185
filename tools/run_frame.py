"""Encode synthetic frames end to end (for profiling): python tools/run_frame.py [level] [mul 0/1] [frames]"""
import sys, time, json
import numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.synth import synth_frame
level = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mul = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
enc = FrameEncoder(model, 'kitti', level, spher=True, mullevel=mul, device=dev)
for i in range(nf):
    xyz = torch.from_numpy(synth_frame(i)).to(dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    res = enc.encode(xyz, timing=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"frame {i}: {dt*1e3:.1f} ms  nodes {res['n_nodes']} bpp {res['bpp']:.3f} windows {len(__import__('scp_amd.encoder').encoder.EncodePlan(res['level_sizes'], 8192).windows)} times",
          {k: round(v*1e3, 1) for k, v in res['times'].items()}, flush=True)
print('max mem GB', torch.cuda.max_memory_allocated() / 2**30)
