#!/usr/bin/env python3
"""Stage G alone (transform -> quantiser -> sort -> octree -> context) of k frames per scp_geom_build_xyz, nothing else on the GPU:
python tools/run_geom_batch.py <frames per build> <builds> [level 12] [mullevel 0] [ford 0]
(under rocprofv3 for tools/stage_g_table.py; prints ms per build and the node count)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.synth import ford_like, synth_frame
k, n = int(sys.argv[1]), int(sys.argv[2])
level = int(sys.argv[3]) if len(sys.argv) > 3 else 12
mul = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
ford = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
dev = torch.device("cuda:0")
enc = FrameEncoder(fill_weights(EHEM(ehem_cfg()), 0).to(dev), "ford" if ford else "kitti", level, spher=True, mullevel=mul, device=dev)
frames = [torch.from_numpy(ford_like(synth_frame(i)) if ford else synth_frame(i)).to(dev) for i in range(k)]
for _ in range(2):
    (enc.preprocess_batch(frames) if k > 1 else enc.preprocess(frames[0]))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    (enc.preprocess_batch(frames) if k > 1 else enc.preprocess(frames[0]))
    torch.cuda.synchronize()
pre = enc.preprocess_batch(frames)[0] if k > 1 else enc.preprocess(frames[0])
print(f"nodes per build {pre['ctx'].shape[0]}")
print(f"{k} frame(s) per build: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per build, {(time.perf_counter() - t0) / n / k * 1e3:.3f} ms per frame (host wall, synchronised)")
