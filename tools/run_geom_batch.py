#!/usr/bin/env python3
"""Stage G alone (transform -> quantiser -> sort -> octree -> context) of k L12 same-level frames per scp_geom_build, nothing else on the
GPU: python tools/run_geom_batch.py <frames per build> <builds>   (under rocprofv3 for tools/stage_g_table.py; prints ms per build)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.synth import synth_frame
k, n = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
enc = FrameEncoder(fill_weights(EHEM(ehem_cfg()), 0).to(dev), "kitti", 12, spher=True, mullevel=False, device=dev)
frames = [torch.from_numpy(synth_frame(i)).to(dev) for i in range(k)]
for _ in range(2):
    (enc.preprocess_batch(frames) if k > 1 else enc.preprocess(frames[0]))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    (enc.preprocess_batch(frames) if k > 1 else enc.preprocess(frames[0]))
    torch.cuda.synchronize()
print(f"{k} frame(s) per build: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per build, {(time.perf_counter() - t0) / n / k * 1e3:.3f} ms per frame (host wall, synchronised)")
