"""The frame's gemm_split launches by shape, with the launch brackets' times: python tools/gemm_split_shapes.py [cfg]
(cfg 1 / 2 / 3 forces the 256 x 256 / 256 x 128 / 128 x 128 tile on every launch that has such a variant: identical bits, different time)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfgs import ehem_cfg
from scp_amd import native
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.synth import synth_frame
dev = torch.device("cuda:0")
enc = FrameEncoder(fill_weights(EHEM(ehem_cfg()), 0).to(dev), "kitti", 16, spher=True, mullevel=True, device=dev)
xyz = torch.from_numpy(synth_frame(0)).to(dev)
enc.encode(xyz); torch.cuda.synchronize()
shapes = []
FORCE = int(sys.argv[1]) if len(sys.argv) > 1 else 0
o1, o2 = native.linear_split, native.linear_split_scatter
def w1(a, sw, bias=None, act=0, residual=None, out=None, out_split=None, want="f32", cfg=0, res_map=None, res_first=False):
    shapes.append((a.M, sw.N, sw.K, act, want, residual is not None, res_map is not None))
    ext = res_map is not None or res_first
    c = FORCE if FORCE and not (ext and FORCE == 3) else cfg
    return o1(a, sw, bias, act, residual, out, out_split, want, c, res_map, res_first)
def w2(a, sw, bias, out_map, table, act=0, cfg=0):
    shapes.append((a.M, sw.N, sw.K, act, "scatter", False, False))
    return o2(a, sw, bias, out_map, table, act, FORCE if FORCE in (1, 2) else cfg)
o3 = native.linear_split_hier2
def w3(a0, sw0, a1, sw1, parent, bias=None, act=0, residual=None, res_map=None, out_split=None):
    shapes.append((a0.M, sw0.N, sw0.K + 128, act, "hier2", residual is not None, res_map is not None))      # algorithmic K: stage 0 + stage 1 at half the rows
    return o3(a0, sw0, a1, sw1, parent, bias, act, residual, res_map, out_split)
native.linear_split, native.linear_split_scatter, native.linear_split_hier2 = w1, w2, w3
import scp_amd.ops as ops
best = None
for _ in range(3):
    shapes.clear()
    with native.launch_profile() as p:
        enc.encode(xyz); torch.cuda.synchronize()
    t = [r[1] for r in p.records() if r[0] == "gemm_split"]
    best = t if best is None else [min(a, b) for a, b in zip(best, t)]
assert len(best) == len(shapes), (len(best), len(shapes))
tot = 0
for (M, N, K, act, want, res, rmap), ms in zip(shapes, best):
    fl = 2.0 * M * N * K
    tot += ms
    print(f"M={M:7d} N={N:5d} K={K:5d} act={act} out={want:7s} res={int(res)} gather={int(rmap)}: {ms*1e3:7.1f} us  {fl/ms/1e9:6.1f} TF = {fl/ms/1e9/833.3:.3f}")
print(f"total {tot:.2f} ms in {len(best)} launches")
