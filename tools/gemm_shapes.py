"""Record the dense-layer shape mix of one L16 mullevel frame and time every distinct shape in isolation.
python tools/gemm_shapes.py  ->  table: count, M, N, K, act, residual, us/launch, bf16-equivalent TFLOP/s (6MNK/t), GB/s (algorithmic)"""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
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
enc.encode(xyz)
shapes = collections.Counter()
orig = native.linear_bf16x3
def rec(x, sw, bias=None, act=0, residual=None, out=None):
    M = x.numel() // sw.K
    shapes[(M, sw.N, sw.K, act, residual is not None)] += 1
    return orig(x, sw, bias, act, residual, out)
native.linear_bf16x3 = rec
import scp_amd.ops as ops
enc.encode(xyz)
native.linear_bf16x3 = orig
tot = 0.0
print(f"{'cnt':>3} {'M':>8} {'N':>5} {'K':>5} act res {'us':>9} {'TF/s(bf16eq)':>12} {'GB/s':>8} {'ms/frame':>8}")
for (M, N, K, act, res), c in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
    a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    r = torch.randn((M, N), device=dev) if res else None
    sw = native.SplitWeight(w)
    out = torch.empty((M, N), device=dev)
    f = lambda: native.linear_bf16x3(a, sw, b, act, r, out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    by = 4.0 * M * (K + N + (N if res else 0))
    print(f"{c:3d} {M:8d} {N:5d} {K:5d} {act:3d} {int(res):3d} {us:9.1f} {6.0 * M * N * K / us / 1e6:12.1f} {by / us / 1e3:8.0f} {c * us / 1e3:8.2f}")
    tot += c * us / 1e3
    del a, w, b, r, out
print("sum ms/frame", round(tot, 2))
