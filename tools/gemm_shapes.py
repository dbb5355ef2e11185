"""Dense-layer launches of one L16 mullevel frame (packed forward), timed in place with HIP events and grouped by (N, K, act,
extras): python tools/gemm_shapes.py  ->  count, rows, ms per frame, bf16-equivalent TFLOP/s of each group"""
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
recs = []
def ev(): return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
o_lin, o_sc, o_mlp = native.linear_split, native.linear_split_scatter, native.mlp_split_fused
def lin(a, sw, bias=None, act=0, residual=None, out=None, out_split=None, want="f32", cfg=0, res_map=None, res_first=False):
    s, e = ev(); s.record(); y = o_lin(a, sw, bias, act, residual, out, out_split, want, cfg, res_map, res_first); e.record()
    recs.append((s, e, a.M, sw.N, sw.K, act, ("res" if residual is not None else "") + ("+map" if res_map is not None else "") + " " + want))
    return y
def sc(a, sw, bias, out_map, table, act=0, cfg=0):
    s, e = ev(); s.record(); o_sc(a, sw, bias, out_map, table, act, cfg); e.record()
    recs.append((s, e, a.M, sw.N, sw.K, act, "scatter"))
def mlp(a, s1, b1, s2, b2, residual=None):
    s, e = ev(); s.record(); y = o_mlp(a, s1, b1, s2, b2, residual); e.record()
    recs.append((s, e, a.M, 256, 2048, 2, "fused mlp (256-1024-256)"))
    return y
native.linear_split, native.linear_split_scatter, native.mlp_split_fused = lin, sc, mlp
enc.encode(xyz); torch.cuda.synchronize()
native.linear_split, native.linear_split_scatter, native.mlp_split_fused = o_lin, o_sc, o_mlp
g = collections.defaultdict(lambda: [0, 0, 0.0, 0.0])
for s, e, M, N, K, act, tag in recs:
    k = (N, K, act, tag); g[k][0] += 1; g[k][1] += M; g[k][2] += s.elapsed_time(e); g[k][3] += 2.0 * M * N * K
print(f"{'cnt':>3} {'rows':>9} {'N':>5} {'K':>5} act {'ms':>7} {'TF/s':>6}  extras")
tot = 0.0
for (N, K, act, tag), (c, rows, ms, fl) in sorted(g.items(), key=lambda kv: -kv[1][2]):
    tot += ms
    print(f"{c:3d} {rows:9d} {N:5d} {K:5d} {act:3d} {ms:7.2f} {fl / ms / 1e9:6.0f}  {tag}")
print(f"total {tot:.1f} ms")
