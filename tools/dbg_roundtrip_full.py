"""Full-size encode -> decode round trip timing/validation: python tools/dbg_roundtrip_full.py [level] [mul 0/1] [stride]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.decoder import FrameDecoder
from scp_amd.synth import synth_frame
level = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mul = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
xyz = synth_frame(0)[::stride].copy()
enc = FrameEncoder(model, 'kitti', level, spher=True, mullevel=mul, device=dev)
res = enc.encode(xyz)
nodes = enc.geom.nodes(("occ",))["occ"]
print("encoded", res["n_nodes"], "nodes", res["bits"], "bits", flush=True)
dec = FrameDecoder(model, level, mullevel=mul, polar=True, device=dev)
torch.cuda.synchronize(); t = time.perf_counter()
shells = dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
torch.cuda.synchronize(); print(f"decode {time.perf_counter()-t:.1f} s")
ok = True
for s, (codes, leaves) in enumerate(shells):
    info = enc.geom.info[s]
    want = nodes[info.node_base:info.node_base + info.n_nodes].cpu().numpy()
    got = torch.cat(codes).cpu().numpy()
    same = len(got) == len(want) and (np.array_equal(got[:-1], want[:-1]) if mul else np.array_equal(got, want))
    print("shell", s, "nodes", len(want), "identical", same)
    ok &= bool(same)
print("ROUNDTRIP", "OK" if ok else "FAILED")
