import os, sys, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.decoder import FrameDecoder
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
cases = {"one": np.array([[10.0, 3.0, -1.0]], np.float32),
         "dup2": np.array([[10.0, 3.0, -1.0]] * 2, np.float32),
         "five": np.array([[10, 3, -1], [12, -4, 0.5], [30, 1, 2], [5, 5, -1.5], [60, -20, 1]], np.float32),
         "line": np.stack([np.linspace(2, 80, 300), np.zeros(300), np.zeros(300)], 1).astype(np.float32)}
for mul in (False, True):
    for name, xyz in cases.items():
        for mode in ("spher", "cart"):
            try:
                enc = FrameEncoder(model, "kitti", 12, spher=(mode == "spher"), mullevel=mul, device=dev)
                res = enc.encode(xyz)
                occ = enc.geom.nodes(("occ",))["occ"].cpu().numpy()
                dec = FrameDecoder(model, 12, mullevel=mul, polar=(mode == "spher"), device=dev)
                shells = dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
                ok = True
                for s, (codes, _) in enumerate(shells):
                    info = enc.geom.info[s]
                    want = occ[info.node_base:info.node_base + info.n_nodes]
                    got = torch.cat(codes).cpu().numpy() if len(codes) else np.zeros(0)
                    ok &= len(got) == len(want) and (np.array_equal(got[:-1], want[:-1]) if mul else np.array_equal(got, want))
                print(f"mul={mul} {name:5s} {mode}: nodes {res['n_nodes']} bits {res['bits']} roundtrip {ok}", flush=True)
            except Exception as e:
                print(f"mul={mul} {name:5s} {mode}: EXC {type(e).__name__}: {str(e)[:150]}", flush=True)
