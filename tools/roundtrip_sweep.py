"""Randomised encode -> decode sweep over modes / levels / point counts (robustness probe)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.decoder import FrameDecoder
from scp_amd.synth import synth_frame, ford_like
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
rng = np.random.default_rng(1)
bad = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    mode = ["spher", "cylin", "cart"][trial % 3]
    mul = bool((trial // 3) % 2) and mode != "cart"
    dtype = "ford" if trial % 5 == 4 else "kitti"
    level = int(rng.choice([9, 11, 12, 13, 15] if dtype == "kitti" else [13, 15, 16]))
    n = int(rng.integers(50, 30000)) | 1
    f = synth_frame(int(rng.integers(0, 1000)))
    xyz = f[rng.choice(len(f), n, replace=False)]
    if dtype == "ford": xyz = ford_like(xyz)
    try:
        enc = FrameEncoder(model, dtype, level, spher=(mode == "spher"), cylin=(mode == "cylin"), mullevel=mul, device=dev)
        res = enc.finish(enc.encode_async(xyz)) if trial % 2 else enc.encode(xyz)
        occ = enc.geom.nodes(("occ",))["occ"].cpu().numpy()
        shells = FrameDecoder(model, level, mullevel=mul, polar=(mode != "cart"), device=dev).decode(res["bytes"], res["n_levels"], res["pos_mm"])
        ok = True
        for s, (codes, _) in enumerate(shells):
            info = enc.geom.info[s]
            want = occ[info.node_base:info.node_base + info.n_nodes]
            got = torch.cat(codes).cpu().numpy()
            ok &= len(got) == len(want) and (np.array_equal(got[:-1], want[:-1]) if mul else np.array_equal(got, want))
        msg = "ok" if ok else "MISMATCH"
    except Exception as e:
        ok, msg = False, f"EXC {type(e).__name__}: {str(e)[:120]}"
    bad += not ok
    print(f"{trial:3d} {dtype} {mode:5s} mul={int(mul)} L{level} n={n}: nodes {res['n_nodes'] if ok or 'res' in dir() else '-'} {msg}", flush=True)
print("FAILURES:", bad)
