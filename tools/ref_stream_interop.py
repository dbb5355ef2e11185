"""Can this decoder read the REFERENCE driver's own streams (tests/golden/e2e_*.npz)?  No - and neither can the reference's CUDA build read its CPU
build's: an arithmetic decoder needs the encoder's integer CDFs bit for bit, and logits that agree to 1e-5 still round to different integers
somewhere in 100k symbols (first differing byte: 19 / 14).  What IS pinned: symbols, bit counts, CDF integers given PMFs, coder bytes given CDFs."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from cfgs import ehem_cfg
from conftest import golden
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.decoder import FrameDecoder
from oracle import scp_oracle as orc
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
for name, L, mul in (("e2e_ehem_spher_L12", 12, False), ("e2e_ehem_mul_spher_L14", 14, True)):
    z = golden(name)
    enc = FrameEncoder(model, "kitti", L, spher=True, mullevel=mul, device=dev)
    if mul:
        res = enc.encode(z["xyz"])
    else:
        _, bin_num, _, _, pt = orc.quantise(z["xyz"], 400 / (2 ** L - 1), "spher")
        res = enc.encode_ints([pt], bin_num, 0.0, len(z["xyz"]))
    ref = z["bytes"].tobytes()
    same = res["bytes"] == ref
    n = min(len(ref), len(res["bytes"]))
    first = next((i for i in range(n) if ref[i] != res["bytes"][i]), n)
    print(name, "our stream == reference stream:", same, "first differing byte", first, "of", len(ref))
    try:
        dec = FrameDecoder(model, L, mullevel=mul, polar=True, device=dev)
        shells = dec.decode(ref, int(str(z["fname"]).split("_")[-3]), z["dat"])
        nodes = enc.geom.nodes(("occ",))["occ"].cpu().numpy()
        ok = True
        for s, (codes, leaves) in enumerate(shells):
            info = enc.geom.info[s]
            want = nodes[info.node_base:info.node_base + info.n_nodes]
            got = torch.cat(codes).cpu().numpy()
            ok &= len(got) == len(want) and np.array_equal(got[:len(want) - (1 if mul else 0)], want[:len(want) - (1 if mul else 0)])
        print("   decoding the REFERENCE's stream with this decoder regenerates the octree:", ok)
    except Exception as e:
        print("   decode of the reference stream failed:", type(e).__name__, str(e)[:200])
