"""Where a decode spends its time, level by level: nodes, windows, synchronised wall ms of phase 1 / phase 2 / the rest.  python tools/decode_levels.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.decoder import FrameDecoder
from scp_amd.synth import synth_frame
dev = torch.device("cuda:0")
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
enc = FrameEncoder(model, "kitti", 16, spher=True, mullevel=True, device=dev)
res = enc.encode(synth_frame(0))
dec = FrameDecoder(model, 16, mullevel=True, polar=True, device=dev)
dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
rows = []
orig = dec._decode_level
def wrapped(d, ctx, pos, n_total):
    dec.stats = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(d, ctx, pos, n_total)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows.append((ctx.shape[0], dict(dec.stats), dt))
    return out
dec._decode_level = wrapped
dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
print("nodes  windows  total ms   phase1   phase2   cdf_d2h  range_decoder")
tot = {}
for n, st, dt in rows:
    w = -(-n // 8192)
    print(f"{n:7d} {w:4d} {1e3 * dt:9.2f} {1e3 * st.get('phase1_model', 0):8.2f} {1e3 * st.get('phase2_model', 0):8.2f} {1e3 * st.get('cdf_d2h', 0):8.2f} {1e3 * st.get('range_decoder', 0):8.2f}")
    b = "n<=512" if n <= 512 else ("n<=8192" if n <= 8192 else "n>8192")
    t = tot.setdefault(b, [0, 0.0, 0.0, 0.0])
    t[0] += 1; t[1] += 1e3 * dt; t[2] += 1e3 * st.get('phase1_model', 0); t[3] += 1e3 * st.get('phase2_model', 0)
for b, t in tot.items():
    print(b, "levels", t[0], "total ms %.1f phase1 %.1f phase2 %.1f" % (t[1], t[2], t[3]))
