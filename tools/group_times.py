import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder, EncodePlan
from scp_amd.synth import synth_frame
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
enc = FrameEncoder(model, 'kitti', 16, spher=True, mullevel=True, device=dev)
xyz = torch.from_numpy(synth_frame(0)).to(dev)
enc.encode(xyz)
pre = enc.preprocess(xyz)
plan = EncodePlan(pre['level_sizes'], 8192)
ctx, pos = pre['ctx'], pre['pos']
tot = 0
rows = []
for c, ws in plan.groups(8):
    starts = [w[0] for w in ws]
    bctx = torch.stack([ctx[s:s + c] for s in starts]); bpos = torch.stack([pos[s:s + c] for s in starts])
    torch.cuda.synchronize(); t = time.perf_counter()
    model.forward_ctx(bctx, bpos)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    rows.append((c, len(ws), dt)); tot += dt
big = sum(d for c, b, d in rows if c == 8192); mid = sum(d for c, b, d in rows if 1000 <= c < 8192); small = sum(d for c, b, d in rows if c < 1000)
print('groups', len(rows), 'total ms', round(tot, 1), ' full-8192 windows:', round(big, 1), ' 1000..8191:', round(mid, 1), ' <1000:', round(small, 1))
for c, b, d in rows: print(c, b, round(d, 2))
