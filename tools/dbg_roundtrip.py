import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import ehem_cfg
from scp_amd import native
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder, EncodePlan
from scp_amd.decoder import FrameDecoder
from scp_amd.synth import synth_frame
from scp_amd.models.packed import PackedPlan, ehem_phase1_packed, ehem_phase2_packed
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
xyz = synth_frame(4)[::30].copy()
enc = FrameEncoder(model, 'kitti', 12, spher=True, mullevel=False, device=dev)
res = enc.encode(xyz)
pre = res['_debug']['pre']; table = res['_debug']['table']
plan = EncodePlan(pre['level_sizes'], 8192)
print('levels', pre['level_sizes'])
# per window: recompute logits with a single-window plan and compare with the encoder's table rows
for (start, c, coded) in plan.windows:
    ctx = pre['ctx'][start:start + c].clone(); pos = pre['pos'][start:start + c]
    pp = PackedPlan([c], device=dev)
    ev, st = ehem_phase1_packed(model, ctx, pos, pp)
    od = ehem_phase2_packed(model, st, pp)
    ne = (c + 1) // 2
    d1 = (ev - table[coded:coded + ne]).abs().max().item()
    d2 = (od - table[coded + ne:coded + c]).abs().max().item() if c > 1 else 0.0
    # with own occupancy hidden (what the decoder sees)
    ctx2 = ctx.clone(); ctx2[:, 11] = 255
    ev2, st2 = ehem_phase1_packed(model, ctx2, pos, pp)
    d3 = (ev2 - ev).abs().max().item()
    print(f'window start {start} c {c}: single-window vs frame-packed max diff even {d1:.3e} odd {d2:.3e}; own-occ hidden changes phase1 by {d3:.3e}')
