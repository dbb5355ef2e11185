import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder, EncodePlan
from scp_amd.synth import synth_frame
dev = torch.device('cuda:0')
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 16
xyz = torch.from_numpy(synth_frame(0)).to(dev)
ea = FrameEncoder(model, "kitti", level, spher=True, mullevel=True, device=dev, packed=True)
eb = FrameEncoder(model, "kitti", level, spher=True, mullevel=True, device=dev, packed=False)
pre = ea.preprocess(xyz)
plan = EncodePlan(pre["level_sizes"], 8192)
ta = ea.logits_in_coding_order(pre, plan)
ta2 = ea.logits_in_coding_order(pre, plan)
tb = eb.logits_in_coding_order(pre, plan)
print("packed run-to-run identical:", torch.equal(ta, ta2))
d = (ta - tb).abs().max(1)[0]
print("rows", ta.shape[0], "max |packed - per-window|", d.max().item(), "rows differing", int((d > 0).sum()))
# which windows differ
bad = (d > 0).nonzero().flatten().cpu().numpy()
if len(bad):
    starts = np.array([w[2] for w in plan.windows]); sizes = np.array([w[1] for w in plan.windows])
    win = np.searchsorted(starts, bad, side='right') - 1
    u, c = np.unique(win, return_counts=True)
    print("windows with differences:", [(int(a), int(sizes[a]), int(b)) for a, b in zip(u, c)][:20])
    print("max diff by window:", [(int(a), float(d[torch.from_numpy(bad[win == a]).to(dev)].max())) for a in u[:20]])
