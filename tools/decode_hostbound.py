"""Is a one-window forward of the decoder host-bound?  Phase 1 and phase 2 of one window of n nodes: time until the launching call returns (host) against
time until the GPU is done (synchronised).  python tools/decode_hostbound.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd import ops
from scp_amd.models.packed import PackedPlan, ehem_phase1_packed, ehem_phase2_packed, ehem_phase2_prepare
dev = torch.device("cuda:0")
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
g = torch.Generator().manual_seed(0)
print("nodes | phase 1: host ms / GPU-complete ms | phase 2 (prepared): host ms / GPU-complete ms | launches")
for n in (1, 8, 91, 520, 2524, 8192):
    ctx = torch.randint(0, 9, (n, 12), generator=g).to(torch.uint8); ctx[:, 2::3] = torch.randint(0, 255, (n, 4), generator=g).to(torch.uint8); ctx[:, 0::3] = 7
    ctx, pos = ctx.to(dev), torch.rand((n, 3), generator=g).to(dev)
    plan = PackedPlan([n], device=dev)
    with ops.frozen_weights():
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            p1, st = ehem_phase1_packed(model, ctx, pos, plan)
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            if n > 1:
                prep = ehem_phase2_prepare(model, st, plan)
                torch.cuda.synchronize(); t3 = time.perf_counter()
                p2 = ehem_phase2_packed(model, st, plan, prep=prep)
                t4 = time.perf_counter(); torch.cuda.synchronize(); t5 = time.perf_counter()
            else:
                t3 = t4 = t5 = 0.0
        print(f"{n:6d} | {1e3 * (t1 - t0):6.2f} / {1e3 * (t2 - t0):6.2f} | {1e3 * (t4 - t3):6.2f} / {1e3 * (t5 - t3):6.2f}")
