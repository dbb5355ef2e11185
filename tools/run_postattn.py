#!/usr/bin/env python3
"""A few launches of scp_swin_post_attn (590 848 rows) for rocprofv3 counter passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device("cuda:0"); native.lib()
M = 590848
g = torch.Generator().manual_seed(1)
rn = lambda *sh, s=1.0: (torch.randn(sh, generator=g) * s).to(dev)
x, ofull = rn(M, 256), rn(M, 256)
pw = native.PostAttnWeights(rn(256, 256, s=0.05), rn(256, s=0.1), 1 + rn(256, s=0.1), rn(256, s=0.1), rn(1024, 256, s=0.05), rn(1024, s=0.1),
                            rn(256, 1024, s=0.03), rn(256, s=0.1))
o = native.split_rows(ofull)
y = torch.empty_like(x)
for _ in range(3):
    native.swin_post_attn(o, x, pw, out=y)
torch.cuda.synchronize()
