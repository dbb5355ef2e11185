#!/usr/bin/env python3
"""A few launches of the row-chain LN + linear kernel (N = 768, 590 848 rows) for rocprofv3 counter passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device("cuda:0"); native.lib()
M = 590848
x = torch.randn((M, 256), device=dev)
W = torch.randn((768, 256), device=dev) * 0.05
fw = native.LnFoldedWeight(W, torch.ones(256, device=dev), torch.zeros(256, device=dev))
out = torch.empty((M, 768), device=dev)
for _ in range(3):
    native.swin_ln_linear(x, fw, None, out=out)
torch.cuda.synchronize()
