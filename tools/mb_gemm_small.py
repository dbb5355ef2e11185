import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd import ops
dev = torch.device('cuda:0')
def timeit(f, reps=20, warm=5):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for M in (128, 512, 4096):
    for K in (32, 256, 1024, 2048):
        a = torch.randn((M, K), device=dev); w = torch.randn((256, K), device=dev); b = torch.randn(256, device=dev)
        t = timeit(lambda: ops.linear(a, w, b))
        import torch.nn.functional as F
        t2 = timeit(lambda: F.linear(a, w, b))
        print(f"M={M} N=256 K={K}: bf16x3 {t:7.1f} us   torch {t2:7.1f} us")
x = torch.randn((128, 256), device=dev)
print("empty-ish kernel (add):", timeit(lambda: x + 1.0))
