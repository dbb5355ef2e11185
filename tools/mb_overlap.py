"""Can MFMA work and HBM streaming overlap on this chip?  Cache-resident GEMM probe on one stream, a pure fill on another."""
import os, sys, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device('cuda:0')
M, N, K = 590848, 1024, 256
a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev) / 16; b = torch.randn(N, device=dev)
sw = native.SplitWeight(w); sa = native.split_rows(a); out = torch.empty((M, N), device=dev)
big = torch.empty(605 * 1024 * 1024, device=dev)          # 2.4 GB
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemm(n):
    with torch.cuda.stream(s1):
        for _ in range(n): native.linear_split(sa, sw, b, 0, None, out=out, cfg=0x30001)
def fill(n):
    with torch.cuda.stream(s2):
        for _ in range(n): big.fill_(1.0)
def wall(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
gemm(2); fill(2); torch.cuda.synchronize()
tg = wall(lambda: gemm(10)); tf = wall(lambda: fill(20))
tb = wall(lambda: (gemm(10), fill(20)))
print(f"gemm probe x10: {tg:.2f} ms   fill 2.4GB x20: {tf:.2f} ms   both concurrently: {tb:.2f} ms   (sum {tg+tf:.2f}, max {max(tg,tf):.2f})")
