"""Do small kernels on a side stream get scheduled while a long queue of big kernels runs on another stream?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device('cuda:0')
M, N, K = 590848, 1024, 256
a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev) / 16; b = torch.randn(N, device=dev)
sw = native.SplitWeight(w); sa = native.split_rows(a); out = torch.empty((M, N), device=dev)
small = torch.zeros(1024, device=dev)
def big(n):
    for _ in range(n): native.linear_split(sa, sw, b, 0, None, out=out)
def probe(side, label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    big(40)                                   # ~60 ms of GPU work queued on the current stream
    t1 = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(20): small.add_(1.0)
        v = small[0].item()                   # sync of the side stream only
    t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"{label}: enqueue big {1e3*(t1-t0):.1f} ms; side-stream small work + sync {1e3*(t2-t1):.1f} ms; total {1e3*(t3-t0):.1f} ms")
big(3); torch.cuda.synchronize()
probe(torch.cuda.Stream(), "main = default stream, side = pool stream")
probe(torch.cuda.Stream(priority=-1), "main = default stream, side = high-priority stream")
m2 = torch.cuda.Stream()
with torch.cuda.stream(m2):
    probe(torch.cuda.Stream(), "main = pool stream, side = pool stream")
    probe(torch.cuda.Stream(priority=-1), "main = pool stream, side = high-priority stream")
