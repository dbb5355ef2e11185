"""Swin MLP: fused kernel vs fc1 (GELU, split out) + fc2 (residual): python tools/mb_mlp_fused.py [M]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for M in ([int(a) for a in sys.argv[1:]] or [577536, 70001, 300]):
    x = torch.randn((M, 256), generator=g).to(dev)
    w1 = (torch.randn((1024, 256), generator=g) / 16).to(dev); b1 = torch.randn(1024, generator=g).to(dev)
    w2 = (torch.randn((256, 1024), generator=g) / 32).to(dev); b2 = torch.randn(256, generator=g).to(dev)
    r = torch.randn((M, 256), generator=g).to(dev)
    a = native.split_rows(x); s1 = native.SplitWeight(w1); s2 = native.SplitWeight(w2)
    def two():
        hid = native.linear_split(a, s1, b1, act=native.ACT_GELU, want="split")
        return native.linear_split(hid, s2, b2, residual=r)
    def one():
        return native.mlp_split_fused(a, s1, b1, s2, b2, r)
    c2 = two(); c1 = one()
    print(f"M={M}: equal {torch.equal(c1, c2)}  max diff {(c1 - c2).abs().max().item():.3e}", flush=True)
    t2, t1 = timeit(two), timeit(one)
    fl = 2.0 * M * 256 * 1024 * 2
    print(f"M={M}: fc1 + fc2 {t2:.3f} ms ({fl/t2/1e9:.0f} TFLOP/s), fused {t1:.3f} ms ({fl/t1/1e9:.0f} TFLOP/s)", flush=True)
    if M > 100000:   # cycle stamps of the diagnostic build
        buf = torch.zeros((256 * 8, 8), dtype=torch.int64, device=dev)
        native.lib().scp_mlp_debug_buffer(buf.data_ptr())
        one(); torch.cuda.synchronize()
        native.lib().scp_mlp_debug_buffer(None)
        b = buf[buf[:, 5] > 0].double()
        per = b[:, :5] / b[:, 5:6]
        names = ["barrier waits", "phase-1 products", "GELU + split", "phase-2 products", "epilogue"]
        print("cycles per 128-row tile and wave: " + "  ".join(f"{n} {per[:, i].mean().item():.0f}" for i, n in enumerate(names)) +
              f"  total {per.sum(1).mean().item():.0f}  (MFMA pipe time of a wave: 64 steps x 24 x 32 = 49152)", flush=True)
