"""scp_linear_split_f16 tile configurations on OctAttention's shapes: python tools/mb_oa_cfg.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scp_amd import native
dev = torch.device('cuda:0')
def timeit(f, reps=10, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for (M, N, K, act) in [(262144, 600, 600, 0), (262144, 1200, 600, 0), (131072, 600, 600, 0), (262144, 300, 600, 3), (262144, 600, 300, 0), (131072, 255, 600, 0), (600000, 1200, 600, 0)]:
    a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    sw16 = native.SplitWeightF16(w)
    pa = native.SplitActF16(a)
    out = torch.empty((M, N), device=dev)
    ref = None
    line = f"M={M} N={N} K={K}:"
    for cfg in (3, 2, 1):
        t = timeit(lambda: native.linear_split_f16(pa, sw16, b, act, cfg=cfg, out=out))
        same = "" if ref is None else (" same" if torch.equal(ref, out) else " DIFF")
        if ref is None: ref = out.clone()
        line += f" | cfg{cfg} {t:.0f} us ({2.0*M*N*K/t/1e6:.0f} TF = {2.0*M*N*K/t/1e6/833.3:.3f}){same}"
    t = timeit(lambda: native.linear_f16x3(a, sw16, b, act))
    line += f" | rows kernel {t:.0f} us ({2.0*M*N*K/t/1e6/833.3:.3f})"
    print(line, flush=True)
