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
for (M, N, K, act) in [(61440, 600, 600, 0), (122880, 600, 600, 0), (122880, 300, 600, 3), (122880, 600, 300, 0), (30720, 600, 600, 0), (245760, 600, 600, 0)]:
    a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    sw16 = native.SplitWeightF16(w); sw = native.SplitWeight(w)
    t16 = timeit(lambda: native.linear_f16x3(a, sw16, b, act))
    rs = native.RowScales(a)
    t16s = timeit(lambda: native.linear_f16x3(a, sw16, b, act, scales=rs))
    sa = native.split_rows(a)
    tsp = timeit(lambda: native.SplitActF16(a))
    pa = native.SplitActF16(a)
    tpl = timeit(lambda: native.linear_split_f16(pa, sw16, b, act))
    trs = timeit(lambda: native.RowScales(a))
    print(f"   planes: split_rows_f16 {tsp:.0f} us (row scales alone {trs:.0f} us), linear_split_f16 {tpl:.0f} us ({6.0*M*N*K/tpl/1e6:.0f} TF)")
    line = f"M={M} N={N} K={K}: f16x3 {t16:.0f} us ({6.0*M*N*K/t16/1e6:.0f} TF)  given scales {t16s:.0f} us ({6.0*M*N*K/t16s/1e6:.0f} TF)"
    for cfg in (1, 2, 3):
        out = torch.empty((M, N), device=dev)
        t = timeit(lambda: native.linear_split(sa, sw, b, act, None, out=out, cfg=cfg))
        line += f" | split cfg{cfg} {t:.0f} us ({6.0*M*N*K/t/1e6:.0f} TF)"
    print(line, flush=True)
