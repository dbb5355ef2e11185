"""(set SCP_GEMM_PROBE=1 for the cfg probe bits 0x10000 / 0x20000)  A/B the split-operand GEMM against the fp32-activation bf16x3 GEMM on the frame's main shapes: bitwise equality + time."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
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
shapes = [(590848, 1024, 256, 2, 0), (590848, 256, 1024, 0, 1), (590848, 768, 256, 0, 0), (590848, 256, 256, 0, 1), (590848, 1024, 1280, 1, 0),
          (303616, 1024, 256, 2, 0), (303616, 256, 512, 0, 0), (51200, 256, 256, 0, 1), (303616, 255, 512, 0, 0), (303616, 240, 240, 0, 0),
          (590848, 128, 128, 1, 0), (1000, 256, 256, 0, 1), (257, 300, 64, 3, 0)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
CFGS = [int(c, 0) for c in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 2)
for (M, N, K, act, res) in shapes:
    torch.manual_seed(M + N + K)
    a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    r = torch.randn((M, N), device=dev) if res else None
    sw = native.SplitWeight(w)
    ref = torch.empty((M, N), device=dev)
    t_old = timeit(lambda: native.linear_bf16x3(a, sw, b, act, r, ref))
    sa = native.split_rows(a)
    assert torch.equal(sa.float()[:, :K], (a.bfloat16().float() + (a - a.bfloat16().float()).bfloat16().float()))
    line = f"M={M:7d} N={N:5d} K={K:5d} act={act} res={res}  old {t_old:8.1f} us ({6.0*M*N*K/t_old/1e6:6.0f} TF)"
    for cfg in CFGS:
        out = torch.full((M, N), float('nan'), device=dev)
        t_new = timeit(lambda: native.linear_split(sa, sw, b, act, r, out=out, cfg=cfg))
        ok = torch.equal(out, ref)
        if cfg > 0xffff:
            line += f' | cfg{cfg:#x} {t_new:8.1f} us ({6.0*M*N*K/t_new/1e6:6.0f} TF)'; continue
        o2 = native.linear_split(sa, sw, b, act, r, want="split", cfg=cfg)
        ok2 = torch.equal(o2.t[0, :, :N].float() + o2.t[1, :, :N].float(), native.split_rows(ref).float())
        padz = bool((o2.t[:, :, N:] == 0).all())
        t_sp = timeit(lambda: native.linear_split(sa, sw, b, act, r, want="split", out_split=o2, cfg=cfg))
        line += f" | cfg{cfg:#x} {t_new:8.1f} us ({6.0*M*N*K/t_new/1e6:6.0f} TF) split-out {t_sp:8.1f} eq={ok},{ok2},{padz}" + ("" if ok else f" maxdiff {(out-ref).abs().max().item():.3e}")
    print(line, flush=True)
    del a, w, b, r, ref, out, sa
