"""Where does a short-K split GEMM's time go?  SCP_GEMM_PROBE=1 python tools/mb_gemm_probe.py : full / no output traffic / cache-resident A / both,
on the frame's K = 256 shapes and OctAttention's K = 600 shapes (bf16 planes: the F16 instantiation shares the loop)."""
import os, sys, torch
os.environ["SCP_GEMM_PROBE"] = "1"
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
for (M, N, K, want) in [(590848, 1024, 256, "split"), (590848, 512, 1024, "split"), (303616, 256, 256, "split"), (262144, 1280, 600, "f32"), (262144, 600, 600, "f32")]:
    a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    sa, sw = native.split_rows(a), native.SplitWeight(w)
    line = f"M={M} N={N} K={K} out={want}:"
    for cfg in (1, 2, 3):
        ts = []
        for probe in (0, 0x20000, 0x10000, 0x30000):
            ts.append(timeit(lambda: native.linear_split(sa, sw, b, native.ACT_LEAKY, None, want=want, cfg=cfg | probe)))
        line += f" | cfg{cfg}: full {ts[0]:.0f} no-out {ts[1]:.0f} A-cached {ts[2]:.0f} both {ts[3]:.0f} us (MFMA floor {6.0*M*N*K/1.95e15*1e6:.0f})"
    print(line, flush=True)
