"""Per-kernel timings at the production shapes (HIP events on the launch stream)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device('cuda:0')

def timeit(f, reps=5, warm=2):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

which = sys.argv[1:] or ['knn', 'attn', 'gemm', 'cdf']
if 'knn' in which:
    for B, n in ((1, 8192), (8, 8192), (1, 1372)):
        for C in (3, 144, 192):
            g = torch.Generator(device='cpu').manual_seed(1)
            if C == 3:
                # Morton-ordered lattice-like positions: sort random points by a coarse key to mimic BFS order
                x = torch.rand((B, n, 3), generator=g)
                key = (x[..., 0] * 16).long() * 256 + (x[..., 1] * 16).long() * 16 + (x[..., 2] * 16).long()
                x = torch.stack([x[b][torch.argsort(key[b])] for b in range(B)])
            else:
                x = torch.randn((B, n, C), generator=g)
            x = x.to(dev)
            ms = timeit(lambda: native.knn_topk(x, 20))
            fl = 2.0 * B * n * n * C
            print(f"knn B={B} n={n} C={C}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.2f} TFLOP/s")
if 'attn' in which:
    tab = torch.randn((1023, 4), device=dev)
    for B, Lp in ((1, 8192), (8, 8192), (8, 4096), (8, 512), (1, 512)):
        qkv = torch.randn((B, Lp, 768), device=dev)
        q, k, v = qkv[..., :256], qkv[..., 256:512], qkv[..., 512:]
        for shift in (0, 256):
            ms = timeit(lambda: native.swin_attention(q, k, v, tab, shift))
            fl = B * (Lp // 512) * 4 * 2 * 2.0 * 512 * 512 * 64
            print(f"attn B={B} Lp={Lp} shift={shift}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.2f} TFLOP/s")
if 'gemm' in which:
    import torch.nn.functional as F
    for M, N, K in ((65536, 768, 256), (65536, 256, 256), (65536, 1024, 256), (65536, 256, 1024), (65536, 1024, 1280),
                    (8192, 768, 256), (8192, 1024, 256), (8192, 256, 1024), (65536, 512, 192), (65536, 255, 256)):
        a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev); b = torch.randn(N, device=dev)
        ms = timeit(lambda: F.linear(a, w, b))
        print(f"torch linear M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.2f} TFLOP/s")
    from scp_amd import ops
    for M, N, K in ((65536, 768, 256), (65536, 256, 256), (65536, 1024, 256), (65536, 256, 1024), (65536, 1024, 1280), (8192, 768, 256), (512, 768, 256)):
        a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev); b = torch.randn(N, device=dev); r = torch.randn((M, N), device=dev)
        ms = timeit(lambda: ops.linear(a, w, b, act="gelu", residual=r))
        print(f"bf16x3 linear+gelu+res M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.2f} TFLOP/s-equivalent")
    x = torch.randn((65536, 256), device=dev); ln_w = torch.ones(256, device=dev); ln_b = torch.zeros(256, device=dev)
    print(f"layer_norm 65536x256: {timeit(lambda: F.layer_norm(x, (256,), ln_w, ln_b))*1e3:.1f} us")
    y = torch.randn((65536, 1024), device=dev)
    print(f"gelu 65536x1024: {timeit(lambda: F.gelu(y))*1e3:.1f} us;  add 65536x256: {timeit(lambda: x + x)*1e3:.1f} us")
if 'cdf' in which:
    n = 577000
    logits = torch.randn((n, 255), device=dev); sym = torch.randint(0, 255, (n,), device=dev).to(torch.uint8)
    ms = timeit(lambda: native.softmax_cdf(logits, sym))
    print(f"softmax_cdf n={n}: {ms*1e3:.1f} us  {n*1024/ms/1e6:.1f} GB/s")
