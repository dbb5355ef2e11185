import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd import ops, native
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'gemm'
if which == 'gemm':
    M, N, K = 65536, 768, 256
    a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev); b = torch.randn(N, device=dev)
    for _ in range(5): ops.linear(a, w, b)
elif which == 'knn':
    x = torch.randn((8, 8192, 192), device=dev)
    for _ in range(3): native.knn_topk(x, 20)
torch.cuda.synchronize()
