"""Short packed kNN launches: the split sweep (workgroup shape 256, round 5) against the single sweep (SCP_KNN_SPLIT=0 in a second process).
python tools/mb_knn_split.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from scp_amd import native
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for lengths in ([512], [2048], [8192], [8192, 8192, 4000], [8192] * 6, [8192] * 15):
    for C in (144, 192):
        rows = sum(-(-n // 512) * 512 for n in lengths)
        x = torch.randn((rows, C), generator=g).to(dev)
        tab, base = [], 0
        for n in lengths:
            tab += [[base, n]] * (-(-n // 512)); base += -(-n // 512) * 512
        td = torch.tensor(tab, dtype=torch.int32, device=dev)
        for _ in range(3): native.knn_topk_packed(x, td)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): native.knn_topk_packed(x, td)
        e.record(); torch.cuda.synchronize()
        print(f"SPLIT={os.environ.get('SCP_KNN_SPLIT', '1')} windows {lengths if len(lengths) < 4 else str(len(lengths)) + ' x 8192'} C={C}: {1e3 * s.elapsed_time(e) / 20:.1f} us per search (incl. split2 + schedule + merge launches)")
