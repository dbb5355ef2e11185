import torch, sys
dev = torch.device('cuda:0')
def timeit(f, reps=10, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for mb in (256, 1024, 2420):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    t = timeit(lambda: a.fill_(1.0)); print(f"{mb} MB fill  {t:8.1f} us  {n*4/t/1e6:6.2f} TB/s")
    t = timeit(lambda: b.copy_(a)); print(f"{mb} MB copy  {t:8.1f} us  {2*n*4/t/1e6:6.2f} TB/s (r+w)")
    t = timeit(lambda: a.sum()); print(f"{mb} MB sum   {t:8.1f} us  {n*4/t/1e6:6.2f} TB/s")
    t = timeit(lambda: torch.add(a, 1.0, out=b)); print(f"{mb} MB add   {t:8.1f} us  {2*n*4/t/1e6:6.2f} TB/s (r+w)")
