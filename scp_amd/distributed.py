"""Multi-GPU plumbing of the encode path: one process per GPU, frames sharded round-robin, one tiny all-reduce.

Frames are independent units (encode.py:274-291 rebuilds every tensor per frame), so there is no data-path collective:
rank r encodes files r, r+R, r+2R, ... with its own model replica and writes its own output files.  The only exchange is the
end-of-run reduction of [sum bpp, sum PSNR, sum chamfer, sum time, count] (encode.py:293-305) - 40 bytes, one RCCL
all-reduce over xGMI (backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def pin_rank_threads(local, local_world):
    """Give every rank its own slice of the host cores (launch thread, range-coder worker, file reader): with 8 ranks the host
    side is the expected limiter, and threads of different ranks migrating over each other's cores is the first
    thing that hurts.  A rank needs ~3 cores; with fewer than 2 per rank the affinity is left alone.  SCP_PIN=0 disables."""
    if os.environ.get("SCP_PIN", "1") == "0" or local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // local_world
    if per < 2:
        return None
    mine = cpus[local * per:(local + 1) * per]
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(per, 4)))
    return mine


def shard(items, rank, world):
    """Round-robin frame sharding: item i belongs to rank i % world."""
    return [(i, it) for i, it in enumerate(items) if i % world == rank]


def reduce_summary(sums, device=None):
    """sums: [sum_bpp, sum_psnr, sum_chamfer, sum_time, count] of this rank -> the same five numbers over all ranks."""
    t = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().tolist()


def summary_means(total):
    n = max(total[4], 1.0)
    return dict(bpp=total[0] / n, psnr=total[1] / n, chamfer=total[2] / n, time=total[3] / n, count=int(total[4]))


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
