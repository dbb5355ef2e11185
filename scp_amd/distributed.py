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


def init(backend=None, force=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process, unless `force` / SCP_DIST_FORCE=1: then a
    world of ONE is initialised too - RCCL library load, communicator creation and the device all-reduce path can be exercised on a
    one-GPU box, tests/test_gpu_dist.py)."""
    rank, world, local = env_rank()
    if force is None:
        force = os.environ.get("SCP_DIST_FORCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def _core_groups(cpus):
    """The logical CPUs of `cpus` grouped by physical core (SMT siblings together), in core order; every CPU its own group when the
    topology files are not readable."""
    seen, groups = set(), []
    for c in sorted(cpus):
        if c in seen:
            continue
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                sib = [x for x in _parse_cpulist(f.read()) if x in cpus]
        except (OSError, ValueError):
            sib = [c]
        sib = [x for x in (sib or [c]) if x not in seen] or [c]
        seen.update(sib)
        groups.append(sorted(sib))
    return groups


def _gpu_numa_node(index):
    """NUMA node of HIP device `index` (PCI address -> sysfs), or None.  Needs the device properties, i.e. an initialised GPU."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            n = int(f.read())
        return n if n >= 0 else None
    except Exception:
        return None


def _node_cpus(node):
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            return set(_parse_cpulist(f.read()))
    except (OSError, ValueError):
        return None


def _set_affinity_all_threads(cpus):
    """sched_setaffinity(0) moves only the calling thread (new threads inherit it); the HIP runtime and OpenMP threads that already
    exist are moved one by one."""
    os.sched_setaffinity(0, cpus)
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except (OSError, ValueError):
                pass
    except OSError:
        pass


def pin_rank_threads(local, local_world, numa=True):
    """Give every rank of a node its own PHYSICAL cores (launch thread, range-coder worker, file reader: ~3 busy threads), taken
    from the NUMA node its GPU hangs on when sysfs tells (ranks whose GPUs share a node split that node's cores; SMT siblings stay
    with one rank).  Threads that already exist are moved too.  Nothing is changed for a single rank, with fewer than two cores per
    rank, or with SCP_PIN=0.  Returns the CPU list of this rank or None.
    The NUMA exchange is a WORLD collective: every rank of an initialised process group enters it FIRST, before any condition that can
    differ between ranks or nodes (SCP_PIN, a node that runs a single rank, a missing sched_setaffinity, a failing sysfs read) - a rank that
    skipped it would leave all the others hanging in it."""
    nodes = _local_nodes(local, local_world) if numa else None
    if os.environ.get("SCP_PIN", "1") == "0" or local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        return _pin_rank_threads(local, local_world, nodes)
    except Exception:               # pinning is an optimisation: an unexpected sysfs layout or a refused affinity call never costs a rank
        return None


def _local_nodes(local, local_world):
    """NUMA node of the device EVERY local rank drives, agreed between the ranks: each rank looks up only the device it actually uses
    (torch.cuda.current_device(): right under per-rank HIP_VISIBLE_DEVICES and under SCP_FORCE_DEVICE alike) and the ranks exchange the
    answers (all_gather_object over the initialised process group).  None when there is no process group to agree through or when ANY
    rank could not resolve its node - then every rank takes the plain split, so two schemes can never hand out overlapping cores."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return None
    try:                            # nothing in front of the collective may raise on one rank only
        node = _gpu_numa_node(torch.cuda.current_device()) if torch.cuda.is_available() else None
    except Exception:
        node = None
    mine = (int(os.environ.get("GROUP_RANK", "0")), int(local), node)
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    here = sorted((l, n) for g, l, n in everyone if g == mine[0])
    if len(here) != local_world or [l for l, _ in here] != list(range(local_world)) or any(n is None for _, n in here):
        return None
    return [n for _, n in here]


def _pin_rank_threads(local, local_world, nodes):
    allowed = set(os.sched_getaffinity(0))
    peers, slot, pool = local_world, local, allowed
    if nodes is not None:
        mine_node = nodes[local]
        cpus_of = {n: _node_cpus(n) for n in set(nodes)}
        # the NUMA split is used only if it works for EVERY node involved (every rank evaluates the same condition on the same data)
        if all(c and (c & allowed) for c in cpus_of.values()):
            same = [i for i, n in enumerate(nodes) if n == mine_node]
            peers, slot, pool = len(same), same.index(local), cpus_of[mine_node] & allowed
    cores = _core_groups(pool)
    per = len(cores) // peers
    if per < 2:
        return None
    mine = sorted(c for g in cores[slot * per:(slot + 1) * per] for c in g)
    _set_affinity_all_threads(mine)
    torch.set_num_threads(max(1, min(per, 4)))
    return mine


def shard(items, rank, world):
    """Round-robin frame sharding: item i belongs to rank i % world."""
    return [(i, it) for i, it in enumerate(items) if i % world == rank]


def reduce_summary(sums, device=None):
    """sums: [sum_bpp, sum_psnr, sum_chamfer, sum_time, count] of this rank -> the same five numbers over all ranks."""
    t = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():          # (a world of one - init(force=True) - runs the collective too)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().tolist()


def summary_means(total):
    n = max(total[4], 1.0)
    return dict(bpp=total[0] / n, psnr=total[1] / n, chamfer=total[2] / n, time=total[3] / n, count=int(total[4]))


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
