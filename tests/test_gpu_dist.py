"""The RCCL path on ONE GPU: `backend="nccl"` (= RCCL on ROCm) brought up with a world of one - library load, communicator
creation, device all-reduce / barrier / object gather - through scp_amd.distributed and through bench.py's own distributed steps.  A
one-GPU box cannot show a scaling curve; it can show that none of this code is dead."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               SCP_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


SCRIPT = r"""
import json, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from scp_amd import distributed as D
rank, world, local = D.init()                       # SCP_DIST_FORCE=1: a world of ONE on the nccl (RCCL) backend
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", local)
total = D.reduce_summary([1.5, 2.0, 0.25, 4.0, 3], dev)          # the end-of-run reduction on a DEVICE tensor: one RCCL all-reduce
big = torch.arange(1 << 20, dtype=torch.float32, device=dev)
dist.all_reduce(big)                                             # a payload that goes through the ring kernels, not only the 40-byte path
dist.barrier()
objs = [None]
dist.all_gather_object(objs, dict(rank=rank, device=torch.cuda.current_device()))
pinned = D.pin_rank_threads(local, 1)                            # enters the NUMA exchange (a collective) and returns: single rank -> no pinning
torch.cuda.synchronize()
print(json.dumps(dict(total=total, big_ok=bool((big == torch.arange(1 << 20, dtype=torch.float32, device=dev)).all().item()), objs=objs,
                      backend=dist.get_backend(), pinned=pinned)))
D.finalize()
assert not dist.is_initialized()
"""


def test_rccl_world_of_one_reduces_the_summary_on_the_device():
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["total"] == [1.5, 2.0, 0.25, 4.0, 3.0] and out["big_ok"] and out["objs"] == [{"rank": 0, "device": 0}]
    assert out["pinned"] is None


def test_bench_runs_its_distributed_steps_on_rccl_in_a_world_of_one():
    """bench.py under the driver's launcher with ONE rank and SCP_DIST_FORCE=1: init_process_group("nccl", device_id=...), the barriers around
    the timed region, the MAX all-reduce of the times, the per-rank gather with the shared-frame stream and the summary all-reduce all run on RCCL."""
    env = _env()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-strict-leg", "--no-legs", "--config", "ehem-L12-s"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["dist_backend"] == "nccl" and out["rccl_world"] == 1 and out["n_gpus"] == 1 and out["value"] > 0
    assert out["ranks"]["shared_frame_streams_identical"] and len(out["ranks"]["per_rank"]) == 1
