"""N>1 path on CPU: frame sharding + the end-of-run summary all-reduce over gloo (world_size 2)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from scp_amd import distributed as D
    r, w, _ = D.init("gloo")
    files = [f"f{i}" for i in range(7)]
    mine = D.shard(files, r, w)
    # pretend every frame i costs bpp = i, time = 2i
    sums = [sum(i for i, _ in mine), 0.0, 0.0, sum(2 * i for i, _ in mine), len(mine)]
    total = D.reduce_summary(sums)
    # NUMA-aware pinning agrees between the ranks: each rank resolves only ITS device's node and the answers are exchanged; if any rank
    # cannot resolve its node (rank 1 here), every rank falls back to the plain split - never two schemes with overlapping cores
    D._gpu_numa_node = lambda i: 0
    torch.cuda.is_available = lambda: True
    torch.cuda.current_device = lambda: 0
    both = D._local_nodes(r, w)
    D._gpu_numa_node = (lambda i: None) if r == 1 else (lambda i: 0)
    one_missing = D._local_nodes(r, w)
    # conditions that differ between ranks must not split the ranks around the exchange: rank 1 runs with SCP_PIN=0 and
    # believes it is alone on its node; both calls return (no rank is left hanging in all_gather_object), then one more collective works
    D._gpu_numa_node = lambda i: 0
    if r == 1:
        os.environ["SCP_PIN"] = "0"
    os.sched_setaffinity = lambda pid, cpus: None
    pinned = D.pin_rank_threads(r, w if r == 0 else 1)
    os.environ.pop("SCP_PIN", None)
    after = D.reduce_summary([1, 0, 0, 0, 1])
    assert after[0] == 2.0 and (pinned is None or r == 0)
    q.put((r, [i for i, _ in mine], total, both, one_missing))
    D.finalize()


def test_frame_sharding_and_summary_reduction_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]          # round-robin, disjoint, complete
    for _, _, total, both, one_missing in res:
        assert total == [21.0, 0.0, 0.0, 42.0, 7.0]                        # both ranks see the global sums
        assert both == [0, 0] and one_missing is None                      # the NUMA view is agreed, or dropped by everyone
    from scp_amd import distributed as D
    m = D.summary_means(res[0][2])
    assert m["bpp"] == 3.0 and m["time"] == 6.0 and m["count"] == 7


def test_single_process_is_a_noop():
    from scp_amd import distributed as D
    assert D.reduce_summary([1, 2, 3, 4, 5]) == [1.0, 2.0, 3.0, 4.0, 5.0]
    assert D.shard(list("abc"), 0, 1) == [(0, "a"), (1, "b"), (2, "c")]


def test_rank_thread_pinning_splits_the_host_cores(monkeypatch):
    """Every rank of a node takes its own slice of the cores (launch thread, range-coder worker, reader); fewer than two cores per rank,
    a single rank or SCP_PIN=0 leave the affinity alone."""
    from scp_amd import distributed as D
    calls = []
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)), raising=False)
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: calls.append(list(cpus)), raising=False)
    monkeypatch.setattr(torch, "set_num_threads", lambda n: None)
    monkeypatch.setattr(D, "_core_groups", lambda cpus: [[c] for c in sorted(cpus)])
    monkeypatch.setattr(D, "_local_nodes", lambda local, lw: None)
    monkeypatch.setattr(D, "_set_affinity_all_threads", lambda cpus: os.sched_setaffinity(0, cpus))
    assert D.pin_rank_threads(0, 8) == [0, 1] and D.pin_rank_threads(7, 8) == [14, 15] and D.pin_rank_threads(1, 2) == list(range(8, 16))
    assert calls == [[0, 1], [14, 15], list(range(8, 16))]
    assert D.pin_rank_threads(0, 1) is None and D.pin_rank_threads(3, 16) is None
    # SMT siblings stay with one rank; ranks whose GPUs hang on one NUMA node split THAT node's cores
    monkeypatch.setattr(D, "_core_groups", lambda cpus: [[c, c + 8] for c in sorted(cpus) if c < 8 and c + 8 in cpus] or [[c] for c in sorted(cpus)])
    assert D.pin_rank_threads(1, 4) == [2, 3, 10, 11]
    monkeypatch.setattr(D, "_core_groups", lambda cpus: [[c] for c in sorted(cpus)])
    monkeypatch.setattr(D, "_local_nodes", lambda local, lw: [i // 2 for i in range(lw)])      # ranks 0,1 drive GPUs on node 0; 2,3 on node 1
    monkeypatch.setattr(D, "_node_cpus", lambda n: set(range(4 * n, 4 * n + 4)) | set(range(8 + 4 * n, 12 + 4 * n)))
    assert D.pin_rank_threads(2, 4) == [4, 5, 6, 7] and D.pin_rank_threads(1, 4) == [8, 9, 10, 11] and D.pin_rank_threads(3, 4) == [12, 13, 14, 15]
    monkeypatch.setattr(D, "_node_cpus", lambda n: None if n == 1 else set(range(8)))           # one node's cpulist unreadable: plain split for all
    assert D.pin_rank_threads(2, 4) == [8, 9, 10, 11] and D.pin_rank_threads(0, 4) == [0, 1, 2, 3]
    n_calls = len(calls)
    monkeypatch.setenv("SCP_PIN", "0")
    assert D.pin_rank_threads(0, 8) is None and len(calls) == n_calls


def test_cli_gpus_flag_spawns_one_rank_per_gpu(monkeypatch):
    """`encode.py --gpus N` outside torchrun starts N ranks through torch.distributed.run (as a child, before any GPU call) and exits
    with its code; flag combinations the reference cannot execute are refused."""
    import subprocess
    import sys
    from scp_amd import cli, native
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["encode.py"])
    with pytest.raises(SystemExit) as e:
        cli.main(["--test_files", "x.bin", "--type", "kitti", "--gpus", "4", "--spher"], mullevel=False)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-7:] == ["--test_files", "x.bin", "--type", "kitti", "--gpus", "4", "--spher"]       # the caller's own arguments, unchanged

    class A:
        spher_circle = False; level_wise = False; preproc_path = ""; metrics = False; type = "kitti"; spher = True; cylin = False; sequential = False
    cli.refuse_unsupported(A, "EHEM", False)                       # a supported combination passes
    for field, value, model, mul in (("spher_circle", True, "EHEM", False), ("level_wise", True, "OctAttention", False),
                                     ("sequential", True, "EHEM", False), ("preproc_path", "pp/", "OctAttention", True), ("type", "obj", "EHEM", True)):
        a = type("B", (A,), {field: value})
        with pytest.raises(native.ScpError):
            cli.refuse_unsupported(a, model, mul)
    a = type("B", (A,), {"level_wise": True})
    cli.refuse_unsupported(a, "OctAttention", True)                 # encode_mullevel.py has the fixed level-wise form


def test_bench_gpus_flag_spawns_ranks_before_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` outside torchrun starts N ranks itself (child torch.distributed.run, 127.0.0.1) and exits with the
    child's code; inside a torchrun environment (RANK set) it does not spawn again."""
    import importlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 7)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "4", "--warmup", "1"])
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: (_ for _ in ()).throw(AssertionError("GPU touched before the spawn")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == [os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1"]
