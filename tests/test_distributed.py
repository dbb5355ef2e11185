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
    q.put((r, [i for i, _ in mine], total))
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
    for _, _, total in res:
        assert total == [21.0, 0.0, 0.0, 42.0, 7.0]                        # both ranks see the global sums
    from scp_amd import distributed as D
    m = D.summary_means(res[0][2])
    assert m["bpp"] == 3.0 and m["time"] == 6.0 and m["count"] == 7


def test_single_process_is_a_noop():
    from scp_amd import distributed as D
    assert D.reduce_summary([1, 2, 3, 4, 5]) == [1.0, 2.0, 3.0, 4.0, 5.0]
    assert D.shard(list("abc"), 0, 1) == [(0, "a"), (1, "b"), (2, "c")]
