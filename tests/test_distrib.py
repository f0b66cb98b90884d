"""The N > 1 path of bench.py on CPU: two gloo processes run the replica timing protocol (barrier, max time, summed
work). The compute itself needs a GPU and is covered by -m gpu tests; this checks the distributed plumbing."""
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
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from soilmachine_amd import distrib
    r, lr, w = distrib.env_rank()
    dist = distrib.init(w, lr, use_cuda=False)
    dist.barrier()
    t, work = distrib.aggregate(dist, seconds=1.0 + rank, work=1000.0 * (rank + 1))
    q.put((rank, t, work, distrib.replica_seed(r)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_replicas_aggregate_like_bench():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [o[0] for o in out] == [0, 1]
    for _, t, work, _ in out:
        assert t == 2.0 and work == 3000.0            # max over ranks, sum over ranks -- identical on every rank
    assert [o[3] for o in out] == [0, 1]              # replica r simulates SEED r


def test_single_process_is_passthrough():
    from soilmachine_amd import distrib
    assert distrib.init(1, 0, use_cuda=False) is None
    assert distrib.aggregate(None, 1.5, 7.0) == (1.5, 7.0)
