"""Multi-process plumbing for bench.py (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The timing protocol -- barrier, max-over-ranks wall time, sum-over-ranks work -- is the same for both multi-GPU modes:
column strips of ONE map (batched engine, soilmachine_amd/strips.py: the data-path exchange lives there) and
independent replicas (the exact engines execute the reference's sequential semantics, which do not shard).
SMX_DIST_BACKEND=gloo forces the CPU transport (e.g. two ranks sharing one GPU, which RCCL refuses)."""
from __future__ import annotations

import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(world: int, local_rank: int, use_cuda: bool):
    """Returns the torch.distributed module (initialised) or None for a single process. SMX_BENCH_FORCE_DIST=1 (a test switch) initialises the process
    group for ONE rank too, so that bench.py's multi-rank path -- backend nccl, device tensors, the strip driver over RCCL -- can run on a single device."""
    if world <= 1 and os.environ.get("SMX_BENCH_FORCE_DIST", "0") != "1":
        return None
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    import torch
    import torch.distributed as dist
    if use_cuda:
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group(backend=os.environ.get("SMX_DIST_BACKEND", "nccl" if use_cuda else "gloo"))
    return dist


def aggregate(dist, seconds: float, work: float, device=None):
    """(max seconds over ranks, total work over ranks): value = total work / max time is the whole-job rate."""
    if dist is None:
        return seconds, work
    import torch
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    w = torch.tensor([float(work)], dtype=torch.float64, device=dev)
    dist.all_reduce(w, op=dist.ReduceOp.SUM)
    return float(t.item()), float(w.item())


def replica_seed(rank: int) -> int:
    """Replica r simulates SEED r (rank 0 is BASELINE.json's SEED 0)."""
    return rank
