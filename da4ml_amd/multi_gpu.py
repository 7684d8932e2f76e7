"""One process per GPU: sharding of independent CMVM solves over the ranks of a ``torch.distributed`` job.

The path shards at the level of independent units (matrices, decompose_dc candidates, model layers): every rank solves
its contiguous shard with no data-path collective.  Collectives are used only for the exchange
steps that really exist: the max-over-ranks timing of the benchmark, the arg-min over candidate costs, and gathering
results to rank 0.  Backend ``nccl`` (= RCCL over xGMI on ROCm) when a GPU is present, ``gloo`` on CPU-only hosts
(used by the CPU tests with world_size 2).
"""

from __future__ import annotations

import os


def env_rank() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when launched directly."""
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def init(backend: str | None = None):
    """Initialise ``torch.distributed`` from the environment; returns (rank, world, local_rank, device)."""
    import torch
    import torch.distributed as dist

    rank, world, local = env_rank()
    use_gpu = torch.cuda.is_available()
    device = torch.device(f'cuda:{local}') if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(backend or ('nccl' if use_gpu else 'gloo'), rank=rank, world_size=world)
    return rank, world, local, device


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous balanced shard [lo, hi) of ``n_items`` independent units for ``rank``."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """All-reduce(max) of a scalar (the benchmark's elapsed time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def argmin_first(costs_local: list[float], lo: int, n_total: int, device=None) -> int:
    """Global index of the first strict minimum of per-candidate costs that are sharded over ranks
    (the reference's candidate selection rule, api.cc:243-247).  ``costs_local`` are this rank's entries [lo, lo+len)."""
    import torch
    import torch.distributed as dist

    full = torch.full((n_total,), float('inf'), dtype=torch.float32, device=device or 'cpu')
    if costs_local:
        full[lo : lo + len(costs_local)] = torch.tensor(costs_local, dtype=torch.float32, device=full.device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(full, op=dist.ReduceOp.MIN)
    best = 0
    vals = full.tolist()
    for i in range(1, n_total):
        if vals[i] < vals[best]:
            best = i
    return best


def gather_to_rank0(obj):
    """Gather picklable per-rank results on rank 0 (list ordered by rank); other ranks get None."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def solve_many_sharded(kernels, **opts):
    """Solve independent matrices over all ranks: rank r solves its shard on its own GPU; rank 0 returns all results."""
    from . import _binary

    rank, world, local, _ = init()
    if _binary.device_count() > 0:
        _binary.set_device(local % _binary.device_count())
    lo, hi = shard_bounds(len(kernels), rank, world)
    mine = _binary.solve_many(kernels[lo:hi], **opts) if hi > lo else []
    parts = gather_to_rank0(mine)
    if parts is None:
        return None
    return [p for part in parts for p in part]
