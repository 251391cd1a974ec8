"""Sharding of one render job over ranks (one process per GPU).

A path depends only on (pixel, iteration) through its sampler seed
(sampler/independent.h:14-23) and on read-only scene data; the film is the only shared mutable
state and it is additive (path_trace.cu:391-407).  So iterations are dealt out to ranks in blocks
and the private films are summed once at the end; no collective runs while rendering.
"""
from __future__ import annotations

from typing import Iterator, List, Tuple


def iteration_blocks(rank: int, world: int, total_iterations: int, block: int) -> List[Tuple[int, int]]:
    """(first_iteration, count) blocks of `rank`: block b covers iterations [b*block, (b+1)*block)
    and belongs to rank b % world.  Every iteration of [0, total) is owned by exactly one rank."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    if block < 1 or total_iterations < 0:
        raise ValueError("block must be >= 1 and total_iterations >= 0")
    out = []
    n_blocks = (total_iterations + block - 1) // block
    for b in range(rank, n_blocks, world):
        first = b * block
        out.append((first, min(block, total_iterations - first)))
    return out


def rays_shard(rank: int, world: int, n_rays: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of a ray batch for `rank`; sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_rays, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_film(film, dst: int = 0):
    """Sums the per-rank fp32 films onto rank `dst` (RCCL on GPUs, gloo on CPU tensors).
    A no-op outside an initialised process group."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def max_over_ranks(seconds: float, device) -> float:
    """The slowest rank's time: what a job's wall clock is."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_owned(world: int, total_iterations: int, block: int) -> Iterator[int]:
    """Every iteration in the order ranks 0..world-1 render them (for checks)."""
    for r in range(world):
        for first, count in iteration_blocks(r, world, total_iterations, block):
            yield from range(first, first + count)
