"""Sharding of one render job over ranks (one process per GPU).

A path depends only on (pixel, iteration) through its sampler seed
(sampler/independent.h:14-23) and on read-only scene data; the film is the only shared mutable
state and it is additive (path_trace.cu:391-407).  So iterations are dealt out to ranks in blocks
and the private films are summed once at the end; no collective runs while rendering.

The alternative is to deal out the image instead of the samples: every rank renders all iterations
of its own band of rows (pt_set_render_window) and the same reduce assembles the frame, since films
of disjoint windows add up to the full film.  Path state shrinks to a rank's share of the pixels, so
this is the plan for frames whose path state would not fit one GPU; its load balance depends on the
scene, where the sample plan's is perfect, so samples are the default.
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


def tile_windows(world: int, width: int, height: int) -> List[Tuple[int, int, int, int]]:
    """(x0, y0, w, h) of every rank's window: bands of whole 8-pixel rows (the pipeline's path slots are
    8x8 pixel tiles), heights differing by at most 8, together covering the frame exactly once.  A rank
    beyond the number of 8-row strips gets an empty window (h = 0) and renders nothing."""
    if world < 1 or width < 1 or height < 1:
        raise ValueError("world, width and height must be positive")
    strips = (height + 7) // 8
    base, extra = divmod(strips, world)
    out, row = [], 0
    for r in range(world):
        rows = min(height - row, 8 * (base + (1 if r < extra else 0)))
        out.append((0, row, width, rows))
        row += rows
    assert row == height
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
        if film.is_cuda and dist.get_backend() == "gloo":  # CPU-only collective (tests): stage through host memory
            host = film.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            film.copy_(host)
        else:
            dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def render_job(ctx, width: int, height: int, total_iterations: int, block: int, rank: int, world: int, film=None, shard: str = "samples"):
    """One rank's part of a sharded render: clears its private film (only the rank that owns
    iteration 0 would get that from pt_render_batch), submits its passes back to back, and,
    given a device tensor `film` of W*H*3 fp32, ends with the one collective of the job: the private
    films summed onto rank 0 (`film` holds the sum there, and the context's film is replaced by it).
    shard = "samples": the rank's iteration blocks of the whole frame; "tiles": every iteration of the
    rank's band of rows.  Returns the number of iterations this rank rendered."""
    if shard not in ("samples", "tiles"):
        raise ValueError("shard must be 'samples' or 'tiles'")
    ctx.integrator_reset(width, height)
    done = 0
    if shard == "samples":
        for first, count in iteration_blocks(rank, world, total_iterations, block):
            ctx.render_batch(width, height, first, count)
            done += count
    else:
        x0, y0, w, h = tile_windows(world, width, height)[rank]
        if h > 0:
            ctx.set_render_window(x0, y0, w, h)
            try:
                for first, count in iteration_blocks(0, 1, total_iterations, block):
                    ctx.render_batch(width, height, first, count)
                    done += count
            finally:
                ctx.synchronize()
                ctx.set_render_window()
    ctx.synchronize()
    if film is not None:
        ctx.film_copy_to_device(film.data_ptr())
        reduce_film(film, dst=0)
        if rank == 0 and world > 1:
            ctx.film_copy_from_device(film.data_ptr())
    return done


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
