#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of incoherent closest-hit rays on the helmet scene at 1080p
(plus spp/s of the full path tracer on the same scene), 1..8 MI355X, one process per GPU.

A "step" is one pass of the closest-hit traversal kernel (pt_trace_closest) over a
device-resident batch of captured incoherent rays: the extension rays entering bounces 1..3
of `--capture-iters` iterations of the helmet scene at 1920x1080.  Rays, hits and the scene are
resident in HBM before the timed region starts.  With N ranks every rank traces its own batch
(captured from its own iterations), no data-path collective: weak scaling.  The spp/s leg
shards iterations over ranks (rank r renders r, r+N, ...) and ends with one RCCL reduce of the
fp32 film to rank 0, inside the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak


def algorithmic_bytes(n_rays: int, interior: int, tlas_leaves: int, tris: int) -> float:
    """Layout-independent traffic of a closest-hit batch: per ray 32 B in + 16 B out + the root
    and both children of every expanded interior node (32 B each) + a 3x4 matrix per instance
    entered (48 B) + three fp32 positions per triangle tested (36 B)."""
    return 48.0 * n_rays + 32.0 * (n_rays + 2.0 * interior) + 48.0 * tlas_leaves + 36.0 * tris


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="helmet")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--capture-iters", type=int, default=8)
    ap.add_argument("--spp-steps", type=int, default=1536, help="render iterations per rank for the spp/s leg (0 = skip)")
    ap.add_argument("--spp-batch", type=int, default=128,
                    help="iterations kept in flight per wavefront pass in the spp/s leg (about 0.5 GB of path state per sample at 1080p: three passes in flight)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--rays", default="captured", choices=["captured", "synthetic"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        dist.init_process_group(backend="nccl", device_id=dev)

    import cuda_path_tracer_amd as pt

    W, H = args.width, args.height
    spec = pt.SceneSpec.load(os.path.join(ROOT, "scenes", args.scene + ".npz"))
    hs = spec.build(width=W, height=H)
    ctx = pt.Context(local_rank)
    hs.update_device(ctx)
    ctx.integrator_init(W, H)
    depth = hs.max_depth

    # ---- the ray batch (device resident)
    if args.rays == "captured":
        cap = W * H
        tmp = torch.empty(cap * 8, dtype=torch.float32, device=dev)
        chunks = []
        for it in range(args.capture_iters):
            iteration = rank * args.capture_iters + it
            for bounce in (1, 2, 3):
                if bounce >= depth:
                    continue
                n = ctx.capture_bounce_rays(W, H, iteration, bounce, tmp.data_ptr(), cap)
                chunks.append(tmp[: n * 8].clone())
        rays = torch.cat(chunks)
        del tmp, chunks
        workload = f"{args.scene} {W}x{H}: extension rays entering bounces 1-3 of {args.capture_iters} iterations"
    else:
        n = 1 << 24
        rays = torch.empty(n * 8, dtype=torch.float32, device=dev)
        ctx.generate_incoherent_rays(rays.data_ptr(), n, rank * n)
        workload = f"{args.scene}: 2^24 synthetic incoherent rays in the scene box"
    n_rays = rays.numel() // 8
    hits = torch.empty(n_rays * 8, dtype=torch.float32, device=dev)

    # visit counts of the batch (for the algorithmic byte count), outside the timed region
    counters = torch.zeros(n_rays * 4, dtype=torch.int32, device=dev)
    ctx.trace_closest_counted(rays.data_ptr(), n_rays, hits.data_ptr(), counters.data_ptr())
    ctx.synchronize()
    csum = counters.view(-1, 4).to(torch.int64).sum(dim=0).cpu().numpy()
    del counters
    bytes_per_launch = algorithmic_bytes(n_rays, int(csum[0]), int(csum[1]), int(csum[2]))

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- leg 1: closest-hit Mrays/s
    for _ in range(args.warmup):
        ctx.trace_closest(rays.data_ptr(), n_rays, hits.data_ptr())
    ctx.synchronize()
    ctx.reset_stats()
    ctx.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.trace_closest(rays.data_ptr(), n_rays, hits.data_ptr())
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    st = ctx.stats()
    kernel_ms = st.ms_closest / max(1, st.n_closest)
    elapsed = pt.sharding.max_over_ranks(elapsed, dev)
    if distributed:
        nr = torch.tensor([n_rays], dtype=torch.int64, device=dev)
        dist.all_reduce(nr, op=dist.ReduceOp.SUM)
        total_rays = int(nr.item())
    else:
        total_rays = n_rays
    mrays = total_rays * args.steps / elapsed / 1e6

    # ---- leg 2: spp/s of the full path tracer, iterations strided over ranks + film reduce
    spp = None
    if args.spp_steps > 0:
        film = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
        B = max(1, args.spp_batch)
        passes = max(1, args.spp_steps // B)
        args.spp_steps = passes * B
        for w_pass in range(4):  # untimed: consecutive passes rotate through the library's sets of path state (allocated on first use)
            ctx.render_batch(W, H, w_pass * B, B)
        ctx.synchronize()
        ctx.integrator_reset(W, H)
        ctx.reset_stats()
        barrier()
        t0 = time.perf_counter()
        # rank r owns the iteration blocks r, r+N, r+2N, ... of B consecutive iterations each
        for first, count in pt.sharding.iteration_blocks(rank, world, args.spp_steps * world, B):
            ctx.render_batch(W, H, first, count)
        ctx.synchronize()
        ctx.film_copy_to_device(film.data_ptr())
        pt.sharding.reduce_film(film, dst=0)
        barrier()
        spp_elapsed = pt.sharding.max_over_ranks(time.perf_counter() - t0, dev)
        st2 = ctx.stats()
        # the same leg with trace mode 1 (any-hit searches skip boxes beyond the shadow ray's t_max, which
        # the reference does not do): reported beside the reference-exact figure, never as `value`
        ctx.set_trace_mode(1)
        ctx.integrator_reset(W, H)
        barrier()
        t0 = time.perf_counter()
        for first, count in pt.sharding.iteration_blocks(rank, world, args.spp_steps * world, B):
            ctx.render_batch(W, H, first, count)
        ctx.synchronize()
        barrier()
        mode1_elapsed = pt.sharding.max_over_ranks(time.perf_counter() - t0, dev)
        ctx.set_trace_mode(0)
        spp = {"spp_per_s": args.spp_steps * world / spp_elapsed, "ms_per_spp": spp_elapsed * 1e3 / (args.spp_steps * world),
               "mpaths_per_s": args.spp_steps * world * W * H / spp_elapsed / 1e6,
               "rays_last_pass": {"closest": int(st2.closest_rays), "shadow": int(st2.shadow_rays)},
               "iterations": args.spp_steps * world, "samples_in_flight": B,
               "spp_per_s_trace_mode_1": args.spp_steps * world / mode1_elapsed, "film_reduce": "rccl reduce(sum) to rank 0" if distributed else "none"}

    # ---- leg 3: CPU baseline (rank 0, N == 1 only): the oracle's traversal on a bounded sample
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ora

        desc = hs.desc()
        threads = max(1, min(os.cpu_count() or 1, 16))
        rays_h = rays[: min(n_rays, 1 << 22) * 8].cpu().numpy().view(pt.abi.RAY).reshape(-1)
        probe = rays_h[: min(len(rays_h), 100000)]
        t0 = time.perf_counter()
        ora.trace_closest(desc, probe, nthreads=threads)
        rate = len(probe) / max(time.perf_counter() - t0, 1e-6)
        sample = rays_h[: int(min(len(rays_h), max(len(probe), rate * args.cpu_seconds)))]
        repeats = int(max(1, min(64, round(rate * args.cpu_seconds / len(sample)))))
        t0 = time.perf_counter()
        for _ in range(repeats):
            ora.trace_closest(desc, sample, nthreads=threads)
        cpu_t = time.perf_counter() - t0
        cpu = {"value": len(sample) * repeats / cpu_t / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
               "sample": f"first {len(sample)} rays of the same batch traced {repeats}x, oracle/liboracle.so ora_trace_closest, "
                         f"{threads} threads, {cpu_t:.1f} s"}

    # HBM-side bytes per launch from the committed PMC passes (counters cannot be read in-process);
    # only quoted when the profile was taken on this same batch
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f)["k_trace_closest"]
        if abs(rec["rays"] - n_rays) <= 0.01 * n_rays and args.rays == "captured":
            traffic = rec["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        traffic = None

    if rank == 0:
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": f"Mrays/s incoherent closest-hit ({args.scene}, {W}x{H} {args.rays} bounce rays)", "value": mrays, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "rays_per_rank": n_rays, "scene": args.scene, "resolution": [W, H], "max_depth": depth,
                       "parallelism": f"rays sharded over {world} rank(s), scene replicated"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_trace_closest", "kernel_ms": kernel_ms, "launches": int(st.n_closest),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "visits_per_ray": {"interior": csum[0] / n_rays, "tlas_leaves": csum[1] / n_rays, "triangles": csum[2] / n_rays}},
            "cpu_baseline": cpu,
            "spp": spp,
        }
        print(json.dumps(out))
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
