#!/usr/bin/env python3
"""Diagnostic: the captured shadow rays of the helmet scene as a stand-alone any-hit batch: rate of
pt_trace_any, occluded fraction, and the visit counts a closest-hit search with the same t_max needs."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_path_tracer_amd as pt  # noqa: E402

W, H = 1920, 1080
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz")).build(width=W, height=H)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
dev = torch.device("cuda", 0)
tmp = torch.empty(W * H * 8, dtype=torch.float32, device=dev)
for bounce in (0, 1, 2):
    chunks = []
    for it in range(4):
        n = ctx.capture_shadow_rays(W, H, it, bounce, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
    rays = torch.cat(chunks)
    n = rays.numel() // 8
    occ = torch.zeros(n, dtype=torch.uint8, device=dev)
    hits = torch.empty(n * 8, dtype=torch.float32, device=dev)
    ctr = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    for _ in range(2):
        ctx.trace_any(rays.data_ptr(), n, occ.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.trace_any(rays.data_ptr(), n, occ.data_ptr())
    ctx.synchronize()
    t_any = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.trace_closest(rays.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    t_closest = (time.perf_counter() - t0) / 10
    ctx.trace_closest_counted(rays.data_ptr(), n, hits.data_ptr(), ctr.data_ptr())
    ctx.synchronize()
    c = ctr.view(-1, 4).to(torch.float64).mean(dim=0).cpu().numpy()
    tm = rays.view(-1, 8)[:, 7]
    print(f"bounce {bounce}: {n} shadow rays, occluded {occ.float().mean().item():.3f}, any-hit {n / t_any / 1e9:.2f} Grays/s, "
          f"closest-hit with the same t_max {n / t_closest / 1e9:.2f} Grays/s; closest visits per ray: interior {c[0]:.1f}, "
          f"instances {c[1]:.2f}, triangles {c[2]:.1f}, leaves {c[3]:.1f}; t_max median {tm.median().item():.2f}", flush=True)
ctx.close()
