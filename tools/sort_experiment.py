#!/usr/bin/env python3
"""Experiment: how much faster is closest-hit traversal when the ray batch is ordered by
(origin cell, direction octant)?  Sorting is done offline with torch; only the trace is timed."""
import os
import sys

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
chunks = []
for it in range(4):
    for b in (1, 2, 3):
        n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
rays = torch.cat(chunks).view(-1, 8)
n = rays.shape[0]
hits = torch.empty(n * 8, dtype=torch.float32, device=dev)


def time_trace(r, label):
    r = r.contiguous()
    for _ in range(2):
        ctx.trace_closest(r.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    ctx.reset_stats()
    ctx.set_profiling(True)
    for _ in range(10):
        ctx.trace_closest(r.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    ctx.set_profiling(False)
    st = ctx.stats()
    ms = st.ms_closest / st.n_closest
    print(f"{label:44s} {ms:7.3f} ms  {n / ms / 1e3:8.1f} Mrays/s")


def morton3(x, y, z, bits):
    key = torch.zeros_like(x)
    for i in range(bits):
        key |= ((x >> i) & 1) << (3 * i) | ((y >> i) & 1) << (3 * i + 1) | ((z >> i) & 1) << (3 * i + 2)
    return key


o, d = rays[:, 0:3], rays[:, 4:7]
lo, hi = o.min(0).values, o.max(0).values
time_trace(rays, "queue order (as captured)")
time_trace(rays[torch.randperm(n, device=dev)], "random permutation")
octant = ((d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2))
for bits in (2, 3, 4, 5, 7):
    cell = ((o - lo) / (hi - lo + 1e-6) * (1 << bits)).long().clamp(0, (1 << bits) - 1)
    m = morton3(cell[:, 0], cell[:, 1], cell[:, 2], bits)
    time_trace(rays[torch.argsort((m << 3) | octant)], f"cell {bits} bits/axis, then octant")
    time_trace(rays[torch.argsort((octant << (3 * bits)) | m)], f"octant, then cell {bits} bits/axis")
# finer direction bins: 6 faces x 4x4
time_trace(rays[torch.argsort(octant)], "octant only")
