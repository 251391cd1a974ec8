#!/usr/bin/env python3
"""Diagnostic: latency of one traversal step when a wave has the GPU to itself.  Traces the 64
longest rays of the bench's ray batch alone, then 64 copies of the longest one."""
import os
import sys
import time

import numpy as np

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
for it in range(2):
    for b in (1, 2, 3):
        n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
rays = torch.cat(chunks).view(-1, 8)
n = rays.shape[0]
hits = torch.empty(n * 8, dtype=torch.float32, device=dev)
ctr = torch.zeros(n * 4, dtype=torch.int32, device=dev)
ctx.trace_closest_counted(rays.data_ptr(), n, hits.data_ptr(), ctr.data_ptr())
ctx.synchronize()
c = ctr.view(n, 4).cpu().numpy().astype(np.int64)
steps = c[:, 0] + c[:, 1] + c[:, 3]
order = np.argsort(-steps)


def timed(sub, label):
    sub = sub.contiguous()
    m = sub.shape[0]
    best = 1e9
    for rep in range(20):
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.trace_closest(sub.data_ptr(), m, hits.data_ptr())
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{label}: {m} rays, {best * 1e6:.1f} us")
    return best


empty = timed(rays[:1] * 0 + torch.tensor([1e6, 1e6, 1e6, 0, 0, 0, 1, 3e38], device=dev), "one ray that misses the scene (launch overhead)")
idx = torch.from_numpy(order[:64].copy()).to(dev)
t = timed(rays[idx], "the 64 longest rays")
print(f"  longest {steps[order[0]]} steps -> {(t - empty) * 1e6 / steps[order[0]]:.2f} us per step of the longest ray")
one = rays[idx[:1]].repeat(64, 1)
t = timed(one, "64 copies of the longest ray")
print(f"  {steps[order[0]]} steps (interior {c[order[0], 0]}, instances {c[order[0], 1]}, triangles {c[order[0], 2]}, leaves {c[order[0], 3]}) -> {(t - empty) * 1e6 / steps[order[0]]:.2f} us per step")
t = timed(rays[idx[:1]], "the longest ray alone")
print(f"  -> {(t - empty) * 1e6 / steps[order[0]]:.2f} us per step")
