#!/usr/bin/env python3
"""Diagnostic: when the waves of one closest-hit query start and finish (the bench's ray batch).
Needs a library built with -DPT_WAVE_TIMES (cuda-path-tracer_amd/build.sh -DPT_WAVE_TIMES),
named by PTAMD_LIB."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_path_tracer_amd as pt  # noqa: E402

W, H = 1920, 1080
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz")).build(width=W, height=H)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
dev = torch.device("cuda", 0)
tmp = torch.empty(W * H * 8, dtype=torch.float32, device=dev)
chunks = []
for it in range(ITERS):
    for b in (1, 2, 3):
        n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
rays = torch.cat(chunks)
n = rays.numel() // 8
hits = torch.empty(n * 8, dtype=torch.float32, device=dev)
lib = pt.abi.load_library()
waves = 4096
out = np.zeros(2 * waves, np.uint64)
spans, tails, p0s, p50s, p99s = [], [], [], [], []
for rep in range(12):
    ctx.trace_closest(rays.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    assert lib.pt_debug_wave_times(out.ctypes.data_as(C.c_void_p), C.c_uint(waves)) == 0
    t = out.reshape(waves, 2).astype(np.int64)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # microseconds
    span = end.max()
    if rep < 2:
        continue
    spans.append(span)
    tails.append((span - end).sum() / (waves * span))
    p0s.append(end.min())
    p50s.append(np.median(end))
    p99s.append(np.percentile(end, 99))
print(f"{n} rays, {waves} waves, 10 launches: span median {np.median(spans):.1f} us (min {min(spans):.1f}, max {max(spans):.1f})")
print(f"wave finish us (medians over launches): first {np.median(p0s):.1f}  p50 {np.median(p50s):.1f}  p99 {np.median(p99s):.1f}  last {np.median(spans):.1f}")
print(f"wave-time unused after a wave's last ray: {np.median(tails):.4f} of the launch")
ctr = torch.zeros(n * 4, dtype=torch.int32, device=dev)
ctx.trace_closest_counted(rays.data_ptr(), n, hits.data_ptr(), ctr.data_ptr())
ctx.synchronize()
c = ctr.view(n, 4).cpu().numpy().astype(np.int64)
steps = c[:, 0] + c[:, 1] + c[:, 3]  # box rounds, instance enters, leaf visits
print("steps per ray (interior + instances + leaves): mean %.1f  " % steps.mean() + " ".join(f"p{q}={np.percentile(steps, q):.0f}" for q in (50, 90, 99, 99.9, 99.99, 100)))
order = np.argsort(-steps)[:10]
print("longest rays: steps", steps[order].tolist(), "at stream positions", (order / n).round(3).tolist())
