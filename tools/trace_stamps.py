#!/usr/bin/env python3
"""Diagnostic: where a wave's clock goes inside the traversal loop of the closest-hit query (bench
ray batch, helmet 1080p).  Needs a library built with -DPT_TRACE_STAMPS; PT_TRACE_MODE picks the mode."""
import ctypes as C
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
for it in range(8):
    for b in (1, 2, 3):
        n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
rays = torch.cat(chunks)
n = rays.numel() // 8
hits = torch.empty(n * 8, dtype=torch.float32, device=dev)
lib = pt.abi.load_library()
out = (C.c_ulonglong * 8)()
for mode in (0, 2):
    ctx.set_trace_mode(mode)
    ctx.trace_closest(rays.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    lib.pt_debug_trace_stamps(out)
    ctx.trace_closest(rays.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    lib.pt_debug_trace_stamps(out)
    names = ["box: votes, load issue, pop", "box: wait for node", "box: tests + push", "triangle rounds", "enter / level / votes", "refill"]
    total = sum(out[k] for k in range(6))
    print(f"trace mode {mode}: {n} rays, {out[6]} box rounds, {out[7]} triangle rounds, {total / 4096:.0f} clocks per wave")
    for k, name in enumerate(names):
        per = out[k] / max(1, out[6] if k < 3 else out[7] if k == 3 else out[6])
        print(f"  {name:30s} {100.0 * out[k] / total:5.1f} %   {per:8.1f} clocks per {'box round' if k != 3 else 'triangle round'}")
