#!/usr/bin/env python3
"""Diagnostic: how concentrated are node visits?  Per-node visit counts of the box loop for the bench's bounce
rays, camera rays and shadow rays (helmet 1080p), and the share of all visits that the N most visited node
records take - what a cache of N records next to the stack would absorb.  Needs a library built with
-DPT_NODE_HIST (PTAMD_LIB=...)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_path_tracer_amd as pt  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "helmet"
W, H = 1920, 1080
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", scene + ".npz")).build(width=W, height=H)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
dev = torch.device("cuda", 0)
tmp = torch.empty(W * H * 8, dtype=torch.float32, device=dev)
sets = {}
chunks = []
for it in range(4):
    for b in (1, 2, 3):
        n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
sets["bounce"] = torch.cat(chunks)
n = ctx.capture_bounce_rays(W, H, 0, 0, tmp.data_ptr(), W * H)
sets["camera"] = tmp[: n * 8].clone()
chunks = []
for it in range(4):
    for b in (0, 1):
        n = ctx.capture_shadow_rays(W, H, it, b, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
sets["shadow"] = torch.cat(chunks)
lib = pt.abi.load_library()
SIZE = 1 << 20
out = (C.c_uint * SIZE)()
lib.pt_debug_node_hist(out, SIZE)
for mode in (0, 2):
    ctx.set_trace_mode(mode)
    for name, rays in sets.items():
        n = rays.numel() // 8
        hits = torch.empty(n * 8, dtype=torch.float32, device=dev)
        ctx.trace_closest(rays.data_ptr(), n, hits.data_ptr())
        ctx.synchronize()
        lib.pt_debug_node_hist(out, SIZE)
        h = np.frombuffer(out, dtype=np.uint32).astype(np.int64)
        total = h.sum()
        order = np.sort(h)[::-1]
        cum = np.cumsum(order) / max(1, total)
        first = np.cumsum(h) / max(1, total)
        line = ", ".join(f"top {k}: {cum[k - 1]:.3f} (first {k} records: {first[k - 1]:.3f})" for k in (8, 32, 64, 128, 256, 512, 1024))
        print(f"{scene} mode {mode} {name}: {n} rays, {total / n:.2f} visits per ray, {int((h > 0).sum())} records visited; share of visits: {line}", flush=True)
