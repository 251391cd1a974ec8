#!/usr/bin/env python3
"""Samples per second of the full path tracer in every trace mode (one GPU):
    python tools/spp_modes.py [scene] [--width W --height H --batch B --passes P]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_path_tracer_amd as pt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scene", nargs="?", default="helmet")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--passes", type=int, default=9)
ap.add_argument("--modes", default="0,1,2,3")
args = ap.parse_args()
W, H, B = args.width, args.height, args.batch
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", args.scene + ".npz")).build(width=W, height=H, max_depth=8)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
for w in range(4):
    ctx.render_batch(W, H, w * B, B)
ctx.synchronize()
films = {}
for mode in [int(m) for m in args.modes.split(",")]:
    ctx.set_trace_mode(mode)
    ctx.render_batch(W, H, 0, B)
    ctx.synchronize()
    t0 = time.perf_counter()
    for p in range(args.passes):
        ctx.render_batch(W, H, p * B, B)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    films[mode] = ctx.film_download(W, H)
    diff = ""
    if 0 in films and mode != 0:
        import numpy as np
        d = np.abs(films[mode] - films[0])
        diff = f"  pixels differing from mode 0: {(d > 0).any(axis=2).mean():.2e}, film mean rel diff {abs(films[mode].mean() - films[0].mean()) / films[0].mean():.2e}"
    print(f"{args.scene} {W}x{H} trace mode {mode}: {args.passes * B / dt:8.1f} spp/s{diff}", flush=True)
