#!/usr/bin/env python3
"""Where the shadow rays of a scene spend their work: visit counts per ray (mean, percentiles, maximum) of the
captured shadow rays in every trace mode, share of occluded rays, and the kernel time of the any-hit query.

    python tools/shadow_stats.py [scene] [--iters N]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default="helmet")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    import torch

    import cuda_path_tracer_amd as pt

    W, H = args.width, args.height
    hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", args.scene + ".npz")).build(width=W, height=H, max_depth=8)
    ctx = pt.Context(0)
    hs.update_device(ctx)
    ctx.integrator_init(W, H)
    dev = torch.device("cuda", 0)
    tmp = torch.empty(W * H * 8, dtype=torch.float32, device=dev)
    for kind in ("shadow", "bounce"):
        chunks = []
        for it in range(args.iters):
            for b in ((0, 1, 2) if kind == "shadow" else (1, 2, 3)):
                cap = ctx.capture_shadow_rays if kind == "shadow" else ctx.capture_bounce_rays
                n = cap(W, H, it, b, tmp.data_ptr(), W * H)
                chunks.append(tmp[: n * 8].clone())
        rays = torch.cat(chunks)
        n = rays.numel() // 8
        occ = torch.empty(n, dtype=torch.uint8, device=dev)
        ctr = torch.empty(n * 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        print(f"{args.scene} {kind} rays: {n}")
        for mode in (0, 1, 2, 3):
            ctx.set_trace_mode(mode)
            ctx.trace_any_counted(rays.data_ptr(), n, occ.data_ptr(), ctr.data_ptr())
            ctx.synchronize()
            c = ctr.cpu().numpy().reshape(n, 4).astype(np.int64)
            o = occ.cpu().numpy()
            for _ in range(2):
                ctx.trace_any(rays.data_ptr(), n, occ.data_ptr())
            ctx.synchronize()
            ctx.reset_stats()
            ctx.set_profiling(True)
            for _ in range(5):
                ctx.trace_any(rays.data_ptr(), n, occ.data_ptr())
            ctx.synchronize()
            ctx.set_profiling(False)
            st = ctx.stats()
            ms = st.ms_any / max(1, st.n_any)
            line = [f"mode {mode}: {ms:.3f} ms {n / ms / 1e6:.3f} Grays/s, occluded {o.mean():.3f}"]
            for k, name in enumerate(("interior", "instances", "triangles", "leaves")):
                v = c[:, k]
                line.append(f"{name} mean {v.mean():.1f} p50 {np.percentile(v, 50):.0f} p90 {np.percentile(v, 90):.0f} p99 {np.percentile(v, 99):.0f} max {v.max()}")
            print("  " + "; ".join(line), flush=True)
            if mode == 0:
                free = o == 0
                print(f"    unoccluded rays: interior mean {c[free, 0].mean():.1f}, triangles mean {c[free, 2].mean():.1f}; "
                      f"occluded rays: interior mean {c[~free, 0].mean() if (~free).any() else 0:.1f}, triangles mean {c[~free, 2].mean() if (~free).any() else 0:.1f}")
        ctx.set_trace_mode(0)


if __name__ == "__main__":
    main()
