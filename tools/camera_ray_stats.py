#!/usr/bin/env python3
"""Camera rays of a scene (the extension rays of bounce 0): how many of them reach no instance box at all, how many enter
an instance and still miss, the visits of each group, and the closest-hit query time on all of them against the time on
only those that reach an instance.

    python tools/camera_ray_stats.py [scene] [--width 1920 --height 1080]
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
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    import torch

    import cuda_path_tracer_amd as pt

    w, h = args.width, args.height
    hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", args.scene + ".npz")).build(width=w, height=h, max_depth=8)
    ctx = pt.Context(0)
    ctx.upload(hs.desc())
    ctx.integrator_init(w, h)
    buf = torch.empty(w * h * 8, dtype=torch.float32, device="cuda:0")
    n = ctx.capture_bounce_rays(w, h, 0, 0, buf.data_ptr(), w * h)
    hits = torch.empty(n * 8, dtype=torch.float32, device="cuda:0")
    ctr = torch.zeros(n * 4, dtype=torch.int32, device="cuda:0")
    for mode in (0, 2):
        ctx.set_trace_mode(mode)
        ctr.zero_()
        ctx.trace_closest_counted(buf.data_ptr(), n, hits.data_ptr(), ctr.data_ptr())
        ctx.synchronize()
        c = ctr.cpu().numpy().reshape(n, 4)
        hit = hits.cpu().numpy().view(pt.abi.HIT).reshape(-1)["inst"] >= 0
        none = c[:, 1] == 0
        print(f"{args.scene} {w}x{h} mode {mode}: {n} camera rays; reach no instance {none.mean():.3f}, enter and miss {(~none & ~hit).mean():.3f}, hit {hit.mean():.3f}")
        for name, m in (("no instance", none), ("enter, miss", ~none & ~hit), ("hit", hit)):
            if m.any():
                print(f"   {name:12s}: interior {c[m, 0].mean():6.2f}  instances {c[m, 1].mean():5.2f}  triangles {c[m, 2].mean():6.2f}")

        def timed(ptr, count):
            import time
            for _ in range(3):
                ctx.trace_closest(ptr, count, hits.data_ptr())
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.trace_closest(ptr, count, hits.data_ptr())
            ctx.synchronize()
            return (time.perf_counter() - t0) * 1e3 / 20

        rays = buf[: n * 8].view(n, 8)
        keep = rays[torch.from_numpy(~none).cuda()].contiguous()
        t_all, t_keep = timed(buf.data_ptr(), n), timed(keep.data_ptr(), keep.shape[0])
        print(f"   query time: all {t_all:.3f} ms ({n / t_all / 1e6:.2f} Grays/s), only rays that reach an instance {t_keep:.3f} ms ({keep.shape[0]} rays)")
    ctx.close()


if __name__ == "__main__":
    main()
