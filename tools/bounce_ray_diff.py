#!/usr/bin/env python3
"""How the extension rays of the GPU pipeline compare with the CPU oracle's, bounce by bounce (same seeds): rays that
are bit-equal, rays that differ in the last bits of their direction (the device's and libm's sin / cos / sqrt round
differently) and rays present on one side only (paths that took another branch after such a difference).

    python tools/bounce_ray_diff.py [scene] [--width W --height H --iteration I]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default="pbr_metallic")
    ap.add_argument("--width", type=int, default=192)
    ap.add_argument("--height", type=int, default=108)
    ap.add_argument("--iteration", type=int, default=0)
    args = ap.parse_args()
    import torch

    import cuda_path_tracer_amd as pt
    import oracle_lib as ora

    w, h = args.width, args.height
    hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", args.scene + ".npz")).build(width=w, height=h, max_depth=8)
    desc = hs.desc()
    ctx = pt.Context(0)
    ctx.upload(desc)
    ctx.integrator_init(w, h)
    buf = torch.empty(w * h * 8, dtype=torch.float32, device="cuda:0")
    for bounce in range(1, 6):
        n = ctx.capture_bounce_rays(w, h, args.iteration, bounce, buf.data_ptr(), w * h)
        gpu = buf[: n * 8].cpu().numpy().view(pt.abi.RAY).reshape(-1).copy()
        ref = ora.capture_bounce_rays(desc, w, h, args.iteration, bounce)
        def table(r):
            return np.concatenate([r["origin"], r["direction"]], axis=1).astype(np.float32)
        g, r = table(gpu), table(ref)
        gset = {row.tobytes() for row in g}
        exact = sum(1 for row in r if row.tobytes() in gset)
        # the rest: match by origin rounded to 1e-4 of the scene scale
        def key(row):
            return tuple(np.round(row[:3] * 1e4).astype(np.int64))
        rest_g = {key(row): row for row in g if row.tobytes() not in {x.tobytes() for x in r}} if len(r) < 200000 else {}
        near = far = 0
        worst = 0.0
        for row in r:
            if row.tobytes() in gset:
                continue
            m = rest_g.get(key(row))
            if m is None:
                far += 1
            else:
                near += 1
                worst = max(worst, float(np.abs(m - row).max()))
        print(f"{args.scene} bounce {bounce}: gpu {len(g)} rays, oracle {len(r)}; bit-equal {exact}, same origin cell with differing bits {near} "
              f"(max |d| {worst:.2e}), without a partner {far}", flush=True)


if __name__ == "__main__":
    main()
