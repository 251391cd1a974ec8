#!/usr/bin/env python3
"""Any-hit flags of every trace mode against the CPU oracle on the shadow rays a scene's shading really emits
(the reference's semantics include a rule fast any-hit searches tend to miss: an instance whose last accepted
triangle lies at a negative recomputed distance reports no hit, kernel/intersection.h:139-149).

    python tools/any_hit_vs_oracle.py [scene ...] [--width W --height H --iters N]
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
    ap.add_argument("scenes", nargs="*", default=["helmet"])
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--iters", type=int, default=1)
    args = ap.parse_args()
    import torch

    import cuda_path_tracer_amd as pt
    import oracle_lib as ora

    w, h = args.width, args.height
    for name in args.scenes:
        hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", name + ".npz")).build(width=w, height=h, max_depth=8)
        ctx = pt.Context(0)
        desc = hs.desc()
        ctx.upload(desc)
        ctx.integrator_init(w, h)
        buf = torch.empty(w * h * 8, dtype=torch.float32, device="cuda:0")
        total = 0
        differ = {m: 0 for m in range(4)}
        gained = {m: 0 for m in range(4)}
        for it in range(args.iters):
            for bounce in (0, 1, 2):
                n = ctx.capture_shadow_rays(w, h, it, bounce, buf.data_ptr(), w * h)
                if n == 0:
                    continue
                rays = buf[: n * 8].cpu().numpy().view(pt.abi.RAY).reshape(-1)
                ref = ora.trace_any(desc, rays, nthreads=16)
                total += n
                for m in range(4):
                    ctx.set_trace_mode(m)
                    occ = ctx.trace_any_host(rays)
                    differ[m] += int((occ != ref).sum())
                    gained[m] += int(((occ == 1) & (ref == 0)).sum())
                ctx.set_trace_mode(0)
        print(name, total, "shadow rays; flags differing from the oracle per mode", differ, "of which occluded only on the GPU", gained, flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
