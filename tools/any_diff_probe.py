#!/usr/bin/env python3
"""Any-hit flags of one trace mode that differ from the CPU oracle's on a scene's captured bounce rays: which way each
differs (the oracle says occluded / free), and the oracle's closest hit along the same ray (instance, distance), which
shows whether the rays involved start on the geometry they hit.

    python tools/any_diff_probe.py [scene] [--mode 2] [--width 960 --height 540 --iters 2]
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
    ap.add_argument("scene", nargs="?", default="party_armadillo")
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--iters", type=int, default=2)
    args = ap.parse_args()
    import torch

    import cuda_path_tracer_amd as pt
    import oracle_lib as ora

    w, h = args.width, args.height
    hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", args.scene + ".npz")).build(width=w, height=h, max_depth=8)
    ctx = pt.Context(0)
    desc = hs.desc()
    ctx.upload(desc)
    ctx.integrator_init(w, h)
    buf = torch.empty(w * h * 8, dtype=torch.float32, device="cuda:0")
    chunks = []
    for it in range(args.iters):
        for b in (1, 2, 3):
            n = ctx.capture_bounce_rays(w, h, it, b, buf.data_ptr(), w * h)
            chunks.append(buf[: n * 8].cpu().numpy().view(pt.abi.RAY).reshape(-1).copy())
    rays = np.concatenate(chunks)
    ref_any = ora.trace_any(desc, rays, nthreads=16)
    ref = ora.trace_closest(desc, rays, nthreads=16)
    ctx.set_trace_mode(args.mode)
    occ = ctx.trace_any_host(rays)
    hits = ctx.trace_closest_host(rays)
    bad = np.nonzero(occ != ref_any)[0]
    print(f"{args.scene}: {len(rays)} rays, mode {args.mode}: {len(bad)} any-hit flags differ "
          f"({int(ref_any[bad].sum())} where the oracle says occluded), loose boxes: {bool(os.environ.get('PTAMD_LOOSE_INSTANCE_BOXES'))}")
    for i in bad:
        print(f"  ray {i}: oracle any {int(ref_any[i])} ours {int(occ[i])}; oracle closest inst {int(ref['inst'][i])} t {float(ref['t'][i]):.3e}; "
              f"ours closest inst {int(hits['inst'][i])} t {float(hits['t'][i]):.3e}; t_max {float(rays['tmax'][i]):.3e}; origin {rays['origin'][i]} direction {rays['direction'][i]}")
    ctx.close()


if __name__ == "__main__":
    main()
