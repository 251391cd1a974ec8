#!/usr/bin/env python3
"""Is the rendered film bit-identical to the CPU oracle's?  Per scene and trace mode: pixels whose accumulated
radiance differs in any bit after N same-seed samples at a resolution larger than the tests use.

    python tools/film_exactness.py [scene ...] [--width W --height H --samples N]
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
    ap.add_argument("scenes", nargs="*", default=["teapot", "cornell", "pbr_metallic", "pbr_ior", "helmet", "instancing_stress", "party_armadillo"])
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--samples", type=int, default=4)
    args = ap.parse_args()
    import torch  # noqa: F401

    import cuda_path_tracer_amd as pt
    import oracle_lib as ora

    w, h = args.width, args.height
    for name in args.scenes:
        hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", name + ".npz")).build(width=w, height=h, max_depth=8)
        desc = hs.desc()
        ctx = pt.Context(0)
        ctx.upload(desc)
        ctx.integrator_init(w, h)
        ref = None
        for it in range(args.samples):
            ref, _ = ora.render(desc, w, h, it, ref, nthreads=16)
        out = []
        for mode in (0, 1, 2, 3):
            ctx.set_trace_mode(mode)
            ctx.integrator_reset(w, h)
            ctx.render_batch(w, h, 0, args.samples)
            film = ctx.film_download(w, h)
            differ = (film.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
            out.append(f"mode {mode}: {int(differ.sum())} of {w * h} pixels differ")
        print(name, f"{w}x{h}, {args.samples} samples, film mean {ref.mean():.4f};", "; ".join(out), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
