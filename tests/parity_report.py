#!/usr/bin/env python3
"""Same-seed agreement of the GPU render with the CPU oracle, per scene and trace mode: share of the
pixels that agree after 1 and after 16 samples, RMSE of the 16-sample image relative to its mean.
The numbers the thresholds of tests/test_gpu_render.py and tests/test_gpu_wide.py are set from; the
output of the last run is kept as profiles/parity_report.json.

    python tests/parity_report.py [out.json]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import cuda_path_tracer_amd as pt  # noqa: E402
import oracle_lib as ora  # noqa: E402
from conftest import make_tiny_scene  # noqa: E402

SCENES = (("tiny", 64, 48, 6), ("teapot", 96, 96, 6), ("cornell", 80, 80, 8), ("pbr_metallic", 96, 54, 8), ("pbr_ior", 96, 54, 8),
          ("helmet", 128, 72, 8), ("instancing_stress", 64, 64, 5), ("party_armadillo", 96, 54, 8))
N = 16


def close_share(film, ref, atol, rtol):
    return float((np.abs(film - ref) <= atol + rtol * np.abs(ref)).all(axis=2).mean())


def main():
    report = {}
    for name, w, h, depth in SCENES:
        hs = make_tiny_scene(pt) if name == "tiny" else pt.SceneSpec.load(os.path.join(ROOT, "scenes", name + ".npz")).build(width=w, height=h, max_depth=depth)
        ctx = pt.Context(0)
        desc = hs.desc()
        ctx.upload(desc)
        ctx.integrator_init(w, h)
        refs = []
        ref = None
        for it in range(N):
            ref, _ = ora.render(desc, w, h, it, ref, nthreads=16)
            refs.append(ref.copy())
        rec = {"resolution": [w, h], "depth": depth}
        for mode in (0, 1, 2, 3):
            ctx.set_trace_mode(mode)
            ctx.render(w, h, 0)
            one = ctx.film_download(w, h)
            for it in range(1, N):
                ctx.render(w, h, it)
            film = ctx.film_download(w, h)
            rmse = float(np.sqrt(np.mean((film / N - refs[-1] / N) ** 2)))
            rec[f"mode{mode}"] = {
                "bit_equal_pixels_1_sample": float((one.view(np.uint32) == refs[0].view(np.uint32)).all(axis=2).mean()),
                "close_pixels_1_sample": close_share(one, refs[0], 1e-4, 1e-3),
                "close_pixels_16_samples": close_share(film, refs[-1], 1e-3, 2e-3),
                "rmse_over_mean_16_samples": rmse / float(refs[-1].mean() / N)}
        report[name] = rec
        print(name, json.dumps(rec), flush=True)
        ctx.close()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
