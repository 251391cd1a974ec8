#!/usr/bin/env python3
"""Same-seed agreement of the GPU render with the CPU oracle: share of pixels that agree and ray counts."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import cuda_path_tracer_amd as pt  # noqa: E402
import oracle_lib as ora  # noqa: E402

for name, w, h, depth in (("helmet", 256, 144, 8), ("teapot", 128, 128, 6), ("cornell", 128, 128, 8), ("pbr_metallic", 192, 108, 8)):
    hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", name + ".npz")).build(width=w, height=h, max_depth=depth)
    ctx = pt.Context(0)
    desc = hs.desc()
    ctx.upload(desc)
    ctx.integrator_init(w, h)
    ref = None
    tot = [0, 0]
    for it in range(2):
        ctx.render(w, h, it)
        ref, c = ora.render(desc, w, h, it, ref, nthreads=8)
        tot[0] += c[0]; tot[1] += c[1]
    st = ctx.stats()
    film = ctx.film_download(w, h)
    exact = (film.view(np.uint32) == ref.view(np.uint32)).all(axis=2).mean()
    close = (np.abs(film - ref) <= 1e-4 + 1e-3 * np.abs(ref)).all(axis=2).mean()
    print(f"{name:14s} bit-equal pixels {exact:.4f}  close pixels {close:.4f}  last-iteration rays gpu {st.closest_rays}/{st.shadow_rays}  oracle(2 it) {tot}")
    ctx.close()
