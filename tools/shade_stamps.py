#!/usr/bin/env python3
"""Diagnostic: where the waves of k_shade spend their clock in one batched pass (needs a library
built with -DPT_PHASE_PROF -DPT_SHADE_STAMPS, named by PTAMD_LIB)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import cuda_path_tracer_amd as pt  # noqa: E402

W, H, B = 1920, 1080, 8
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz")).build(width=W, height=H)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
lib = pt.abi.load_library()
out = (C.c_ulonglong * 16)()
ctx.render_batch(W, H, 0, B)
ctx.synchronize()
lib.pt_debug_phase_prof(out)
ctx.render_batch(W, H, 8, B)
ctx.synchronize()
lib.pt_debug_phase_prof(out)
names = ["loop head -> chunk start", "path state loads", "tri records, small tables, hit resolve", "texels + light triangle",
         "light texels, NEE", "bsdf sample, stores", "queue appends", "group end -> next fetch"]
tot = sum(out[k] for k in range(8))
for k, n in enumerate(names):
    print(f"{n:40s} {out[k] / 1e6:10.1f} M clocks  {100.0 * out[k] / max(1, tot):5.1f} %")
