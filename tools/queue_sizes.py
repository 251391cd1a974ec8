#!/usr/bin/env python3
"""Prints the extension-queue size of every bounce for one batched pass of the bench scene."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import cuda_path_tracer_amd as pt  # noqa: E402

W, H, B = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz")).build(width=W, height=H)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
ctx.set_trace_mode(int(os.environ.get("PT_TRACE_MODE", "0")))
ctx.render_batch(W, H, 0, B)
ctx.synchronize()
st = ctx.stats()
print("paths", st.paths, "closest", st.closest_rays, "shadow", st.shadow_rays, "bounces", st.bounces_run)
print("queue sizes", list(st.queue_sizes)[: st.bounces_run + 1])
