#!/usr/bin/env python3
"""One render pass run serially with kernel timing on: time per kernel kind and rays per second of
the closest-hit (k_extend) and any-hit (k_shadow) kernels.  Usage: python tools/pass_breakdown.py [samples]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PTAMD_OVERLAP_PASSES"] = "0"
import cuda_path_tracer_amd as pt  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 1920, 1080
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz")).build(width=W, height=H)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
ctx.render_batch(W, H, 0, samples)
ctx.synchronize()
ctx.reset_stats()
ctx.set_profiling(True)
ctx.render_batch(W, H, samples, samples)
ctx.synchronize()
st = ctx.stats()
print(f"{samples} samples: closest rays {st.closest_rays / 1e6:.1f} M in {st.ms_closest:.2f} ms = {st.closest_rays / st.ms_closest / 1e6:.2f} Grays/s | "
      f"shadow rays {st.shadow_rays / 1e6:.1f} M in {st.ms_any:.2f} ms = {st.shadow_rays / st.ms_any / 1e6:.2f} Grays/s | "
      f"shade {st.ms_shade:.2f} ms | generate {st.ms_generate:.2f} ms | film {st.ms_film:.2f} ms")
print("queue sizes per bounce:", [st.queue_sizes[b] for b in range(min(9, st.bounces_run + 1))])
ctx.close()
