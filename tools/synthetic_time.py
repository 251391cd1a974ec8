#!/usr/bin/env python3
"""Kernel time of the closest-hit query on synthetic incoherent rays (no render pipeline involved), trace modes 0 and 2."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_path_tracer_amd as pt  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "helmet"
n = 1 << 22
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", scene + ".npz")).build(width=64, height=64, max_depth=8)
ctx = pt.Context(0)
hs.update_device(ctx)
dev = torch.device("cuda", 0)
rays = torch.empty(n * 8, dtype=torch.float32, device=dev)
hits = torch.empty(n * 8, dtype=torch.float32, device=dev)
ctx.generate_incoherent_rays(rays.data_ptr(), n, 0)
res = []
for mode in (0, 2):
    ctx.set_trace_mode(mode)
    for _ in range(3):
        ctx.trace_closest(rays.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    ctx.reset_stats()
    ctx.set_profiling(True)
    for _ in range(10):
        ctx.trace_closest(rays.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    ctx.set_profiling(False)
    st = ctx.stats()
    ms = st.ms_closest / st.n_closest
    res.append(f"mode {mode}: {ms:.3f} ms {n / ms / 1e6:.3f} Grays/s")
print(os.path.basename(os.environ.get("PTAMD_LIB", "libptamd.so")), scene, n, "synthetic rays;", "; ".join(res), flush=True)
