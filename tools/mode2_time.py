#!/usr/bin/env python3
"""Kernel time of the closest-hit query on the bench's ray batch (helmet 1080p, bounce rays of 8
iterations) in trace modes 0 and 2; one line per library named in PTAMD_LIB (diagnostic variants)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_path_tracer_amd as pt  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "helmet"
W, H = 1920, 1080
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", scene + ".npz")).build(width=W, height=H, max_depth=8)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
dev = torch.device("cuda", 0)
tmp = torch.empty(W * H * 8, dtype=torch.float32, device=dev)
chunks = []
for it in range(int(os.environ.get("ITERS", "8"))):
    for b in (1, 2, 3):
        n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
rays = torch.cat(chunks)
n = rays.numel() // 8
hits = torch.empty(n * 8, dtype=torch.float32, device=dev)
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
print(os.path.basename(os.environ.get("PTAMD_LIB", "libptamd.so")), scene, n, "rays;", "; ".join(res), flush=True)
