"""Experiment: one closest-hit launch over a batch against two concurrent launches over its halves (two contexts on
one GPU), back to back and with a wait after every query."""
import os, sys, time
ROOT0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT0)
import torch
import cuda_path_tracer_amd as pt
ROOT = ROOT0
W, H = 1920, 1080
spec = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz"))
ctxs = [pt.Context(0) for _ in range(2)]
for c in ctxs:
    spec.build(width=W, height=H).update_device(c)
ctx = ctxs[0]
ctx.integrator_init(W, H)
dev = torch.device("cuda", 0)
tmp = torch.empty(W * H * 8, dtype=torch.float32, device=dev)
chunks = []
for it in range(8):
    for b in (1, 2, 3):
        n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
rays = torch.cat(chunks)
n = rays.numel() // 8
hits = torch.empty(n * 8, dtype=torch.float32, device=dev)
for mode in (0, 2):
    for c in ctxs:
        c.set_trace_mode(mode)
    def one():
        ctx.trace_closest(rays.data_ptr(), n, hits.data_ptr())
    def two(parts=2):
        h = n // 2
        ctxs[0].trace_closest(rays.data_ptr(), h, hits.data_ptr())
        ctxs[1].trace_closest(rays.data_ptr() + h * 32, n - h, hits.data_ptr() + h * 32)
    for name, fn in (("one launch", one), ("two contexts, half each", two)):
        for _ in range(3):
            fn()
        for c in ctxs:
            c.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        for c in ctxs:
            c.synchronize()
        dt = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
            for c in ctxs:
                c.synchronize()
        dts = (time.perf_counter() - t0) / 20
        print(f"mode {mode} {name}: back to back {dt * 1e3:.3f} ms  {n / dt / 1e9:.3f} Grays/s; waited for after every query {dts * 1e3:.3f} ms  {n / dts / 1e9:.3f} Grays/s", flush=True)
