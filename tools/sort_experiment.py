#!/usr/bin/env python3
"""Does ordering the ray batch pay?  Times the closest-hit query on the bench's ray batch as captured
and after sorting it (with torch, outside the timing) by several keys built from the origin's cell
and the direction's octant.  Diagnostic: decides whether a sort inside the library is worth its cost."""
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
rays = torch.cat(chunks).view(-1, 8)
n = rays.shape[0]
hits = torch.empty(n * 8, dtype=torch.float32, device=dev)


def time_it(r, mode=0):
    ctx.set_trace_mode(mode)
    r = r.contiguous()
    torch.cuda.synchronize()  # the gather runs on torch's stream, the query on the context's
    for _ in range(3):
        ctx.trace_closest(r.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    ctx.reset_stats()
    ctx.set_profiling(True)
    for _ in range(10):
        ctx.trace_closest(r.data_ptr(), n, hits.data_ptr())
    ctx.synchronize()
    ctx.set_profiling(False)
    st = ctx.stats()
    ctx.set_trace_mode(0)
    return st.ms_closest / st.n_closest


def morton(q, bits):
    key = torch.zeros_like(q[:, 0])
    for b in range(bits):
        for a in range(3):
            key |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return key


o, d = rays[:, 0:3], rays[:, 4:7]
lo, hi = o.min(dim=0).values, o.max(dim=0).values
octant = ((d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2))
print(scene, n, "rays", flush=True)
print(f"as captured: mode 0 {time_it(rays):.3f} ms, mode 2 {time_it(rays, 2):.3f} ms", flush=True)
perm = torch.randperm(n, device=dev)
print(f"shuffled:    mode 0 {time_it(rays[perm]):.3f} ms", flush=True)
for bits in (3, 4, 5, 6, 8):
    q = ((o - lo) / (hi - lo).clamp_min(1e-20) * ((1 << bits) - 1e-3)).long().clamp_(0, (1 << bits) - 1)
    m = morton(q, bits)
    for name, key in (("cell", m), ("octant|cell", (octant << (3 * bits)) | m), ("cell|octant", (m << 3) | octant)):
        order = torch.argsort(key, stable=True)
        t0 = time_it(rays[order])
        print(f"{bits} bits/axis {name:12s}: mode 0 {t0:.3f} ms, mode 2 {time_it(rays[order], 2):.3f} ms", flush=True)
# direction-only keys
dq = ((d * 0.5 + 0.5) * (16 - 1e-3)).long().clamp_(0, 15)
order = torch.argsort(morton(dq, 4), stable=True)
print(f"direction cell 4 bits: mode 0 {time_it(rays[order]):.3f} ms", flush=True)
q = ((o - lo) / (hi - lo).clamp_min(1e-20) * (16 - 1e-3)).long().clamp_(0, 15)
dq = ((d * 0.5 + 0.5) * (4 - 1e-3)).long().clamp_(0, 3)
order = torch.argsort((morton(q, 4) << 6) | morton(dq, 2), stable=True)
print(f"cell 4 bits | direction 2 bits: mode 0 {time_it(rays[order]):.3f} ms", flush=True)
order = torch.argsort((morton(dq, 2) << 12) | morton(q, 4), stable=True)
print(f"direction 2 bits | cell 4 bits: mode 0 {time_it(rays[order]):.3f} ms", flush=True)
