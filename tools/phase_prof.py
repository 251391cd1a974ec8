#!/usr/bin/env python3
"""Diagnostic: SIMD-lane utilisation of each phase of the traversal loop on the bench's ray batch.
Needs a library built with -DPT_PHASE_PROF (cuda-path-tracer_amd/build.sh -DPT_PHASE_PROF)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_path_tracer_amd as pt  # noqa: E402

W, H = 1920, 1080
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz")).build(width=W, height=H)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
dev = torch.device("cuda", 0)
tmp = torch.empty(W * H * 8, dtype=torch.float32, device=dev)
chunks = []
# PT_BOUNCES=0 profiles the camera rays (coherent), the default 1,2,3 the bench's incoherent batch
for it in range(2):
    for b in [int(x) for x in os.environ.get("PT_BOUNCES", "1,2,3").split(",")]:
        n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
        chunks.append(tmp[: n * 8].clone())
rays = torch.cat(chunks)
n = rays.numel() // 8
hits = torch.empty(n * 8, dtype=torch.float32, device=dev)
mode = int(os.environ.get("PT_TRACE_MODE", "0"))
ctx.set_trace_mode(mode)
ctr = torch.zeros(n * 4, dtype=torch.int32, device=dev)
ctx.trace_closest_counted(rays.data_ptr(), n, hits.data_ptr(), ctr.data_ptr())
ctx.synchronize()
c = ctr.view(-1, 4).to(torch.int64).sum(dim=0).cpu().numpy() / n
print(f"trace mode {mode}: per ray interior {c[0]:.2f}, instances entered {c[1]:.2f}, triangles {c[2]:.2f}, leaves {c[3]:.2f}")
lib = pt.abi.load_library()
out = (C.c_ulonglong * 16)()
ctx.synchronize()
lib.pt_debug_phase_prof(out)
ctx.trace_closest(rays.data_ptr(), n, hits.data_ptr())
ctx.synchronize()
lib.pt_debug_phase_prof(out)
names = ["interior round", "instance enter", "triangle round", "pop round", "instance exit", "refill", "step (busy lanes)", "outer round"]
print(f"{n} rays")
tot_l = tot_w = 0
for k, name in enumerate(names):
    l, w = out[k], out[8 + k]
    print(f"{name:20s} lane-rounds {l:12d}  wave-rounds {w:10d}  utilisation {l / max(1, 64 * w):.3f}  per ray {l / n:.2f}")
