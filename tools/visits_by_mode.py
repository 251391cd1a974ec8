"""Visit counts per ray (interior nodes, instances entered, triangles, leaves) of a scene's bounce rays in trace modes 0 and 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import cuda_path_tracer_amd as pt
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 960, 540
for scene in ("party_armadillo", "helmet"):
    hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", scene + ".npz")).build(width=W, height=H, max_depth=8)
    ctx = pt.Context(0)
    hs.update_device(ctx)
    ctx.integrator_init(W, H)
    dev = torch.device("cuda", 0)
    tmp = torch.empty(W * H * 8, dtype=torch.float32, device=dev)
    chunks = []
    for it in range(2):
        for b in (1, 2, 3):
            n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
            chunks.append(tmp[: n * 8].clone())
    rays = torch.cat(chunks)
    n = rays.numel() // 8
    for m in (0, 2):
        ctx.set_trace_mode(m)
        ctr = torch.zeros(n * 4, dtype=torch.int32, device=dev)
        hh = torch.empty(n * 8, dtype=torch.float32, device=dev)
        ctx.trace_closest_counted(rays.data_ptr(), n, hh.data_ptr(), ctr.data_ptr())
        ctx.synchronize()
        c = ctr.view(n, 4).double().mean(0).cpu().numpy()
        print(f"{scene} mode {m}: {n} rays; per ray interior {c[0]:.2f} instances entered {c[1]:.2f} triangles {c[2]:.2f} leaves {c[3]:.2f}", flush=True)
    ctx.close()
