"""Diagnostic: what pt_render_batch does when the path state of the passes in flight does not fit
in device memory (4K frame, large passes)."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
import cuda_path_tracer_amd as pt
ROOT = "/root/repo"
W, H = 3840, 2160
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "teapot.npz")).build(width=W, height=H, max_depth=4)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(W, H)
free0, total = torch.cuda.mem_get_info()
print("free GB before", free0 / 1e9)
try:
    for k in range(4):
        ctx.render_batch(W, H, k * 80, 80)   # 184 B x 8.3 M x 80 = 122 GB per pass in flight: two fit, the third does not
        f, _ = torch.cuda.mem_get_info()
        print("pass", k, "submitted, free GB", f / 1e9)
    ctx.synchronize()
    print("all passes ran")
except pt.PtError as e:
    print("PtError:", str(e)[:200])
free1, _ = torch.cuda.mem_get_info()
print("free GB after the three big passes", free1 / 1e9)
film = ctx.film_download(W, H)
print("film mean after 320 samples", float(film.mean()) / 320)
try:
    ctx.render_batch(W, H, 320, 250)   # one pass of 250 samples = 390 GB: cannot fit at all
    print("all passes ran")
except pt.PtError as e:
    print("PtError:", str(e)[:260])
ctx.render_batch(W, H, 320, 4)
ctx.synchronize()
print("small pass after the failure ok, film mean", float(ctx.film_download(W, H).mean()) / 324)
ctx.close()
