import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_path_tracer_amd as pt
w, h = 1920, 1080
hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz")).build(width=w, height=h)
ctx = pt.Context(0); ctx.upload(hs.desc()); ctx.integrator_init(w, h)
buf = torch.empty(w * h * 8, dtype=torch.float32, device="cuda:0")
out = []
for rep in range(3):
    out.append([ctx.capture_bounce_rays(w, h, 0, b, buf.data_ptr(), w * h) for b in (1, 2, 3)])
print(os.path.basename(os.environ.get("PTAMD_LIB", "libptamd.so")), out, "oracle [481126, 98630, 40630]")
