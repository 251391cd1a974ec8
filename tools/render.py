#!/usr/bin/env python3
"""Headless render on one GPU: progressive PNGs, a film checkpoint, and resuming from it.  The scene is
either a scene file of the reference (scene JSON with its OBJ / glTF assets beside it), read by the C++
front end (pth_scene_load; Python only lends it PIL for image files), or a cooked scene
(scenes/<name>.npz) for machines that do not have the asset files.

    python tools/render.py /path/to/Scenes/teapot/teapot.json --iterations 256 --out out/teapot
    python tools/render.py helmet --iterations 256 --save-every 64 --out out/helmet --checkpoint out/helmet.film
    python tools/render.py helmet --iterations 1024 --out out/helmet --checkpoint out/helmet.film --resume
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_path_tracer_amd as pt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", help="a scene .json file, a name under scenes/, or a path to a cooked .npz")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--depth", type=int)
    ap.add_argument("--iterations", type=int, default=0, help="0 = the scene's own iteration count")
    ap.add_argument("--batch", type=int, default=0, help="samples traced per pass (0 = chosen from the iteration count)")
    ap.add_argument("--save-every", type=int, default=0)
    ap.add_argument("--out", default="render", help="output prefix: <out>.<n>samp.png")
    ap.add_argument("--checkpoint")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--device-bvh", action="store_true", help="build the acceleration structures on the GPU")
    a = ap.parse_args()
    path = a.scene if os.path.exists(a.scene) else os.path.join(ROOT, "scenes", a.scene + ".npz")
    ctx = pt.Context(a.device)
    if path.endswith(".json"):
        hs = pt.HostScene.load(path, a.width or 0, a.height or 0, a.depth or 0, ctx=ctx if a.device_bvh else None)
    else:
        hs = pt.SceneSpec.load(path).build(width=a.width, height=a.height, max_depth=a.depth, ctx=ctx if a.device_bvh else None)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    r = pt.render_offscreen(ctx, hs, a.iterations, a.batch, a.save_every, a.out, a.checkpoint, a.resume)
    n = r["iterations_done"] - r["first_iteration"]
    print(f"{r['iterations_done']} samples per pixel in the film ({n} rendered now in {r['seconds']:.2f} s, "
          f"{n / max(r['seconds'], 1e-9):.1f} spp/s), {r['images_written']} image(s) written")
    ctx.close()


if __name__ == "__main__":
    main()
