#!/usr/bin/env python3
"""Cooks the reference's scene files into scenes/*.npz (SceneSpec containers) so that GPU hosts,
which have neither the reference checkout nor its assets, can build the same scenes.

Run in the development container only:  python tools/cook_scenes.py [/root/reference/Scenes]
Geometry is stored as parsed; textures are stored as their original encoded bytes (decoded
with PIL at build time).  `instancing_stress` replaces the party / armadillo scenes whose
meshes are absent from the reference checkout: a seeded grid of teapot instances in the
cornell room, generated here (scripts/armadillo.py of the reference is unseeded).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_path_tracer_amd as pt  # noqa: E402
from cuda_path_tracer_amd import scene_io  # noqa: E402


def instancing_stress(src_dir: str, n_side: int = 24, seed: int = 7) -> "pt.SceneSpec":
    """Teapot room with n_side^2 * 2 extra instanced teapots (shared BLAS), seeded transforms."""
    spec = pt.load_scene_json(os.path.join(src_dir, "teapot", "teapot.json"))
    # the first teapot instance op is the template
    teapots = [o for o in spec.ops if o["op"] == "instance" and o.get("index_count", 0) > 1000]
    base = teapots[0]
    rng = np.random.RandomState(seed)
    mats = [o for o in spec.ops if o["op"] == "material"]
    n_mat = len(mats) + 1
    for layer in range(2):
        for i in range(n_side):
            for k in range(n_side):
                t = [-4.5 + 9.0 * (i + 0.5) / n_side, 0.6 + 7.5 * rng.rand() * (layer + 1) / 2, -4.5 + 9.0 * (k + 0.5) / n_side]
                r = [0.0, float(rng.rand() * 360.0), 0.0]
                s = [0.012 + 0.01 * rng.rand()] * 3
                o = dict(base)
                o["transform"] = [float(x) for x in pt.transform_matrix(t, r, s)]
                o["material"] = int(1 + rng.randint(n_mat - 1))
                if mats[o["material"] - 1]["type"] == pt.abi.PT_MAT_EMISSIVE:
                    o["material"] = 0
                spec.ops.append(o)
    spec.max_depth = 8
    return spec


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Scenes"
    out = os.path.join(ROOT, "scenes")
    os.makedirs(out, exist_ok=True)
    jobs = {
        "teapot": "teapot/teapot.json",
        "helmet": "helmet/helmet.json",
        "cornell": "cornell.json",
        "pbr_metallic": "pbr_metallic.json",
        "pbr_ior": "pbr_ior.json",
        "sphere": "sphere.json",
    }
    for name, rel in jobs.items():
        spec = pt.load_scene_json(os.path.join(src, rel))
        spec.save(os.path.join(out, name + ".npz"))
        print(name, "ops", len(spec.ops), "arrays", len(spec.arrays), "->", os.path.getsize(os.path.join(out, name + ".npz")) // 1024, "KiB")
    spec = instancing_stress(src)
    spec.save(os.path.join(out, "instancing_stress.npz"))
    print("instancing_stress", len([o for o in spec.ops if o["op"] == "instance"]), "instances ->",
          os.path.getsize(os.path.join(out, "instancing_stress.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
