#!/usr/bin/env python3
"""Cooks the reference's scene files into scenes/*.npz (SceneSpec containers) so that GPU hosts,
which have neither the reference checkout nor its assets, can build the same scenes.

Run in the development container only:  python tools/cook_scenes.py [/root/reference/Scenes]
Geometry is stored as parsed; textures are stored as their original encoded bytes (decoded
with PIL at build time).  `party_armadillo` is the reference's scripts/armadillo.py (20 000 instances)
with a seed and stand-in meshes; `instancing_stress` is a smaller scene of the same kind that replaces the party / armadillo scenes whose
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


def party_armadillo(src_dir: str, n: int = 10000, seed: int = 20250829) -> "pt.SceneSpec":
    """scripts/armadillo.py:4-90 of the reference restated with a seeded generator: the same scene
    JSON (camera, three materials, two named instance bases, then n x {dragon, armadillo} instances
    with TRANS uniform in [-10,10]^3, ROTAT uniform in [0,360)^3, uniform scales 0.15..0.5 and
    0.05..0.3, drawn in the script's order), loaded through the scene-JSON reader.  The dragon and
    armadillo meshes are absent from the reference checkout: the DamagedHelmet mesh stands in for
    the dragon (mirror) and the helmet scene's sphere mesh for the armadillo (diffuse red).  The
    script defines a "light" material but places no emitter, so its image is black; one emissive
    sphere (scale 2 at (0, 3, 0)) is added here so that the render exercises shading - that object is
    this project's addition."""
    import json
    import random
    import tempfile

    rnd = random.Random(seed)
    scene = {
        "integrator": {"TYPE": "path", "DEPTH": 8, "ITERATIONS": 8124},
        "camera": {"TYPE": "pinhole", "FOVY": 45.0, "EYE": [0.0, 0.0, -5.0], "LOOKAT": [0.0, 0.0, 0.0], "UP": [0.0, 1.0, 0.0],
                   "film": {"WIDTH": 1920, "HEIGHT": 1080, "COLOR_SPACE": "rgb"}, "sampler": {"TYPE": "independent"},
                   "reconstruction_filter": {"TYPE": "box"}},
        "materials": {"light": {"TYPE": "emitting", "RADIANCE": {"RGB": [1.0, 1.0, 1.0], "EMITTANCE": 5.0}},
                      "mirror": {"TYPE": "mirror", "ALBEDO": {"RGB": [0.98, 0.98, 0.98]}},
                      "diffuse_red": {"TYPE": "diffuse", "ALBEDO": {"RGB": [0.86, 0.35, 0.35]}}},
        "objects": [
            {"TYPE": "instance", "NAME": "armadillo",
             "INSTANCE": {"TYPE": "gltf", "MATERIAL": "diffuse_red", "PATH": os.path.join(src_dir, "helmet", "sphere", "sphere.gltf"), "FACE_NORMALS": False}},
            {"TYPE": "instance", "NAME": "dragon",
             "INSTANCE": {"TYPE": "gltf", "MATERIAL": "mirror", "PATH": os.path.join(src_dir, "helmet", "helmet", "DamagedHelmet.gltf"), "FACE_NORMALS": False}},
            {"TYPE": "gltf", "MATERIAL": "light", "PATH": os.path.join(src_dir, "helmet", "sphere", "sphere.gltf"), "TRANS": [0.0, 3.0, 0.0], "SCALE": [2.0, 2.0, 2.0]},
        ],
    }
    for _ in range(n):
        scale = rnd.uniform(0.05, 0.3)
        scale2 = rnd.uniform(0.15, 0.5)
        scene["objects"].append({"TYPE": "instance", "SOURCE": "dragon", "TRANS": [rnd.uniform(-10, 10) for _ in range(3)],
                                 "ROTAT": [rnd.uniform(0, 360) for _ in range(3)], "SCALE": [scale2, scale2, scale2]})
        scene["objects"].append({"TYPE": "instance", "SOURCE": "armadillo", "TRANS": [rnd.uniform(-10, 10) for _ in range(3)],
                                 "ROTAT": [rnd.uniform(0, 360) for _ in range(3)], "SCALE": [scale, scale, scale]})
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "party_armadillo.json")
        with open(path, "w") as f:
            json.dump(scene, f)
        return pt.load_scene_json(path)


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
    spec = party_armadillo(src)
    spec.save(os.path.join(out, "party_armadillo.npz"))
    print("party_armadillo", len([o for o in spec.ops if o["op"] == "instance"]), "instances ->",
          os.path.getsize(os.path.join(out, "party_armadillo.npz")) // 1024, "KiB")
    spec = instancing_stress(src)
    spec.save(os.path.join(out, "instancing_stress.npz"))
    print("instancing_stress", len([o for o in spec.ops if o["op"] == "instance"]), "instances ->",
          os.path.getsize(os.path.join(out, "instancing_stress.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
