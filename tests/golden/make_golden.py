#!/usr/bin/env python3
"""Writes tests/golden/*.npz: small input/output vectors produced by the CPU oracle
(oracle/liboracle.so) on scenes this repository builds itself.  They freeze the oracle's answers
so that a later change to the oracle, the host facade or the kernels shows up as a diff against
committed data.  The reference ships no vectors for this path, so none are derived from it.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cuda_path_tracer_amd as pt  # noqa: E402
import oracle_lib as ora  # noqa: E402
from conftest import make_tiny_scene  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def traversal_case(name, hs, n):
    desc = hs.desc()
    rays = ora.generate_incoherent_rays(desc, n, 0)
    # aim half of them at points of the central part of the scene's floor so that most of those hit
    lo, hi = rays["origin"].min(axis=0), rays["origin"].max(axis=0)  # origins fill the scene box
    rng = np.random.default_rng(7)
    target = (lo + (hi - lo) * rng.uniform(0.3, 0.7, (len(rays[::2]), 3))).astype(np.float32)
    target[:, 1] = lo[1] + np.float32(0.1) * (hi[1] - lo[1])
    d = target - rays["origin"][::2]
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6).astype(np.float32)
    rays["direction"][::2] = d.astype(np.float32)
    hits, counters = ora.trace_closest(desc, rays, nthreads=4, counters=True)
    short = rays.copy()
    k = np.arange(len(rays)) % 3
    short["tmax"] = np.where(hits["t"] > 0, hits["t"] * np.where(k == 0, np.float32(1.01), np.float32(0.99)), np.float32(2.0)).astype(np.float32)
    short["tmax"][k == 2] = np.float32(3.4028235e38)
    occluded = ora.trace_any(desc, short, nthreads=4)
    np.savez_compressed(os.path.join(OUT, f"traversal_{name}.npz"), rays=rays.view(np.uint8), hits=hits.view(np.uint8),
                        counters=counters, short_tmax=short["tmax"], occluded=occluded)
    print(name, "rays", n, "hit fraction", float((hits["t"] > 0).mean()), "occluded", float(occluded.mean()))


def main():
    os.makedirs(OUT, exist_ok=True)
    tiny = make_tiny_scene(pt)
    traversal_case("tiny", tiny, 4096)
    teapot = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "teapot.npz")).build(width=64, height=64, max_depth=6)
    traversal_case("teapot", teapot, 4096)

    film, counts = ora.render(tiny.desc(), 64, 48, 0, nthreads=4)
    film2, _ = ora.render(tiny.desc(), 64, 48, 1, film.copy(), nthreads=4)
    np.savez_compressed(os.path.join(OUT, "render_tiny.npz"), film_iter0=film, film_iter01=film2, ray_counts=np.array(counts, np.int64),
                        rgba_iter1=ora.tonemap(film2, iteration=1))
    print("render tiny mean", float(film.mean()), counts)

    # debug integrator: every BVH map (scale 60) and every attribute view of the tiny scene, 8-bit
    debug = {}
    for k, name in enumerate(("bounds_hit", "tlas", "triangles_hit")):
        debug["bvh_" + name] = ora.debug_render(tiny.desc(), 64, 48, pt.abi.IntegratorState(1, k, 60, 0), nthreads=4)[1]
    for k, name in enumerate(("normals", "albedo", "uv", "depth", "metallic", "roughness"), start=1):
        debug["view_" + name] = ora.debug_render(tiny.desc(), 64, 48, pt.abi.IntegratorState(0, 0, 500, k), nthreads=4)[1]
    np.savez_compressed(os.path.join(OUT, "debug_tiny.npz"), **debug)

    streams = np.stack([ora.pcg_stream(px, py, it, 16) for px, py, it in ((0, 0, 0), (5, 7, 0), (1919, 1079, 3), (65535, 65535, 1000))])
    weights = np.array([0.5, 3.0, 1.0, 0.0, 2.5, 1.0, 1.0, 7.0], np.float32)
    _, prob, alias = ora.alias_build(weights)
    np.savez_compressed(os.path.join(OUT, "sampling.npz"), pcg_pixels=np.array([[0, 0, 0], [5, 7, 0], [1919, 1079, 3], [65535, 65535, 1000]]),
                        pcg_streams=streams, alias_weights=weights, alias_table=alias, alias_prob=prob)


if __name__ == "__main__":
    main()
