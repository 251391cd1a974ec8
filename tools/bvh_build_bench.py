#!/usr/bin/env python3
"""Times pt_bvh_build (GPU) against the sequential host builder on the same inputs and checks
that both produce the same bytes.  Usage: python tools/bvh_build_bench.py [--big]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_path_tracer_amd as pt  # noqa: E402

KEYS = ("tlas_nodes", "blases", "blas_nodes", "tri_indices")


def soup(rs, n_tris):
    centre = (rs.rand(n_tris, 1, 3).astype(np.float32) - 0.5) * 40
    pos = ((rs.rand(n_tris, 3, 3).astype(np.float32) - 0.5) * 0.2 + centre).reshape(-1, 3)
    return np.ascontiguousarray(pos, np.float32), np.arange(n_tris * 3, dtype=np.uint32)


def geometry(meshes, instances):
    geo = np.zeros(len(instances), pt.abi.GEOMETRY)
    for k, (mesh, T) in enumerate(instances):
        first, count, verts = meshes[mesh]
        geo[k]["first_index"], geo[k]["index_count"], geo[k]["vertex_count"] = first, count, verts
        geo[k]["transform"] = np.asarray(T, np.float32).reshape(-1)
        geo[k]["blas_index"] = mesh
    return geo


def run(ctx, name, pos, idx, geo, host=True):
    ctx.bvh_build(pos, idx, geo)  # warm-up (first launches, allocator)
    t0 = time.perf_counter()
    dev = ctx.bvh_build(pos, idx, geo, timing=True)
    t_dev = time.perf_counter() - t0
    line = (f"{name}: {len(idx) // 3} triangles in the index array, {len(geo)} instances | device {t_dev * 1e3:.1f} ms wall "
            f"(BLAS {dev['timing']['ms_blas']:.2f} ms, TLAS {dev['timing']['ms_tlas']:.2f} ms, {dev['timing']['levels']} levels)")
    if host:
        t0 = time.perf_counter()
        ref = pt.bvh_build(pos, idx, geo)
        t_host = time.perf_counter() - t0
        same = all(np.array_equal(np.ascontiguousarray(dev[k]).view(np.uint8), np.ascontiguousarray(ref[k]).view(np.uint8)) for k in KEYS)
        line += f" | sequential {t_host * 1e3:.1f} ms | identical {same} | speed-up {t_host / t_dev:.1f}x"
    print(line, flush=True)


def main():
    big = "--big" in sys.argv
    ctx = pt.Context(0)
    eye = np.eye(4, dtype=np.float32)
    for name in ("teapot", "helmet"):
        arr = pt.SceneSpec.load(os.path.join(ROOT, "scenes", name + ".npz")).build(width=64, height=64, max_depth=4).arrays()
        run(ctx, name, arr["positions"], arr["indices"], arr["geometries"])
    rs = np.random.RandomState(1)
    for n in ((100_000, 1_000_000, 4_000_000) if big else (100_000, 1_000_000)):
        pos, idx = soup(rs, n)
        run(ctx, f"soup {n}", pos, idx, geometry({0: (0, len(idx), len(pos))}, [(0, eye)]))
    pos, idx = soup(rs, 64)
    for n_inst in ((2000, 8000, 32767) if big else (2000, 8000)):
        T = np.tile(eye, (n_inst, 1, 1))
        T[:, 3, :3] = rs.rand(n_inst, 3) * 400 - 200
        run(ctx, f"instances {n_inst}", pos, idx, geometry({0: (0, len(idx), len(pos))}, [(0, T[k]) for k in range(n_inst)]))
    ctx.close()


if __name__ == "__main__":
    main()
