#!/usr/bin/env python3
"""One device BVH build of a synthetic triangle soup (for rocprofv3 --kernel-trace --stats).
Usage: python3 tools/bvh_build_profile.py [triangles] [instances]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_path_tracer_amd as pt  # noqa: E402

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_inst = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(1)
centre = (rs.rand(n_tris, 1, 3).astype(np.float32) - 0.5) * 40
pos = np.ascontiguousarray(((rs.rand(n_tris, 3, 3).astype(np.float32) - 0.5) * 0.2 + centre).reshape(-1, 3), np.float32)
idx = np.arange(n_tris * 3, dtype=np.uint32)
geo = np.zeros(n_inst, pt.abi.GEOMETRY)
T = np.tile(np.eye(4, dtype=np.float32), (n_inst, 1, 1))
if n_inst > 1:
    T[:, 3, :3] = rs.rand(n_inst, 3) * 400 - 200
geo["index_count"], geo["vertex_count"] = len(idx), len(pos)
geo["transform"] = T.reshape(n_inst, 16)
ctx = pt.Context(0)
for _ in range(3):
    out = ctx.bvh_build(pos, idx, geo, timing=True)
print(out["timing"])
ctx.close()
