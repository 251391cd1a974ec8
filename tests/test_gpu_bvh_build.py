"""pt_bvh_build (BLAS + TLAS build on the GPU) against the sequential builders: the host
facade's (pth_bvh_build) and the oracle's (ora_tlas_build), both restatements of bvh.cpp:5-332.

The bar is bit-exact for all four arrays: TLAS nodes, DeviceBLAS records, flattened BLAS nodes
(boxes, child / first-primitive indices, counts: the reference's depth-first node numbering)
and the triangle permutation."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("tlas_nodes", "blases", "blas_nodes", "tri_indices")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _same(a, b, what=""):
    for k in KEYS:
        assert len(a[k]) == len(b[k]), (what, k, len(a[k]), len(b[k]))
        if not np.array_equal(_bits(a[k]), _bits(b[k])):
            bad = np.flatnonzero((_bits(a[k]).reshape(len(a[k]), -1) != _bits(b[k]).reshape(len(b[k]), -1)).any(axis=1))
            raise AssertionError(f"{what}: {k} differs at {len(bad)} of {len(a[k])} records, first {bad[:5]}: {a[k][bad[0]]} vs {b[k][bad[0]]}")


def _geometry(pt, meshes, instances):
    """meshes: list of (first_index, index_count, vertex_count); instances: list of (mesh id, 4x4 column-major)."""
    geo = np.zeros(len(instances), pt.abi.GEOMETRY)
    for k, (mesh, T) in enumerate(instances):
        first, count, verts = meshes[mesh]
        geo[k]["first_index"], geo[k]["index_count"], geo[k]["vertex_count"] = first, count, verts
        geo[k]["transform"] = np.asarray(T, np.float32).reshape(-1)
        geo[k]["blas_index"] = mesh
    return geo


def _soup(rs, n_tris, spread=1.0, quantise=None):
    pos = (rs.rand(n_tris * 3, 3).astype(np.float32) - 0.5) * np.float32(spread)
    centre = (rs.rand(n_tris, 1, 3).astype(np.float32) - 0.5) * 4
    pos = (pos.reshape(n_tris, 3, 3) * 0.2 + centre).reshape(-1, 3).astype(np.float32)
    if quantise:
        pos = (np.round(pos * quantise) / quantise).astype(np.float32)
    return pos, np.arange(n_tris * 3, dtype=np.uint32)


@pytest.mark.parametrize("name", ["teapot", "cornell", "pbr_metallic", "sphere", "instancing_stress", "helmet"])
def test_device_build_equals_sequential_builders(pt, ora, gpu_ctx, scene_factory, name):
    arr = scene_factory(name, 64, 64, 4).arrays()
    dev = gpu_ctx.bvh_build(arr["positions"], arr["indices"], arr["geometries"], timing=True)
    _same(dev, arr, name + " vs host facade")
    _same(dev, ora.tlas_build(arr["positions"], arr["indices"], arr["geometries"]), name + " vs oracle")
    assert dev["timing"]["levels"] >= 1


def test_device_build_random_soups_and_degenerate_inputs(pt, ora, gpu_ctx):
    """Random triangle soups: coarse coordinate grids make equal centroids, equal SAH costs, planes
    that leave one side empty and boxes that mix +0 and -0; several meshes of different sizes are
    built in one call, with an unused blas_index in between."""
    rs = np.random.RandomState(11)
    eye = np.eye(4, dtype=np.float32)
    for case in range(12):
        sizes = [int(rs.randint(1, 6)), int(rs.randint(3, 400)), int(rs.randint(500, 5000))]
        quant = [None, 2, 4, 16][case % 4]
        pos_list, idx_list, meshes, base_v, base_i = [], [], {}, 0, 0
        ids = [0, 1, 3]  # blas_index 2 is never named
        for mesh_id, n in zip(ids, sizes):
            p, i = _soup(rs, n, quantise=quant)
            if case % 3 == 0:
                p[rs.rand(len(p)) < 0.2] *= np.float32(-0.0)  # exact zeros of both signs
            pos_list.append(p)
            idx_list.append(i + base_v)
            meshes[mesh_id] = (base_i, len(i), len(p))
            base_v += len(p)
            base_i += len(i)
        pos = np.concatenate(pos_list)
        idx = np.concatenate(idx_list).astype(np.uint32)
        inst = []
        for k in range(int(rs.randint(3, 40))):
            T = eye.copy()
            T[3, :3] = rs.rand(3) * 10 - 5          # column-major: row 3 of the array = translation column
            T[0, 0], T[1, 1], T[2, 2] = rs.rand(3) + 0.5
            inst.append((ids[k % 3] if k >= 3 else ids[k], T))
        geo = _geometry(pt, meshes, inst)
        dev = gpu_ctx.bvh_build(pos, idx, geo)
        _same(dev, pt.bvh_build(pos, idx, geo), f"soup {case}")
        assert len(dev["blases"]) == 4 and dev["blases"][2]["node_count"] == 0


def test_scene_finalize_on_device_gives_the_same_scene(pt, gpu_ctx):
    """Scene::Finalize through the facade with the device builder: every array of the description,
    light tables included, equals the host-built scene's."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = pt.SceneSpec.load(os.path.join(root, "scenes", "helmet.npz"))
    host = spec.build(width=64, height=64, max_depth=4).arrays()
    dev = spec.build(width=64, height=64, max_depth=4, ctx=gpu_ctx).arrays()
    for k, v in host.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(_bits(v), _bits(dev[k])), k
        else:
            assert v == dev[k], k


def test_device_build_small_and_error_cases(pt, gpu_ctx):
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    eye = np.eye(4, dtype=np.float32)
    for reps in (1, 2, 3, 5):  # identical centroids: no split plane exists
        idx = np.array([0, 1, 2] * reps, np.uint32)
        geo = _geometry(pt, {0: (0, len(idx), 3)}, [(0, eye)])
        dev = gpu_ctx.bvh_build(tri, idx, geo)
        _same(dev, pt.bvh_build(tri, idx, geo), f"{reps} coincident triangles")
        assert len(dev["blas_nodes"]) == 1 and dev["blas_nodes"][0]["primitive_count"] == reps and len(dev["tlas_nodes"]) == 2
    with pytest.raises(pt.PtError, match="at least one"):
        gpu_ctx.bvh_build(tri, np.array([0, 1, 2], np.uint32), np.zeros(0, pt.abi.GEOMETRY))
    with pytest.raises(pt.PtError, match="empty mesh"):
        gpu_ctx.bvh_build(tri, np.array([0, 1, 2], np.uint32), _geometry(pt, {0: (0, 2, 3)}, [(0, eye)]))
    with pytest.raises(pt.PtError, match="out of range"):
        gpu_ctx.bvh_build(tri, np.array([0, 1, 7], np.uint32), _geometry(pt, {0: (0, 3, 3)}, [(0, eye)]))
    with pytest.raises(pt.PtError, match="index range"):
        gpu_ctx.bvh_build(tri, np.array([0, 1, 2], np.uint32), _geometry(pt, {0: (0, 6, 3)}, [(0, eye)]))


def test_device_built_scene_renders_like_the_host_built_one(pt, gpu_ctx, scene_factory):
    """The arrays go straight into pt_scene_desc: a scene assembled from the device build traces the
    same hits as the facade's own scene."""
    hs = scene_factory("teapot", 64, 64, 4)
    arr = hs.arrays()
    dev = gpu_ctx.bvh_build(arr["positions"], arr["indices"], arr["geometries"])
    desc = pt.abi.SceneDesc.from_buffer_copy(hs.desc())
    keep = {k: np.ascontiguousarray(dev[k]) for k in KEYS}
    desc.tlas_nodes = keep["tlas_nodes"].ctypes.data
    desc.blases = keep["blases"].ctypes.data
    desc.blas_nodes = keep["blas_nodes"].ctypes.data
    desc.tri_indices = pt.abi.uptr(keep["tri_indices"])
    rs = np.random.RandomState(5)
    rays = np.zeros(4096, pt.abi.RAY)
    rays["origin"] = rs.rand(4096, 3) * 8 - 4
    d = rs.randn(4096, 3)
    rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["tmax"] = np.float32(3.4028235e38)
    gpu_ctx.upload(hs.desc())
    a = gpu_ctx.trace_closest_host(rays)
    gpu_ctx.upload(desc)
    b = gpu_ctx.trace_closest_host(rays)
    assert np.array_equal(_bits(a), _bits(b)) and (a["t"] > 0).mean() > 0.2


@pytest.mark.parametrize("n_inst,lattice", [(2, 1), (3, 1), (7, 2), (500, 4), (3000, 8), (3000, 64), (5000, 1000)])
def test_device_tlas_pairing_with_exact_ties(pt, gpu_ctx, n_inst, lattice):
    """Instances of one mesh translated onto a coarse integer lattice: many pairs have exactly the
    same merged surface area (and many instances coincide), so the pairing order is decided by the
    slot-number tie rule all the way; any slip in the cached partner answers changes the tree."""
    rs = np.random.RandomState(n_inst + lattice)
    pos, idx = _soup(rs, 16, quantise=4)
    T = np.tile(np.eye(4, dtype=np.float32), (n_inst, 1, 1))
    T[:, 3, :3] = rs.randint(0, lattice, size=(n_inst, 3)).astype(np.float32) * 8
    geo = _geometry(pt, {0: (0, len(idx), len(pos))}, [(0, T[k]) for k in range(n_inst)])
    _same(gpu_ctx.bvh_build(pos, idx, geo), pt.bvh_build(pos, idx, geo), f"{n_inst} instances on a {lattice}^3 lattice")


def test_device_build_many_instances_is_faster_than_the_sequential_build(pt, gpu_ctx):
    """8000 instances of three small meshes: the pairing loop is the sequential builder's O(n^2) part."""
    rs = np.random.RandomState(3)
    pos_list, idx_list, meshes, base_v, base_i = [], [], {}, 0, 0
    for mesh_id, n in enumerate((12, 200, 2000)):
        p, i = _soup(rs, n)
        pos_list.append(p)
        idx_list.append(i + base_v)
        meshes[mesh_id] = (base_i, len(i), len(p))
        base_v += len(p)
        base_i += len(i)
    pos, idx = np.concatenate(pos_list), np.concatenate(idx_list).astype(np.uint32)
    n_inst = 8000
    T = np.tile(np.eye(4, dtype=np.float32), (n_inst, 1, 1))
    T[:, 3, :3] = rs.rand(n_inst, 3) * 400 - 200
    geo = _geometry(pt, meshes, [(k % 3, T[k]) for k in range(n_inst)])
    t0 = time.perf_counter()
    dev = gpu_ctx.bvh_build(pos, idx, geo, timing=True)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    host = pt.bvh_build(pos, idx, geo)
    t_host = time.perf_counter() - t0
    _same(dev, host, "8000 instances")
    print(f"device build {t_dev * 1e3:.1f} ms (BLAS {dev['timing']['ms_blas']:.2f} ms, TLAS {dev['timing']['ms_tlas']:.2f} ms, "
          f"{dev['timing']['levels']} levels), sequential build {t_host * 1e3:.1f} ms")
    assert t_dev < t_host
