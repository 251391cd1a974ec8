import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pt():
    import cuda_path_tracer_amd as pt_mod

    pt_mod.abi.load_library()
    return pt_mod


@pytest.fixture(scope="session")
def ora():
    import oracle_lib

    oracle_lib.lib()
    return oracle_lib


_SCENE_CACHE = {}


@pytest.fixture(scope="session")
def scene_factory(pt):
    """Builds (and caches) a HostScene from scenes/<name>.npz at a given resolution / depth."""

    def make(name, width=None, height=None, max_depth=None):
        key = (name, width, height, max_depth)
        if key not in _SCENE_CACHE:
            spec = pt.SceneSpec.load(os.path.join(ROOT, "scenes", name + ".npz"))
            _SCENE_CACHE[key] = spec.build(width=width, height=height, max_depth=max_depth)
        return _SCENE_CACHE[key]

    return make


@pytest.fixture(scope="session")
def tiny_scene(pt):
    return make_tiny_scene(pt)


def make_tiny_scene(pt):
    """Hand-made scene: a lit quad light above a floor quad, a tilted cube and an instanced,
    non-uniformly scaled copy of it; one material of every kind; checkerboard floor."""
    hs = pt.HostScene()
    hs.set_integrator(6, 16)
    hs.set_camera([0.0, 2.0, 6.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0], 50.0, 64, 48)
    chk = hs.add_checker_texture3([0.8, 0.8, 0.8], [0.2, 0.3, 0.8], scale=(6.0, 6.0))
    red = hs.add_constant_texture3([0.8, 0.2, 0.2])
    white = hs.add_constant_texture3([0.9, 0.9, 0.9])
    rough = hs.add_constant_texture1(0.35)
    metal = hs.add_constant_texture1(1.0)
    m_floor = hs.add_material(pt.abi.PT_MAT_LAMBERTIAN, chk)
    m_light = hs.add_material(pt.abi.PT_MAT_EMISSIVE, white, np.float32(12.0))
    m_mr = hs.add_material(pt.abi.PT_MAT_METALLIC_ROUGHNESS, red, int(metal), int(rough))
    m_glass = hs.add_material(pt.abi.PT_MAT_DIELECTRIC, white, np.float32(1.45))
    m_mirror = hs.add_material(pt.abi.PT_MAT_MIRROR, white)
    quad_p = np.array([[-1, 0, 1], [1, 0, 1], [-1, 0, -1], [1, 0, -1]], np.float32)
    quad_n = np.tile(np.array([[0, 1, 0]], np.float32), (4, 1))
    quad_t = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], np.float32)
    quad_i = np.array([0, 1, 2, 1, 3, 2], np.uint32)
    q_first = hs.add_mesh(quad_p, quad_n, quad_t, quad_i)
    cp, cn, ct, ci = pt.unit_cube()
    c_first = hs.add_mesh(cp, cn, ct, ci)
    sp, sn, st, si = pt.uv_sphere(12, 24)
    s_first = hs.add_mesh(sp, sn, st, si)
    b_quad, b_cube, b_sphere, b_quad2 = (hs.new_blas_index() for _ in range(4))
    hs.add_instance(q_first, 6, 4, m_floor, pt.transform_matrix([0, 0, 0], [0, 0, 0], [4, 1, 4]), b_quad)
    hs.add_instance(q_first, 6, 4, m_light, pt.transform_matrix([0, 3.5, 0], [180, 0, 0], [0.8, 1, 0.8]), b_quad2)
    hs.add_instance(c_first, len(ci), len(cp), m_mr, pt.transform_matrix([-1.2, 0.6, 0], [0, 30, 0], [1, 1.2, 1]), b_cube)
    hs.add_instance(c_first, len(ci), len(cp), m_mirror, pt.transform_matrix([1.5, 0.4, -0.8], [10, -20, 5], [0.8, 0.8, 1.6]), b_cube)
    hs.add_instance(s_first, len(si), len(sp), m_glass, pt.transform_matrix([0.2, 0.7, 1.0], [0, 0, 0], [1.2, 1.2, 1.2]), b_sphere)
    hs.finalize()
    return hs


def rays_from_arrays(pt, origins, directions, tmax=None):
    n = len(origins)
    r = np.zeros(n, pt.abi.RAY)
    r["origin"] = np.asarray(origins, np.float32)
    r["direction"] = np.asarray(directions, np.float32)
    r["tmax"] = np.float32(3.4028235e38) if tmax is None else np.asarray(tmax, np.float32)
    return r


@pytest.fixture(scope="session")
def _session_gpu_ctx(pt):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = pt.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture()
def gpu_ctx(_session_gpu_ctx):
    """One context for the whole session; every test starts with its counters at zero (some tests overflow the
    traversal stack on purpose, and the drop counter would otherwise leak into whichever test runs next)."""
    _session_gpu_ctx.reset_stats()
    return _session_gpu_ctx


class DeviceBuf:
    """A torch-owned device buffer viewed as an array of C structs."""

    def __init__(self, torch, nbytes):
        self.t = torch.zeros(max(4, (nbytes + 3) // 4), dtype=torch.int32, device="cuda:0")
        self.nbytes = nbytes

    @property
    def ptr(self):
        return self.t.data_ptr()

    def upload(self, torch, arr):
        b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        pad = (-len(b)) % 4
        if pad:
            b = np.concatenate([b, np.zeros(pad, np.uint8)])
        self.t[: len(b) // 4].copy_(torch.from_numpy(b.view(np.int32).copy()))

    def download(self, dtype, count):
        raw = self.t.cpu().numpy().view(np.uint8)[: count * np.dtype(dtype).itemsize]
        return raw.view(dtype).copy()


@pytest.fixture(scope="session")
def devbuf():
    import torch

    def make(nbytes):
        return DeviceBuf(torch, nbytes)

    return make


def make_vine_scene(pt, depth, nearest_is_deepest):
    """One mesh of depth+1 parallel triangles and a hand-made BLAS shaped like a vine: every interior
    node has one leaf child and one interior child, and all interior boxes coincide, so a ray along
    the stem finds the interior child nearer at every level and pushes one leaf per level."""
    n_tri = depth + 1
    xs = (400.0 - np.arange(n_tri)) if nearest_is_deepest else (100.0 + np.arange(n_tri))
    pos = np.zeros((n_tri, 3, 3), np.float32)
    pos[:, :, 0] = xs[:, None]
    pos[:, 1, 1] = 1.0
    pos[:, 2, 2] = 1.0
    pos = pos.reshape(-1, 3)
    nrm = np.tile(np.array([[-1, 0, 0]], np.float32), (len(pos), 1))
    hs = pt.HostScene()
    hs.set_integrator(2, 1)
    hs.set_camera([-10, 0.25, 0.25], [0, 0.25, 0.25], [0, 1, 0], 45.0, 8, 8)
    tex = hs.add_constant_texture3([1, 1, 1])
    hs.add_material(pt.abi.PT_MAT_LAMBERTIAN, tex)
    first = hs.add_mesh(pos, nrm, np.zeros((len(pos), 2), np.float32), np.arange(len(pos), dtype=np.uint32))
    hs.add_instance(first, len(pos), len(pos), 0, np.eye(4, dtype=np.float32).reshape(-1), hs.new_blas_index())
    hs.finalize()
    nodes = np.zeros(2 * depth + 1, pt.abi.BVH_NODE)

    def tri_box(k):
        return [xs[k] - 0.01, 0.0, 0.0], [xs[k] + 0.01, 1.0, 1.0]
    for k in range(depth):  # stem node k: index 0 for the root, else 2k; children 2k+1 (leaf k) and 2k+2
        i = 0 if k == 0 else 2 * k
        nodes[i]["min"], nodes[i]["max"] = [-1, -1, -1], [500, 2, 2]
        nodes[i]["first_primitive"], nodes[i]["primitive_count"] = 2 * k + 1, 0
        leaf = 2 * k + 1
        nodes[leaf]["min"], nodes[leaf]["max"] = tri_box(k)
        nodes[leaf]["first_primitive"], nodes[leaf]["primitive_count"] = k, 1
    last = 2 * depth  # the deepest stem node's second child is the last leaf
    nodes[last]["min"], nodes[last]["max"] = tri_box(depth)
    nodes[last]["first_primitive"], nodes[last]["primitive_count"] = depth, 1
    blases = np.zeros(1, pt.abi.BLAS)
    blases[0] = (0, len(nodes), 0, n_tri)
    keep = {"nodes": nodes, "blases": blases, "tris": np.arange(n_tri, dtype=np.uint32), "hs": hs}
    desc = pt.abi.SceneDesc.from_buffer_copy(hs.desc())
    desc.blas_nodes, desc.blas_node_count = nodes.ctypes.data, len(nodes)
    desc.blases, desc.blas_count = blases.ctypes.data, 1
    desc.tri_indices, desc.tri_index_count = pt.abi.uptr(keep["tris"]), n_tri
    return desc, keep
