"""Host C++ facade (product) against the CPU oracle: matrices, camera, alias tables, the SAH
BLAS / agglomerative TLAS build and the light tables must agree bit for bit; plus the C ABI
surface itself (every declared symbol is exported; errors are reported, not thrown)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


# ------------------------------------------------------------------------------ ABI surface
def test_library_exports_every_declared_symbol(pt):
    lib = pt.abi.load_library()
    declared = set()
    for header in ("ptamd.h", "ptamd_host.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(pth?_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    assert declared == set(pt.abi.PT_SYMBOLS) | set(pt.abi.PTH_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported by libptamd.so"


def test_struct_sizes_match_the_reference_layout(pt):
    assert pt.abi.BVH_NODE.itemsize == 32 and pt.abi.TLAS_NODE.itemsize == 32 and pt.abi.GEOMETRY.itemsize == 224
    assert C.sizeof(pt.abi.Material) == 16 and C.sizeof(pt.abi.Texture1) == 32 and C.sizeof(pt.abi.Texture3) == 44


def test_errors_are_status_codes_with_messages(pt):
    lib = pt.abi.load_library()
    assert lib.pt_version().startswith(b"ptamd")
    # no context: every entry point refuses politely
    assert lib.pt_scene_upload(None, None) == pt.abi.PT_ERR_INVALID
    assert lib.pt_render(None, 4, 4, 0, None) == pt.abi.PT_ERR_INVALID
    assert lib.pt_trace_closest(None, None, 0, None) == pt.abi.PT_ERR_INVALID
    assert lib.pt_ctx_create(0, None) == pt.abi.PT_ERR_INVALID
    assert b"out is null" in lib.pt_last_error(None)
    # host facade argument checking
    out = np.zeros(16, np.float32)
    assert lib.pth_transform_matrix(None, None, None, pt.abi.fptr(out)) == pt.abi.PT_ERR_INVALID
    assert b"null" in lib.pth_last_error()
    with pytest.raises(pt.PtError, match="CDF is 0"):
        pt.alias_build(np.zeros(3, np.float32))
    hs = pt.HostScene()
    with pytest.raises(pt.PtError, match="no objects"):
        hs.finalize()
    with pytest.raises(pt.PtError, match="index out of range"):
        hs.add_mesh(np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.zeros((3, 2), np.float32), np.array([0, 1, 3], np.uint32))
    with pytest.raises(pt.PtError, match="multiple of 3"):
        hs.add_mesh(np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.zeros((3, 2), np.float32), np.array([0, 1], np.uint32))
    with pytest.raises(pt.PtError, match="unknown material"):
        hs.add_instance(0, 3, 3, 99, np.eye(4, dtype=np.float32).reshape(-1), 0)


def test_context_creation_without_gpu_fails_cleanly(pt):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pt.PtError, match="no HIP device"):
        pt.Context(0)


# ----------------------------------------------------------------------------------- math
@pytest.mark.parametrize("t,r,s", [([0, 0, 0], [0, 0, 0], [1, 1, 1]), ([1, 2, 3], [30, 40, 50], [1, 2, 3]),
                                     ([0, 9.999, 0], [180, 0, 0], [1.5, 1, 1.5]), ([-2, 6, -1], [0, -60, 0], [0.1, 0.1, 0.1]),
                                     ([5, 5, 0], [0, 0, 90], [5, 1, 5])])
def test_transform_inverse_inverse_transpose_bit_exact(pt, ora, t, r, s):
    T = pt.transform_matrix(t, r, s)
    inv, invt = pt.matrix_derive(T)
    oT, oI, oIT = ora.transform_matrices(t, r, s)
    assert np.array_equal(_bits(T), _bits(oT))
    assert np.array_equal(_bits(inv), _bits(oI))
    assert np.array_equal(_bits(invt), _bits(oIT))
    # and they are what they claim to be
    M = T.reshape(4, 4).T.astype(np.float64)
    assert np.allclose(inv.reshape(4, 4).T @ M, np.eye(4), atol=1e-5)
    assert np.allclose(invt.reshape(4, 4).T, np.linalg.inv(M).T, rtol=1e-4, atol=1e-5)


def test_rotation_order_is_x_then_y_then_z(pt):
    T = pt.transform_matrix([0, 0, 0], [90, 0, 90], [1, 1, 1]).reshape(4, 4).T  # R = Rx * Ry * Rz
    v = T[:3, :3] @ np.array([1.0, 0, 0])
    # Rz first maps x -> y, then Rx maps y -> z
    assert np.allclose(v, [0, 0, 1], atol=1e-6)


def test_gltf_local_matrix(pt):
    s = np.float32(np.sqrt(0.5))
    M = pt.gltf_local_matrix([1, 2, 3], [s, 0, 0, s], [2, 2, 2], np.eye(4, dtype=np.float32).reshape(-1)).reshape(4, 4).T
    # 90 degrees about x: y -> z
    assert np.allclose(M[:3, :3] @ np.array([0, 1.0, 0]), [0, 0, 2], atol=1e-5)
    assert np.allclose(M[:3, 3], [1, 2, 3])


def test_camera_matches_oracle(pt, ora):
    for args in [([0, 0, 5], [0, 0, 0], [0, 1, 0], 45.0, 1920, 1080), ([0, 5, 10.5], [0, 5, 0], [0, 1, 0], 90.0, 800, 800),
                 ([3, 2, 1], [-1, 0.5, 2], [0.1, 1, 0], 33.0, 123, 77)]:
        a, b = pt.camera_make(*args), ora.camera_update(*args)
        for f in ("resolution", "position", "front", "up", "right", "pixel_length"):
            assert list(getattr(a, f)) == list(getattr(b, f)), f
        assert a.focal_length == b.focal_length == 1.0 and a.fovy == b.fovy


# --------------------------------------------------------------------------- alias tables
@pytest.mark.parametrize("seed,n", [(0, 1), (1, 2), (2, 7), (3, 64), (4, 1000), (5, 48768)])
def test_alias_tables_bit_exact(pt, ora, seed, n):
    rs = np.random.RandomState(seed)
    w = rs.rand(n).astype(np.float32)
    if n > 10:
        w[rs.randint(n, size=n // 5)] = w[0]  # exact ties
        w[rs.randint(n, size=n // 10)] = 0.0
    prob, alias = pt.alias_build(w)
    st, oprob, oalias = ora.alias_build(w)
    assert st == 0
    assert np.array_equal(_bits(prob), _bits(oprob)) and np.array_equal(alias, oalias)


# ------------------------------------------------------------------------------ BVH build
@pytest.mark.parametrize("name", ["teapot", "cornell", "pbr_metallic", "sphere", "instancing_stress", "helmet"])
def test_bvh_build_byte_exact(pt, ora, scene_factory, name):
    hs = scene_factory(name, 64, 64, 4)
    arr = hs.arrays()
    ref = ora.tlas_build(arr["positions"], arr["indices"], arr["geometries"])
    for key in ("tlas_nodes", "blases", "blas_nodes", "tri_indices"):
        assert len(arr[key]) == len(ref[key]), key
        assert np.array_equal(_bits(arr[key]), _bits(ref[key])), f"{name}: {key} differs from the oracle build"
    # the stand-alone builder entry point gives the same arrays
    again = pt.bvh_build(arr["positions"], arr["indices"], arr["geometries"])
    for key in ("tlas_nodes", "blases", "blas_nodes", "tri_indices"):
        assert np.array_equal(_bits(again[key]), _bits(ref[key]))


def test_bvh_edge_cases(pt, ora):
    """One triangle; two coincident triangles (degenerate centroid extent); a long strip."""
    def build(pos, idx):
        geo = np.zeros(1, pt.abi.GEOMETRY)
        geo["index_count"] = len(idx)
        geo["vertex_count"] = len(pos)
        ident = np.eye(4, dtype=np.float32).reshape(-1)
        geo["transform"] = geo["inv_transform"] = geo["inv_transpose"] = ident
        a = pt.bvh_build(pos, idx, geo)
        b = ora.tlas_build(pos, idx, geo)
        for k in a:
            assert np.array_equal(_bits(a[k]), _bits(b[k])), k
        return a

    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    a = build(tri, np.array([0, 1, 2], np.uint32))
    assert len(a["blas_nodes"]) == 1 and a["blas_nodes"][0]["primitive_count"] == 1 and len(a["tlas_nodes"]) == 2
    a = build(tri, np.array([0, 1, 2] * 5, np.uint32))  # identical centroids: no split plane exists
    assert len(a["blas_nodes"]) == 1 and a["blas_nodes"][0]["primitive_count"] == 5
    n = 300
    x = np.arange(n + 1, dtype=np.float32)
    pos = np.concatenate([np.stack([x, 0 * x, 0 * x], 1), np.stack([x, 0 * x + 1, 0 * x], 1)]).astype(np.float32)
    idx = np.array([[i, i + 1, n + 1 + i] for i in range(n)], np.uint32).reshape(-1)
    a = build(pos, idx)
    assert len(a["blas_nodes"]) > n // 2
    with pytest.raises(pt.PtError, match="at least one"):
        pt.bvh_build(pos, idx, np.zeros(0, pt.abi.GEOMETRY))


# --------------------------------------------------------------------------- light tables
@pytest.mark.parametrize("name", ["teapot", "cornell", "helmet"])
def test_light_data_bit_exact(pt, ora, scene_factory, name):
    hs = scene_factory(name, 64, 64, 4)
    arr = hs.arrays()
    geo_in = arr["geometries"].copy()
    geo_in["total_area"] = 0
    geo_in["area_light_id"] = -1
    geo_in["triangle_sampler_index"] = -1
    geo, lights, lp, la, tris, total = ora.light_data(arr["positions"], arr["indices"], geo_in, arr["materials"])
    assert np.array_equal(_bits(geo), _bits(arr["geometries"]))
    assert np.array_equal(_bits(lights), _bits(arr["lights"]))
    assert np.float32(total) == np.float32(arr["total_light_power"])
    d = hs.desc()
    assert d.light_count == len(lights) and d.triangle_sampler_count == len(tris)
    got_lp = np.ctypeslib.as_array(d.light_sampler.contents.probability_table, shape=(d.light_count,))
    got_la = np.ctypeslib.as_array(d.light_sampler.contents.alias_table, shape=(d.light_count,))
    assert np.array_equal(_bits(got_lp.copy()), _bits(lp)) and np.array_equal(got_la, la)
    for i, (p, a) in enumerate(tris):
        t = d.triangle_samplers[i]
        assert t.size == len(p)
        assert np.array_equal(_bits(np.ctypeslib.as_array(t.probability_table, shape=(t.size,)).copy()), _bits(p))
        assert np.array_equal(np.ctypeslib.as_array(t.alias_table, shape=(t.size,)), a)


# ------------------------------------------------------------------------ scene ingestion
def test_material_and_instance_ids_follow_the_loader_order(pt, scene_factory):
    hs = scene_factory("helmet", 64, 64, 4)
    arr = hs.arrays()
    mats = arr["materials"]
    # material 0 = default Lambertian; named materials in lexicographic order:
    # back_light, diffuse_blue, diffuse_green, diffuse_red, diffuse_white, fill_light, glass, key_light, metal, mirror
    E, L, D, M, MR = pt.abi.PT_MAT_EMISSIVE, pt.abi.PT_MAT_LAMBERTIAN, pt.abi.PT_MAT_DIELECTRIC, pt.abi.PT_MAT_MIRROR, pt.abi.PT_MAT_METALLIC_ROUGHNESS
    assert list(mats["type"][:11]) == [L, E, L, L, L, L, E, D, E, MR, M]
    assert np.float32(40.0).view(np.uint32) == mats["w1"][1]  # back_light emittance
    # three sphere loads add one untextured glTF material each, then the helmet's textured one ...
    assert list(mats["type"][11:]) == [MR] * 6
    geos = arr["geometries"]
    assert list(geos["material_id"]) == [8, 6, 1, 14, 3, 4]  # key, fill, back lights; helmet's own; green; red
    assert list(geos["blas_index"]) == [0, 1, 2, 3, 4, 5]
    assert list(geos["index_count"]) == [146304] * 3 + [46356] * 3
    # the helmet node carries a 90-degree rotation about x and the object a translation
    T = geos["transform"][4].reshape(4, 4).T
    assert np.allclose(T[:3, 3], [2, 0, 0]) and np.allclose(T[:3, :3] @ np.array([0, 1.0, 0]), [0, 0, 1], atol=1e-6)
    # face normals requested for the spheres: unit length and constant per triangle
    n = arr["normals"][arr["indices"][:300]].reshape(100, 3, 3)
    assert np.allclose(np.linalg.norm(n, axis=2), 1.0, atol=1e-5) and np.array_equal(n[:, 0], n[:, 1])


def test_cooked_scene_roundtrip(pt, tmp_path):
    spec = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "teapot.npz"))
    p = str(tmp_path / "again.npz")
    spec.save(p)
    again = pt.SceneSpec.load(p)
    a, b = spec.build(32, 32, 4).arrays(), again.build(32, 32, 4).arrays()
    for k in a:
        assert np.array_equal(_bits(np.asarray(a[k])), _bits(np.asarray(b[k]))), k


def test_legacy_adapter_primitives(pt):
    p, n, t, i = pt.unit_cube()
    assert len(i) == 36 and np.abs(p).max() == 0.5 and np.allclose(np.linalg.norm(n, axis=1), 1)
    e1, e2 = p[i[1::3]] - p[i[0::3]], p[i[2::3]] - p[i[0::3]]
    assert (np.einsum("ij,ij->i", np.cross(e1, e2), n[i[0::3]]) > 0).all()  # outward winding
    p, n, t, i = pt.uv_sphere()
    assert np.allclose(np.linalg.norm(p, axis=1), 0.5, atol=1e-6) and np.allclose(np.linalg.norm(n, axis=1), 1, atol=1e-6)
    assert len(i) // 3 == 2 * 64 * 31


def test_png_and_hdr_writers(pt, tmp_path):
    from PIL import Image

    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, size=(37, 53, 4)).astype(np.uint8)
    path = str(tmp_path / "out.png")
    pt.save_png(path, img)
    back = np.asarray(Image.open(path).convert("RGB"))
    assert np.array_equal(back, img[:, :, :3])
    film = (rs.rand(9, 11, 3) * 8).astype(np.float32)
    hdr = str(tmp_path / "out.hdr")
    pt.save_hdr(hdr, film, 2)
    raw = open(hdr, "rb").read()
    assert raw.startswith(b"#?RADIANCE") and b"-Y 9 +X 11" in raw
    body = np.frombuffer(raw[-9 * 11 * 4:], np.uint8).reshape(9, 11, 4)
    dec = body[..., :3].astype(np.float64) * (2.0 ** (body[..., 3:4].astype(np.float64) - 136.0))
    ref = film.astype(np.float64) / 2
    assert (np.abs(dec - ref) <= ref.max(axis=2, keepdims=True) / 128).all()  # 8-bit mantissa, shared exponent


def test_checkpoint_reader_rejects_foreign_files(pt, tmp_path):
    p = tmp_path / "x.film"
    with pytest.raises(pt.PtError, match="no checkpoint"):
        pt.checkpoint_read(str(p))
    p.write_bytes(b"PTFILM1\0" + b"\0" * 4)
    with pytest.raises(pt.PtError, match="not a film checkpoint"):
        pt.checkpoint_read(str(p))
    import struct
    p.write_bytes(b"PTFILM1\0" + struct.pack("<iiII", 2, 1, 7, 0) + np.arange(6, dtype=np.float32).tobytes())
    film, it = pt.checkpoint_read(str(p))
    assert it == 7 and film.shape == (1, 2, 3) and np.array_equal(film.reshape(-1), np.arange(6, dtype=np.float32))
    p.write_bytes(b"PTFILM1\0" + struct.pack("<iiII", 2, 1, 7, 0) + np.arange(5, dtype=np.float32).tobytes())  # truncated
    with pytest.raises(pt.PtError, match="not a film checkpoint"):
        pt.checkpoint_read(str(p))
    # a header that announces a 51 GB film over a 24-byte body is refused before anything is allocated
    p.write_bytes(b"PTFILM1\0" + struct.pack("<iiII", 65536, 65536, 7, 0) + np.arange(6, dtype=np.float32).tobytes())
    with pytest.raises(pt.PtError, match="not a film checkpoint"):
        pt.checkpoint_read(str(p))
