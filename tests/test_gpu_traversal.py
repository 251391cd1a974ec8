"""GPU traversal kernels against the CPU oracle, through the C ABI (pt_trace_closest,
pt_trace_any, pt_trace_closest_counted, pt_resolve_hits, ray generation).

Bar (north star): closest-hit primitive / instance ids equal except at exact ties
(<= 1e-5 of the rays), hit distance within 1e-5 relative.  The kernels keep the oracle's
operation order with fp contraction off, so the tests additionally require bit-equal
distances and barycentrics and equal node / triangle visit counts, which is what actually
holds; the looser bar is the fallback documented for other compilers."""
import numpy as np
import pytest

from conftest import make_vine_scene, rays_from_arrays

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(3.4028235e38)


def _upload_scene(gpu_ctx, hs):
    hs_desc = hs.desc()
    gpu_ctx.upload(hs_desc)
    return hs_desc


def _compare_closest(pt, hits, ref, n):
    ids_differ = (hits["prim"] != ref["prim"]) | (hits["inst"] != ref["inst"])
    assert ids_differ.sum() <= max(0, int(n * 1e-5)), f"{ids_differ.sum()} of {n} closest-hit ids differ"
    both = (hits["inst"] >= 0) & (ref["inst"] >= 0) & ~ids_differ
    rel = np.abs(hits["t"][both] - ref["t"][both]) / np.maximum(np.abs(ref["t"][both]), 1e-30)
    assert rel.size == 0 or rel.max() <= 1e-5
    # what this build actually achieves: bit equality
    same = ~ids_differ
    for f in ("t", "u", "v"):
        assert np.array_equal(hits[f][same].view(np.uint32), ref[f][same].view(np.uint32)), f
    assert np.array_equal(hits["material_id"][same], ref["material_id"][same])


@pytest.mark.parametrize("name,n", [("tiny", 30000), ("teapot", 60000), ("cornell", 40000), ("pbr_metallic", 40000),
                                    ("instancing_stress", 60000), ("helmet", 100000)])
def test_closest_hit_incoherent_rays(pt, ora, gpu_ctx, devbuf, scene_factory, tiny_scene, name, n):
    import torch

    hs = tiny_scene if name == "tiny" else scene_factory(name, 128, 96, 5)
    desc = _upload_scene(gpu_ctx, hs)
    rays_d, hits_d, ctr_d = devbuf(n * 32), devbuf(n * 32), devbuf(n * 16)
    gpu_ctx.generate_incoherent_rays(rays_d.ptr, n, 5)
    gpu_ctx.trace_closest_counted(rays_d.ptr, n, hits_d.ptr, ctr_d.ptr)
    gpu_ctx.synchronize()
    rays = rays_d.download(pt.abi.RAY, n)
    hits = hits_d.download(pt.abi.HIT, n)
    ctr = ctr_d.download(np.uint32, n * 4).reshape(n, 4).astype(np.uint64).sum(axis=0)
    # device-generated rays follow the same recipe as the oracle's generator, the azimuth's sin / cos included
    # (the device's sincos_azimuth against the oracle's restatement of it, bit for bit)
    ref_rays = ora.generate_incoherent_rays(desc, n, 5)
    assert rays.tobytes() == ref_rays.tobytes()
    ref, ref_ctr = ora.trace_closest(desc, rays, nthreads=8, counters=True)
    assert (ref["inst"] >= 0).sum() > n // 50
    _compare_closest(pt, hits, ref, n)
    # identical traversal work: interior nodes, TLAS leaves, triangles, BLAS leaves
    assert list(ctr) == list(ref_ctr[:4])
    # the plain (non-counting) kernel returns the same records
    hits2_d = devbuf(n * 32)
    gpu_ctx.trace_closest(rays_d.ptr, n, hits2_d.ptr)
    gpu_ctx.synchronize()
    assert np.array_equal(hits2_d.download(np.uint8, n * 32), hits_d.download(np.uint8, n * 32))
    st = gpu_ctx.stats()
    assert st.stack_drops == 0
    del torch


@pytest.mark.parametrize("name", ["teapot", "helmet"])
def test_closest_hit_captured_path_rays(pt, ora, gpu_ctx, devbuf, scene_factory, name):
    """Camera rays and the incoherent extension rays of bounces 1..3, captured on the GPU."""
    w, h = 160, 120
    hs = scene_factory(name, w, h, 5)
    desc = _upload_scene(gpu_ctx, hs)
    gpu_ctx.integrator_init(w, h)
    rays_d, hits_d = devbuf(w * h * 32), devbuf(w * h * 32)
    counts = []
    for bounce in range(4):
        n = gpu_ctx.capture_bounce_rays(w, h, 3, bounce, rays_d.ptr, w * h)
        counts.append(n)
        assert n > 0
        gpu_ctx.trace_closest(rays_d.ptr, n, hits_d.ptr)
        gpu_ctx.synchronize()
        rays = rays_d.download(pt.abi.RAY, n)
        hits = hits_d.download(pt.abi.HIT, n)
        ref = ora.trace_closest(desc, rays, nthreads=8)
        _compare_closest(pt, hits, ref, n)
    assert counts[0] == w * h and counts[1] < counts[0]


def test_any_hit_matches_oracle(pt, ora, gpu_ctx, devbuf, scene_factory, tiny_scene):
    for hs, n in ((tiny_scene, 40000), (scene_factory("teapot", 128, 96, 5), 60000)):
        desc = _upload_scene(gpu_ctx, hs)
        rays = ora.generate_incoherent_rays(desc, n, 77)
        closest = ora.trace_closest(desc, rays, nthreads=8)
        rs = np.random.RandomState(1)
        # t_max around the closest hit (both sides), short, and unbounded
        scale = rs.choice(np.array([0.5, 0.999, 1.001, 2.0], np.float32), size=n)
        rays["tmax"] = np.where(closest["inst"] >= 0, closest["t"] * scale, rs.rand(n).astype(np.float32) * 20)
        rays["tmax"][::7] = FLT_MAX
        occ = gpu_ctx.trace_any_host(rays)
        ref = ora.trace_any(desc, rays, nthreads=8)
        assert np.array_equal(occ, ref)
        assert 0.05 < occ.mean() < 0.95


@pytest.mark.parametrize("name", ["teapot", "helmet", "cornell", "pbr_ior", "instancing_stress"])
def test_any_hit_on_captured_shadow_rays(pt, ora, gpu_ctx, devbuf, scene_factory, name):
    """The next-event-estimation rays the shading really emits (t_max = distance to the light sample
    - 1e-4): the occlusion flags equal the oracle's, and both outcomes occur."""
    w, h = 96, 96
    hs = scene_factory(name, w, h, 6)
    desc = _upload_scene(gpu_ctx, hs)
    gpu_ctx.integrator_init(w, h)
    buf = devbuf(w * h * 32)
    total = 0
    for bounce in (0, 1):
        n = gpu_ctx.capture_shadow_rays(w, h, 0, bounce, buf.ptr, w * h)
        assert 0 < n <= w * h
        rays = buf.download(pt.abi.RAY, n)
        assert np.isfinite(rays["tmax"]).all() and (rays["tmax"] < 1e6).all()
        occ = gpu_ctx.trace_any_host(rays)
        ref = ora.trace_any(desc, rays, nthreads=8)
        assert np.array_equal(occ, ref), (name, bounce, int((occ != ref).sum()), n)
        assert 0.002 < occ.mean() < 0.998, (name, bounce, float(occ.mean()))
        # trace mode 1 skips boxes beyond t_max * (1 + 2^-12): it could only lose occluders whose triangle
        # test lands within rounding of t_max, and the margin leaves those to the triangle test
        gpu_ctx.set_trace_mode(1)
        fast = gpu_ctx.trace_any_host(rays)
        gpu_ctx.set_trace_mode(0)
        assert (fast <= ref).all() and (fast != ref).sum() <= max(1, int(n * 1e-5))
        total += n
    # the same rays are what the render traced: one iteration's shadow-ray count covers them
    gpu_ctx.reset_stats()
    gpu_ctx.render(w, h, 0)
    assert gpu_ctx.stats().shadow_rays >= total


def test_any_hit_with_negative_recomputed_distance_counts_as_free(pt, ora, gpu_ctx, scene_factory):
    """The reference's any-hit search does not stop at the first accepted triangle: it finishes the instance,
    and reports a hit only if the world-space distance recomputed from the last accepted triangle is not
    negative (kernel/intersection.h:101-109, :139-149).  The golden ray touches a triangle at its origin
    (t = 6.5e-6) and is therefore free, in every trace mode, like in the oracle; rays around it keep their
    flags."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "any_hit_negative_distance.npz"))
    w, h = int(g["width"]), int(g["height"])
    hs = scene_factory(str(g["scene"]), w, h, 8)
    desc = _upload_scene(gpu_ctx, hs)
    golden = g["rays"].astype(pt.abi.RAY)
    rs = np.random.RandomState(3)
    rays = np.repeat(golden, 4096)
    jitter = rs.normal(0.0, 2e-6, size=(len(rays), 3)).astype(np.float32)
    jitter[0] = 0
    for k, axis in enumerate(("x", "y", "z")):
        rays["origin"][:, k] += jitter[:, k]
    ref = ora.trace_any(desc, rays, nthreads=8)
    assert ref[0] == 0 and 0 < ref.mean() < 1  # the neighbourhood holds both outcomes
    try:
        for mode in (0, 1):
            gpu_ctx.set_trace_mode(mode)
            assert np.array_equal(gpu_ctx.trace_any_host(rays), ref), mode
        for mode in (2, 3):  # another visiting order: the last accepted triangle may be another one
            gpu_ctx.set_trace_mode(mode)
            occ = gpu_ctx.trace_any_host(rays)
            assert occ[0] == 0 and (occ != ref).mean() < 0.02, mode
    finally:
        gpu_ctx.set_trace_mode(0)
    assert gpu_ctx.trace_closest_host(rays[:1])["inst"][0] == -1  # the closest-hit query gives up the ray in the same way


@pytest.mark.parametrize("cursors", [None, "1", "3", "64"])
def test_ragged_batch_sizes_and_work_feed_settings(pt, ora, scene_factory, cursors, monkeypatch):
    """Every ray of a batch is traced exactly once whatever the batch size (below, at and just above
    multiples of the 64-ray fetch) and however many work-fetch cursors hand the rays out: untouched
    entries of the output would keep the 0xFF fill, and double work would show as nothing, so the
    check is equality with the oracle on every entry."""
    import torch
    if cursors is None:
        monkeypatch.delenv("PTAMD_QUERY_CURSORS", raising=False)
    else:
        monkeypatch.setenv("PTAMD_QUERY_CURSORS", cursors)
    ctx = pt.Context(0)
    try:
        hs = scene_factory("teapot", 64, 64, 4)
        desc = hs.desc()
        ctx.upload(desc)
        all_rays = ora.generate_incoherent_rays(desc, 70000, 123)
        ref_all = ora.trace_closest(desc, all_rays, nthreads=8)
        dev = torch.device("cuda", 0)
        for n in (1, 2, 63, 64, 65, 127, 128, 129, 511, 513, 4097, 70000):
            rays_t = torch.from_numpy(all_rays[:n].view(np.uint8).reshape(-1).copy()).to(dev)
            hits_t = torch.full((n * 32,), 0xFF, dtype=torch.uint8, device=dev)
            occ_t = torch.full((n,), 0xFF, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()  # the context launches on a stream of its own
            ctx.trace_closest(rays_t.data_ptr(), n, hits_t.data_ptr())
            ctx.trace_any(rays_t.data_ptr(), n, occ_t.data_ptr())
            ctx.synchronize()
            hits = hits_t.cpu().numpy().view(pt.abi.HIT)
            _compare_closest(pt, hits, ref_all[:n], n)
            assert np.array_equal(occ_t.cpu().numpy(), (ref_all[:n]["inst"] >= 0).astype(np.uint8)), n
    finally:
        ctx.close()


def test_closest_hit_honours_tmax_and_degenerate_rays(pt, ora, gpu_ctx, scene_factory):
    hs = scene_factory("cornell", 64, 64, 4)
    desc = _upload_scene(gpu_ctx, hs)
    n = 20000
    rays = ora.generate_incoherent_rays(desc, n, 9)
    closest = ora.trace_closest(desc, rays, nthreads=8)
    rays["tmax"] = np.where(closest["inst"] >= 0, closest["t"] * np.float32(0.9), np.float32(1.0))
    ref = ora.trace_closest(desc, rays, nthreads=8)
    hits = gpu_ctx.trace_closest_host(rays)
    _compare_closest(pt, hits, ref, n)
    # axis-parallel directions (inv_d = inf, 0*inf NaNs in the slab test), zero direction, NaN origin
    o = np.array([[0, 5, 0], [0, 5, 0], [0, 5, 0], [0, 5, 0], [np.nan, 5, 0], [0, 5, 20]], np.float32)
    d = np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1], [0, 0, 0], [0, 0, -1], [0, 0, 1]], np.float32)
    rays = rays_from_arrays(pt, o, d)
    hits = gpu_ctx.trace_closest_host(rays)
    ref = ora.trace_closest(desc, rays, nthreads=1)
    assert np.array_equal(hits["inst"], ref["inst"]) and np.array_equal(hits["prim"], ref["prim"])
    assert np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert ref["inst"][0] >= 0 and ref["inst"][3] < 0 and ref["inst"][4] < 0 and ref["inst"][5] < 0


def test_exact_ties_resolve_like_the_oracle(pt, ora, gpu_ctx):
    """Coplanar duplicated geometry: two instances of one quad at the same place and a mesh whose
    triangles are all identical.  Strict '<' keeps the first candidate found; the GPU follows the
    oracle's traversal order, so even the ids agree."""
    hs = pt.HostScene()
    hs.set_integrator(2, 1)
    hs.set_camera([0, 0, 5], [0, 0, 0], [0, 1, 0], 45.0, 8, 8)
    tex = hs.add_constant_texture3([1, 1, 1])
    m_light = hs.add_material(pt.abi.PT_MAT_EMISSIVE, tex, np.float32(1.0))
    quad_p = np.array([[-1, -1, 0], [1, -1, 0], [-1, 1, 0], [1, 1, 0]], np.float32)
    quad_n = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
    quad_t = np.zeros((4, 2), np.float32)
    first = hs.add_mesh(quad_p, quad_n, quad_t, np.array([0, 1, 2, 1, 3, 2], np.uint32))
    dup = hs.add_mesh(quad_p[:3], quad_n[:3], quad_t[:3], np.array([0, 1, 2] * 9, np.uint32))  # 9 identical triangles, one big leaf
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    b0, b1 = hs.new_blas_index(), hs.new_blas_index()
    hs.add_instance(first, 6, 4, m_light, ident, b0)
    hs.add_instance(first, 6, 4, 0, ident, b0)
    hs.add_instance(dup, 27, 3, 0, pt.transform_matrix([0, 0, -1], [0, 0, 0], [1, 1, 1]), b1)
    hs.finalize()
    desc = _upload_scene(gpu_ctx, hs)
    rs = np.random.RandomState(4)
    n = 5000
    o = np.concatenate([rs.uniform(-1.2, 1.2, size=(n, 2)), np.full((n, 1), 3.0)], axis=1).astype(np.float32)
    d = np.tile(np.array([[0, 0, -1]], np.float32), (n, 1))
    rays = rays_from_arrays(pt, o, d)
    hits = gpu_ctx.trace_closest_host(rays)
    ref = ora.trace_closest(desc, rays, nthreads=1)
    assert np.array_equal(hits["inst"], ref["inst"]) and np.array_equal(hits["prim"], ref["prim"])
    assert np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert (ref["inst"] >= 0).mean() > 0.5


@pytest.mark.parametrize("depth,nearest_is_deepest", [(40, True), (63, True), (64, True), (65, True), (100, True), (100, False)])
def test_stack_overflow_drops_pushes_like_the_reference(pt, ora, gpu_ctx, depth, nearest_is_deepest):
    """The reference's traversal stack holds 64 entries and silently ignores pushes beyond that
    (utils/stack.h:8-11).  A vine-shaped BLAS drives the stack past 64: from there on the nearer
    child's push is lost, the walk turns back, and the ray reports a farther triangle than the true
    closest one.  The kernel keeps deeper entries than its LDS part in a spill area and reproduces
    which pushes are lost, so it returns the same (wrong) triangle as the oracle."""
    desc, keep = make_vine_scene(pt, depth, nearest_is_deepest)
    gpu_ctx.upload(desc)
    gpu_ctx.reset_stats()
    n = 256
    rs = np.random.RandomState(depth)
    o = np.concatenate([np.full((n, 1), -10.0), rs.uniform(0.05, 0.45, size=(n, 2))], axis=1).astype(np.float32)
    d = np.tile(np.array([[1, 0, 0]], np.float32), (n, 1))
    rays = rays_from_arrays(pt, o, d)
    hits = gpu_ctx.trace_closest_host(rays)
    ref = ora.trace_closest(desc, rays, nthreads=1)
    assert np.array_equal(hits["prim"], ref["prim"]) and np.array_equal(hits["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert (ref["prim"] >= 0).all()
    occluded = gpu_ctx.trace_any_host(rays)
    assert np.array_equal(occluded, ora.trace_any(desc, rays))
    st = gpu_ctx.stats()
    true_nearest = depth if nearest_is_deepest else 0
    if depth >= 64:
        assert st.stack_drops > 0 and st.stack_spills > 0
        if nearest_is_deepest:
            assert (ref["prim"] != true_nearest).all()  # the closest triangle sits behind a lost push
    else:
        assert st.stack_drops == 0 and (ref["prim"] == true_nearest).all()


def test_resolve_hits_matches_oracle_intersection(pt, ora, gpu_ctx, scene_factory, tiny_scene):
    for hs in (tiny_scene, scene_factory("helmet", 128, 96, 5)):
        desc = _upload_scene(gpu_ctx, hs)
        n = 30000
        rays = ora.generate_incoherent_rays(desc, n, 21)
        hits = gpu_ctx.trace_closest_host(rays)
        full = gpu_ctx.resolve_hits_host(rays, hits)
        ref_hits, ref_full = ora.trace_closest(desc, rays, nthreads=8, full=True)
        ok = ref_hits["inst"] >= 0
        assert np.array_equal(full["geometry_id"], ref_full["geometry_id"])
        assert np.array_equal(full["material_id"], ref_full["material_id"])
        for f in ("time", "pos", "normal", "uv"):
            assert np.array_equal(full[f][ok].view(np.uint32), ref_full[f][ok].view(np.uint32)), f
        assert (full["time"][~ok] == -1).all()


def test_queries_reject_bad_arguments(pt, gpu_ctx, devbuf, tiny_scene):
    fresh = pt.Context(0)
    buf = devbuf(64 * 32)
    with pytest.raises(pt.PtError, match="before pt_scene_upload"):
        fresh.trace_closest(buf.ptr, 64, buf.ptr)
    with pytest.raises(pt.PtError, match="before pt_scene_upload"):
        fresh.render(8, 8, 0)
    fresh.close()
    _upload_scene(gpu_ctx, tiny_scene)
    with pytest.raises(pt.PtError, match="16-byte aligned"):
        gpu_ctx.trace_closest(buf.ptr + 4, 8, buf.ptr)
    with pytest.raises(pt.PtError, match="null buffer"):
        gpu_ctx.trace_closest(0, 8, buf.ptr)
    gpu_ctx.trace_closest(buf.ptr, 0, buf.ptr)  # empty batch is a no-op
    assert len(gpu_ctx.trace_closest_host(np.zeros(0, pt.abi.RAY))) == 0
    fresh = pt.Context(0)
    fresh.upload(tiny_scene.desc())
    with pytest.raises(pt.PtError, match="before pt_integrator_init"):
        fresh.render(64, 48, 0)
    fresh.integrator_init(64, 48)
    with pytest.raises(pt.PtError, match="differs from the integrator"):
        fresh.render(32, 32, 0)
    fresh.close()


def test_upload_validation(pt, gpu_ctx, tiny_scene):
    import copy
    import ctypes as C

    d = tiny_scene.desc()
    bad = pt.abi.SceneDesc.from_buffer_copy(d)
    bad.max_depth = 0
    with pytest.raises(pt.PtError, match="max_depth"):
        gpu_ctx.upload(bad)
    bad = pt.abi.SceneDesc.from_buffer_copy(d)
    bad.geometry_count = 0
    with pytest.raises(pt.PtError, match="no geometry"):
        gpu_ctx.upload(bad)
    arr = tiny_scene.arrays()
    idx = arr["indices"].copy()
    idx[5] = 10 ** 6
    bad = pt.abi.SceneDesc.from_buffer_copy(d)
    bad.indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
    with pytest.raises(pt.PtError, match="out of range"):
        gpu_ctx.upload(bad)
    geo = arr["geometries"].copy()
    geo["material_id"][0] = 99
    bad = pt.abi.SceneDesc.from_buffer_copy(d)
    bad.geometries = geo.ctypes.data
    with pytest.raises(pt.PtError, match="bad material"):
        gpu_ctx.upload(bad)
    # a BLAS whose interior nodes share children is a graph, not a tree: refused (the 4-wide collapse would
    # unfold it into exponentially many nodes)
    nodes = arr["blas_nodes"].copy()
    blases = arr["blases"]
    b = next(i for i in range(len(blases)) if blases["node_count"][i] >= 7)
    first, count = int(blases["first_node"][b]), int(blases["node_count"][b])
    interior = [k for k in range(first, first + count) if nodes["primitive_count"][k] == 0]
    assert len(interior) >= 2
    k0, k1 = interior[0], interior[1]
    nodes["first_primitive"][k0] = nodes["first_primitive"][k1]  # k0 now names k1's children as its own
    bad = pt.abi.SceneDesc.from_buffer_copy(d)
    bad.blas_nodes = nodes.ctypes.data
    with pytest.raises(pt.PtError, match="more than one node|bad children"):
        gpu_ctx.upload(bad)
    gpu_ctx.upload(d)  # and the good description still uploads
    del copy
