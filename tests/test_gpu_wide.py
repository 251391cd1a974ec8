"""Trace modes 2 and 3 (4-wide BVH with the reference's own fp32 boxes, top level rebuilt, searches
inside an instance bounded by the best hit so far) against the CPU oracle, through the C ABI.

Bar for these modes: closest-hit primitive / instance ids equal to the oracle's except at ties,
<= 1e-5 of the rays; hit distance within 1e-5 relative (the winner's record comes from the same
triangle test on the same triangle record, so wherever the ids agree the distance and the
barycentrics are in fact bit-equal, which the tests also require).  Any-hit flags: mode 2 tests every
box like the reference, flags equal except <= 1e-5 of the rays; mode 3 also skips boxes that start
beyond t_max * (1 + 2^-12), which could only turn "occluded" into "free" (an occluder within rounding of
t_max, see pt_set_trace_mode): the same bar, <= 1e-5 of the rays.  Rendered film of one
sample: bit-identical to the oracle's except on pixels whose path met such a tie, at most 1e-3 of them
(measured: 0 to 3 of 129 600 pixels after four samples, tools/film_exactness.py)."""
import numpy as np
import pytest

from conftest import make_vine_scene, rays_from_arrays

pytestmark = pytest.mark.gpu


@pytest.fixture()
def wide_ctx(gpu_ctx):
    """The shared context in trace mode 2; always handed back in mode 0."""
    gpu_ctx.set_trace_mode(2)
    yield gpu_ctx
    gpu_ctx.set_trace_mode(0)


def _check_closest(hits, ref, n, tie_budget=1e-5, tie_t_tol=1e-5):
    ids_differ = (hits["prim"] != ref["prim"]) | (hits["inst"] != ref["inst"])
    assert ids_differ.sum() <= int(n * tie_budget), f"{ids_differ.sum()} of {n} closest-hit ids differ"
    assert not (ids_differ & ((hits["inst"] >= 0) != (ref["inst"] >= 0))).any()  # never a hit against a miss
    same = ~ids_differ
    for f in ("t", "u", "v"):
        assert np.array_equal(hits[f][same].view(np.uint32), ref[f][same].view(np.uint32)), f
    assert np.array_equal(hits["material_id"][same], ref["material_id"][same])
    both = ids_differ & (hits["inst"] >= 0) & (ref["inst"] >= 0)  # ties: another triangle at the same distance
    if both.any():
        rel = np.abs(hits["t"][both] - ref["t"][both]) / np.maximum(np.abs(ref["t"][both]), 1e-30)
        assert rel.max() <= tie_t_tol
    return int(ids_differ.sum())


@pytest.mark.parametrize("name,n", [("tiny", 100000), ("teapot", 200000), ("cornell", 200000), ("pbr_metallic", 200000),
                                    ("pbr_ior", 200000), ("helmet", 300000), ("instancing_stress", 200000), ("party_armadillo", 300000)])
def test_wide_closest_and_any_hit_on_incoherent_rays(pt, ora, wide_ctx, devbuf, scene_factory, tiny_scene, name, n):
    hs = tiny_scene if name == "tiny" else scene_factory(name, 128, 96, 5)
    desc = hs.desc()
    wide_ctx.upload(desc)
    rays_d, hits_d, occ_d = devbuf(n * 32), devbuf(n * 32), devbuf(n)
    wide_ctx.generate_incoherent_rays(rays_d.ptr, n, 31)
    wide_ctx.reset_stats()
    wide_ctx.trace_closest(rays_d.ptr, n, hits_d.ptr)
    wide_ctx.trace_any(rays_d.ptr, n, occ_d.ptr)
    wide_ctx.synchronize()
    rays = rays_d.download(pt.abi.RAY, n)
    hits = hits_d.download(pt.abi.HIT, n)
    occ = occ_d.download(np.uint8, n)
    ref = ora.trace_closest(desc, rays, nthreads=8)
    assert (ref["inst"] >= 0).sum() > n // 50
    # the hand-made scene stands a cube exactly on the floor quad: rays through the shared face see two
    # instances at the same distance, and which one the reference keeps depends on its TLAS order.  For a ray
    # that grazes the shared plane the two distances the reference recomputes differ by 1e-5 of the distance
    # (their error grows with 1 / sin of the grazing angle), so that is how far apart the two answers can be.
    _check_closest(hits, ref, n, tie_budget=1e-3 if name == "tiny" else 1e-5, tie_t_tol=1e-4 if name == "tiny" else 1e-5)
    ref_occ = ora.trace_any(desc, rays, nthreads=8)
    assert (occ != ref_occ).sum() <= int(n * 1e-5)
    assert wide_ctx.stats().stack_drops == 0
    # mode 3: the same closest hits; any-hit flags may only lose occluders at the very end of the ray
    wide_ctx.set_trace_mode(3)
    wide_ctx.trace_any(rays_d.ptr, n, occ_d.ptr)
    wide_ctx.synchronize()
    occ3 = occ_d.download(np.uint8, n)
    assert not ((occ3 == 1) & (ref_occ == 0)).any()
    assert (occ3 != ref_occ).sum() <= int(n * 1e-5)


@pytest.mark.parametrize("name", ["teapot", "cornell", "helmet", "party_armadillo"])
def test_wide_on_the_scenes_own_bounce_and_shadow_rays(pt, ora, wide_ctx, devbuf, scene_factory, name):
    """Secondary rays start on surfaces, graze them and end on lights: where rounding decides, the
    wide traversal must decide like the reference (same boxes, same arithmetic)."""
    w, h = 320, 180
    hs = scene_factory(name, w, h, 8)
    desc = hs.desc()
    wide_ctx.upload(desc)
    wide_ctx.integrator_init(w, h)
    buf = devbuf(w * h * 32)
    total = differ = 0
    for bounce in (1, 2, 3):
        n = wide_ctx.capture_bounce_rays(w, h, 0, bounce, buf.ptr, w * h)
        if n == 0:
            continue
        rays = buf.download(pt.abi.RAY, n)
        hits = wide_ctx.trace_closest_host(rays)
        ref = ora.trace_closest(desc, rays, nthreads=8)
        ids = (hits["prim"] != ref["prim"]) | (hits["inst"] != ref["inst"])
        same = ~ids
        assert np.array_equal(hits["t"][same].view(np.uint32), ref["t"][same].view(np.uint32))
        total += n
        differ += int(ids.sum())
    assert total > 10000 and differ <= max(1, int(total * 1e-5)), (differ, total)
    sh_total = sh_differ = sh_differ3 = 0
    for bounce in (0, 1):
        n = wide_ctx.capture_shadow_rays(w, h, 0, bounce, buf.ptr, w * h)
        if n == 0:
            continue
        rays = buf.download(pt.abi.RAY, n)
        occ = wide_ctx.trace_any_host(rays)
        wide_ctx.set_trace_mode(3)
        occ3 = wide_ctx.trace_any_host(rays)
        wide_ctx.set_trace_mode(2)
        ref_occ = ora.trace_any(desc, rays, nthreads=8)
        sh_total += n
        sh_differ += int((occ != ref_occ).sum())
        sh_differ3 += int((occ3 != ref_occ).sum())
        assert not ((occ3 == 1) & (ref_occ == 0)).any()
    assert sh_total > 1000 and sh_differ <= max(1, int(sh_total * 1e-5)), (sh_differ, sh_total)
    assert sh_differ3 <= max(1, int(sh_total * 1e-5)), (sh_differ3, sh_total)


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("name,w,h,depth", [("tiny", 64, 48, 6), ("teapot", 96, 96, 6), ("cornell", 80, 80, 8), ("helmet", 128, 72, 8),
                                            ("party_armadillo", 96, 54, 8)])
def test_wide_one_sample_film_matches_oracle(pt, ora, gpu_ctx, scene_factory, tiny_scene, name, w, h, depth, mode):
    hs = tiny_scene if name == "tiny" else scene_factory(name, w, h, depth)
    desc = hs.desc()
    gpu_ctx.upload(desc)
    gpu_ctx.integrator_init(w, h)
    try:
        gpu_ctx.set_trace_mode(mode)
        gpu_ctx.reset_stats()
        gpu_ctx.render(w, h, 0)
        film = gpu_ctx.film_download(w, h)
        st = gpu_ctx.stats()
    finally:
        gpu_ctx.set_trace_mode(0)
    ref, _ = ora.render(desc, w, h, 0, nthreads=8)
    assert np.isfinite(film).all() and ref.mean() > 1e-3
    differ = (film.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    assert differ.mean() <= 1e-3, f"{name} mode {mode}: {int(differ.sum())} of {w * h} pixels differ from the oracle's film"
    assert st.paths == w * h and st.nan_pixels == 0 and st.stack_drops == 0


def test_wide_batched_render_is_reproducible_and_equals_mode_0_almost_everywhere(pt, gpu_ctx, scene_factory):
    w, h = 192, 108
    hs = scene_factory("helmet", w, h, 8)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    films = {}
    try:
        for mode in (0, 2, 2):
            gpu_ctx.set_trace_mode(mode)
            gpu_ctx.render_batch(w, h, 0, 6)
            gpu_ctx.render_batch(w, h, 6, 2)
            gpu_ctx.synchronize()
            film = gpu_ctx.film_download(w, h)
            if mode in films:
                assert np.array_equal(films[mode].view(np.uint32), film.view(np.uint32))
            films[mode] = film
    finally:
        gpu_ctx.set_trace_mode(0)
    differing = (films[0] != films[2]).any(axis=2).mean()
    assert differing <= 1e-3, differing


def _rotated_ellipsoids_scene(pt, count=60, seed=5):
    """Flat ellipsoids (a sphere mesh scaled to 3 x 0.15 x 0.6) at random rotations: the reference's instance box,
    the box of the rotated object-space box, is much larger than the ellipsoid's own world box."""
    rng = np.random.default_rng(seed)
    hs = pt.HostScene()
    hs.set_integrator(4, 4)
    hs.set_camera([0.0, 0.0, 12.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], 50.0, 64, 48)
    white = hs.add_constant_texture3([0.8, 0.8, 0.8])
    mat = hs.add_material(pt.abi.PT_MAT_LAMBERTIAN, white)
    cp, cn, ct, ci = pt.uv_sphere(10, 20)
    first = hs.add_mesh(cp, cn, ct, ci)
    blas = hs.new_blas_index()
    transforms = []
    for _ in range(count):
        t = pt.transform_matrix(list(rng.uniform(-3, 3, 3)), list(rng.uniform(0, 360, 3)), [3.0, 0.15, 0.6])
        transforms.append(np.asarray(t, np.float32).reshape(4, 4).T)  # column-major float[16] -> matrix
        hs.add_instance(first, len(ci), len(cp), mat, t, blas)
    hs.finalize()
    return hs, cp, transforms


def test_clipped_instance_boxes_change_no_result_and_skip_instances(pt, ora, wide_ctx, devbuf, monkeypatch):
    """The 4-wide layout's top level holds each instance's box clipped to the world box of its transformed mesh
    (wide_bvh.h).  On rotated flat ellipsoids - incoherent rays, and rays aimed at every vertex, among them the ones
    that lie on the faces of the clipped boxes - the records equal the oracle's, and fewer instances are entered than
    with the reference's boxes."""
    hs, corners, transforms = _rotated_ellipsoids_scene(pt)
    desc = hs.desc()
    n_inc = 200000
    rays_d = devbuf(n_inc * 32)
    wide_ctx.upload(desc)
    wide_ctx.generate_incoherent_rays(rays_d.ptr, n_inc, 77)
    wide_ctx.synchronize()
    inc = rays_d.download(pt.abi.RAY, n_inc)
    # rays from outside towards the world position of every vertex of every ellipsoid, from three sides each
    targets = []
    for m in transforms:
        for c in corners:
            targets.append((m @ np.append(c, 1.0).astype(np.float32))[:3])
    targets = np.asarray(targets, np.float32)
    origins, dirs = [], []
    for axis in range(3):
        o = targets.copy()
        o[:, axis] += 20.0
        origins.append(o)
        dirs.append(targets - o)
    aimed = rays_from_arrays(pt, np.concatenate(origins), np.concatenate(dirs))
    rays = np.concatenate([inc, aimed])
    n = len(rays)
    ref = ora.trace_closest(desc, rays, nthreads=8)
    assert (ref["inst"] >= 0).sum() > n // 20

    import torch

    def run():
        buf, hits_d, ctr_d = devbuf(n * 32), devbuf(n * 32), devbuf(n * 16)
        buf.upload(torch, rays)
        wide_ctx.trace_closest_counted(buf.ptr, n, hits_d.ptr, ctr_d.ptr)
        wide_ctx.synchronize()
        return hits_d.download(pt.abi.HIT, n), ctr_d.download(np.uint32, n * 4).reshape(n, 4)

    def check(h):
        # incoherent rays: the usual bar.  A ray through a vertex meets all the triangles around it at one
        # distance, and which of them a layout reports is its visiting order: there the instance, the distance and
        # hit-against-miss have to agree, not the triangle.
        _check_closest(h[:n_inc], ref[:n_inc], n_inc)
        a, r = h[n_inc:], ref[n_inc:]
        assert np.array_equal(a["inst"] >= 0, r["inst"] >= 0)
        both = r["inst"] >= 0
        assert both.sum() > len(r) // 2
        assert (a["inst"][both] != r["inst"][both]).sum() <= 2  # two ellipsoids can meet at a vertex
        same = both & (a["inst"] == r["inst"])
        assert np.max(np.abs(a["t"][same] - r["t"][same]) / np.abs(r["t"][same])) <= 1e-5

    hits, ctr = run()
    check(hits)
    monkeypatch.setenv("PTAMD_LOOSE_INSTANCE_BOXES", "1")
    wide_ctx.upload(desc)
    hits_loose, ctr_loose = run()
    monkeypatch.delenv("PTAMD_LOOSE_INSTANCE_BOXES")
    wide_ctx.upload(desc)
    check(hits_loose)
    # the clipped boxes only skip instances that hold nothing for the ray: the records do not depend on them
    differ = (hits["prim"] != hits_loose["prim"]) | (hits["inst"] != hits_loose["inst"]) | (hits["t"].view(np.uint32) != hits_loose["t"].view(np.uint32))
    assert differ.sum() <= 2, int(differ.sum())
    entered, entered_loose = int(ctr[:, 1].sum()), int(ctr_loose[:, 1].sum())
    assert entered < 0.7 * entered_loose, (entered, entered_loose)


@pytest.mark.parametrize("depth", [40, 100, 200])
def test_wide_deep_tree_keeps_every_entry(pt, ora, wide_ctx, depth):
    """The wide traversal has no 64-entry limit to reproduce: on a vine-shaped BLAS deeper than the
    reference's stack it returns the true closest triangle (the brute-force answer), where the
    reference - and trace mode 0 - lose pushes and report a farther one.  Its own stack (LDS part plus
    spill area, 128 entries or more) does end somewhere: beyond that pushes are dropped and counted."""
    desc, keep = make_vine_scene(pt, depth, True)
    wide_ctx.upload(desc)
    wide_ctx.reset_stats()
    n = 256
    rs = np.random.RandomState(depth)
    o = np.concatenate([np.full((n, 1), -10.0), rs.uniform(0.05, 0.45, size=(n, 2))], axis=1).astype(np.float32)
    d = np.tile(np.array([[1, 0, 0]], np.float32), (n, 1))
    rays = rays_from_arrays(pt, o, d)
    hits = wide_ctx.trace_closest_host(rays)
    if depth == 200:  # three leaves pushed per level, 66 levels: more than the stack holds
        assert wide_ctx.stats().stack_drops > 0 and (hits["prim"] >= 0).all()
        return
    brute = ora.trace_naive(desc, rays)
    assert (hits["prim"] == depth).all()
    assert np.array_equal(hits["prim"], brute["prim"]) and np.array_equal(hits["t"].view(np.uint32), brute["t"].view(np.uint32))
    assert wide_ctx.stats().stack_drops == 0


def test_trace_mode_argument_is_checked(pt, gpu_ctx):
    for bad in (-1, 4):
        with pytest.raises(pt.PtError, match="trace mode"):
            gpu_ctx.set_trace_mode(bad)
    gpu_ctx.set_trace_mode(0)


@pytest.mark.parametrize("scale", [0.25, 3.0])
def test_wide_closest_hit_with_directions_that_are_not_unit_length(pt, ora, wide_ctx, devbuf, scene_factory, scale):
    """pt_ray does not require unit directions.  The distance the reference reports is dot(p - o, d) = t * |d|^2,
    which is no bound on the ray parameter t unless |d| = 1, so for other rays the 4-wide layout searches every
    instance unbounded.  With |d| > 1 that is the reference's answer.  With |d| < 1 the reference itself culls
    top-level boxes (entry parameter t) against that smaller number (kernel/intersection.h:209-210) and can lose
    the nearest instance, depending on the order of its own top-level tree; trace mode 0 reproduces that, the
    4-wide layout returns the nearest hit: checked against the brute-force search over every triangle."""
    # overlapping instance boxes along every ray; the brute-force search wants a scene with few triangles
    hs = scene_factory("instancing_stress" if scale > 1.0 else "teapot", 128, 96, 5)
    desc = hs.desc()
    wide_ctx.upload(desc)
    n = 200000 if scale > 1.0 else 5000
    rays_d, hits_d = devbuf(n * 32), devbuf(n * 32)
    wide_ctx.generate_incoherent_rays(rays_d.ptr, n, 57)
    rays = rays_d.download(pt.abi.RAY, n).copy()
    rays["direction"] *= np.float32(scale)
    import torch

    rays_d.upload(torch, rays)
    wide_ctx.trace_closest(rays_d.ptr, n, hits_d.ptr)
    wide_ctx.synchronize()
    hits = hits_d.download(pt.abi.HIT, n)
    if scale > 1.0:
        ref = ora.trace_closest(desc, rays, nthreads=8)
        assert (ref["inst"] >= 0).sum() > n // 50
        _check_closest(hits, ref, n)
    else:
        naive = ora.trace_naive(desc, rays, nthreads=8)
        assert (naive["inst"] >= 0).sum() > n // 50
        ids_differ = (hits["prim"] != naive["prim"]) | (hits["inst"] != naive["inst"])
        assert ids_differ.sum() <= 2, f"{ids_differ.sum()} of {n} ids differ from the brute-force search"
        assert not (ids_differ & ((hits["inst"] >= 0) != (naive["inst"] >= 0))).any()
        # ... and the reference-exact mode still is the reference, wrong culls included
        wide_ctx.set_trace_mode(0)
        wide_ctx.trace_closest(rays_d.ptr, n, hits_d.ptr)
        wide_ctx.synchronize()
        exact = hits_d.download(pt.abi.HIT, n)
        ref = ora.trace_closest(desc, rays, nthreads=8)
        for f in ("prim", "inst"):
            assert np.array_equal(exact[f], ref[f]), f
        assert np.array_equal(exact["t"].view(np.uint32), ref["t"].view(np.uint32))
