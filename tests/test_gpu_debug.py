"""pt_debug_render (the debug integrator: BVH heat maps and attribute views) against the CPU
oracle's restatement of DebugIntegrator::Render, through the C ABI and the host facade.

Traversal and hit resolution are bit-exact, so the visit counts behind the heat maps and the
attribute images (normals, uv, depth, texture lookups) must be identical.  The heat-map colours
go through expf / powf / logf, which differ by a few ulps between libm and the device library.
Stated tolerance: attribute views bit-exact; heat-map float images |d| <= 1e-5 (values lie in
[0, 1]); 8-bit output within one level, on fewer than 1 % of the channel values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BVH_TYPES = ["PT_BVH_BOUNDS_HIT", "PT_BVH_TLAS", "PT_BVH_TRIANGLES_HIT"]
VIEWS = ["PT_DEBUG_GEOMETRIC_NORMALS", "PT_DEBUG_ALBEDO", "PT_DEBUG_UV", "PT_DEBUG_DEPTH", "PT_DEBUG_METALLIC", "PT_DEBUG_ROUGHNESS"]


def _state(pt, visualize_bvh=False, bvh_type="PT_BVH_BOUNDS_HIT", depth=500, mode="PT_DEBUG_NONE"):
    return pt.abi.IntegratorState(int(visualize_bvh), getattr(pt.abi, bvh_type), depth, getattr(pt.abi, mode))


def _gpu(gpu_ctx, devbuf, w, h, state):
    image, rgba = devbuf(w * h * 12), devbuf(w * h * 4)
    gpu_ctx.debug_render(w, h, state, image.ptr, rgba.ptr)
    gpu_ctx.synchronize()
    return image.download(np.float32, w * h * 3).reshape(h, w, 3), rgba.download(np.uint8, w * h * 4).reshape(h, w, 4)


def _check_rgba(img, ref):
    assert (img[..., 3] == 255).all()
    diff = np.abs(img.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01


@pytest.mark.parametrize("name,w,h", [("tiny", 64, 48), ("helmet", 128, 72), ("instancing_stress", 61, 37), ("cornell", 80, 80)])
def test_bvh_heat_maps_match_oracle(pt, ora, gpu_ctx, devbuf, scene_factory, tiny_scene, name, w, h):
    hs = tiny_scene if name == "tiny" else scene_factory(name, w, h, 5)
    desc = hs.desc()
    gpu_ctx.upload(desc)
    for bvh_type in BVH_TYPES:
        # a small scale saturates part of the map, a large one keeps it in the violet end
        for depth in (40, 500):
            st = _state(pt, True, bvh_type, depth)
            image, rgba = _gpu(gpu_ctx, devbuf, w, h, st)
            ref_image, ref_rgba = ora.debug_render(desc, w, h, st, nthreads=8)
            assert np.isfinite(image).all()
            assert np.abs(image - ref_image).max() <= 1e-5, (name, bvh_type, depth)
            _check_rgba(rgba, ref_rgba)
            if bvh_type == "PT_BVH_TRIANGLES_HIT":
                break  # the triangle map has its own fixed scale
    # the maps are not constant: rays through geometry visit more nodes than rays past it
    assert ref_image.std() > 1e-3


@pytest.mark.parametrize("name,w,h", [("tiny", 64, 48), ("helmet", 128, 72), ("pbr_metallic", 96, 54), ("instancing_stress", 61, 37)])
def test_attribute_views_match_oracle_bit_for_bit(pt, ora, gpu_ctx, devbuf, scene_factory, tiny_scene, name, w, h):
    hs = tiny_scene if name == "tiny" else scene_factory(name, w, h, 5)
    desc = hs.desc()
    gpu_ctx.upload(desc)
    seen = set()
    for mode in VIEWS:
        st = _state(pt, mode=mode)
        image, rgba = _gpu(gpu_ctx, devbuf, w, h, st)
        ref_image, ref_rgba = ora.debug_render(desc, w, h, st, nthreads=8)
        assert np.array_equal(image.view(np.uint32), ref_image.view(np.uint32)), (name, mode)
        assert np.array_equal(rgba, ref_rgba), (name, mode)
        seen.add(image.tobytes())
    assert len(seen) >= 4  # the views really differ from one another


def test_attribute_view_replaces_bvh_view_and_none_writes_nothing(pt, ora, gpu_ctx, devbuf, tiny_scene):
    w, h = 64, 48
    desc = tiny_scene.desc()
    gpu_ctx.upload(desc)
    both = _state(pt, True, "PT_BVH_TLAS", 100, "PT_DEBUG_UV")
    only = _state(pt, mode="PT_DEBUG_UV")
    a, _ = _gpu(gpu_ctx, devbuf, w, h, both)
    b, _ = _gpu(gpu_ctx, devbuf, w, h, only)
    assert np.array_equal(a, b)
    # no view selected: the output buffers keep their contents (debug.cu:216-237 returns early)
    import torch
    image, rgba = devbuf(w * h * 12), devbuf(w * h * 4)
    sentinel = np.full(w * h * 4, 7, np.uint8)
    rgba.upload(torch, sentinel)
    gpu_ctx.debug_render(w, h, _state(pt), image.ptr, rgba.ptr)
    gpu_ctx.synchronize()
    assert np.array_equal(rgba.download(np.uint8, w * h * 4), sentinel)
    ref_image, ref_rgba = ora.debug_render(desc, w, h, _state(pt))
    assert not ref_image.any() and not ref_rgba.any()


def test_bounds_hit_counts_equal_the_counted_query(pt, ora, gpu_ctx, devbuf, scene_factory):
    """The heat map's depth is the sum of the visit counters pt_trace_closest_counted reports for the
    same camera rays: a pixel's colour is HeatMapColor(count, scale), so equal counts give equal
    colours and a larger count never maps to an earlier wavelength."""
    import torch
    w, h = 96, 96
    hs = scene_factory("teapot", w, h, 5)
    desc = hs.desc()
    gpu_ctx.upload(desc)
    rays_d = devbuf(w * h * 32)
    rays = np.zeros(w * h, pt.abi.RAY)
    o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    for y in range(h):
        for x in range(w):
            ora.lib().ora_camera_ray(desc.camera, x, y, 0.5, 0.5, pt.abi.fptr(o), pt.abi.fptr(d))
            rays[y * w + x]["origin"], rays[y * w + x]["direction"], rays[y * w + x]["tmax"] = o, d, np.float32(3.4028235e38)
    rays_d.upload(torch, rays)
    hits_d, ctr_d = devbuf(w * h * 32), devbuf(w * h * 16)
    gpu_ctx.trace_closest_counted(rays_d.ptr, w * h, hits_d.ptr, ctr_d.ptr)
    gpu_ctx.synchronize()
    ctr = ctr_d.download(np.uint32, w * h * 4).reshape(h, w, 4).astype(np.int64)
    visits = ctr[..., 0] + ctr[..., 1] + ctr[..., 3]
    image, _ = _gpu(gpu_ctx, devbuf, w, h, _state(pt, True, "PT_BVH_BOUNDS_HIT", 500))
    by_count = {}
    for v, c in zip(visits.reshape(-1), image.reshape(-1, 3)):
        assert by_count.setdefault(int(v), c.tobytes()) == c.tobytes()
    assert len(by_count) > 8
    tri_image, _ = _gpu(gpu_ctx, devbuf, w, h, _state(pt, True, "PT_BVH_TRIANGLES_HIT"))
    t = np.clip(np.log(ctr[..., 2] + 1.0) / np.log(51.0), 0.0, 1.0)
    assert np.abs(tri_image[..., 0] - t).max() < 1e-5 and np.abs(tri_image[..., 1] - (0.3 * (1 - t) + t)).max() < 1e-5
    assert not tri_image[..., 2].any()


def test_debug_integrator_facade_and_argument_checks(pt, ora, gpu_ctx, devbuf):
    from conftest import make_tiny_scene
    tiny_scene = make_tiny_scene(pt)  # a scene of its own: Scene::UpdateDevice clears the scene's dirty flags
    w, h = 64, 48
    rgba = devbuf(w * h * 4)
    st = _state(pt, mode="PT_DEBUG_GEOMETRIC_NORMALS")
    integ = pt.DebugIntegrator(gpu_ctx)
    integ.init(w, h)
    integ.render(w, h, tiny_scene, rgba.ptr, 0, st)
    gpu_ctx.synchronize()
    _, ref_rgba = ora.debug_render(tiny_scene.desc(), w, h, st)
    assert np.array_equal(rgba.download(np.uint8, w * h * 4).reshape(h, w, 4), ref_rgba)
    with pytest.raises(pt.PtError):
        gpu_ctx.debug_render(w + 1, h, st, 0, rgba.ptr)  # not the camera's resolution
    with pytest.raises(pt.PtError):
        gpu_ctx.debug_render(w, h, pt.abi.IntegratorState(0, 0, 500, 9), 0, rgba.ptr)  # unknown debug_mode
    with pytest.raises(pt.PtError):
        gpu_ctx.debug_render(w, h, _state(pt, True, depth=0), 0, rgba.ptr)
    fresh = pt.Context(0)
    try:
        with pytest.raises(pt.PtError):
            fresh.debug_render(w, h, st, 0, rgba.ptr)  # no scene
    finally:
        fresh.close()
