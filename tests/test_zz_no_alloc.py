"""pt_trace_closest_static (the entry point that allocates nothing: scratch in a file-scope device
array) against pt_trace_closest, through the C ABI: bit-equal hit records on two calls in a row, the
second on a smaller input, plus the size limit.  Collected last on purpose (file name)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hits(pt, gpu_ctx, devbuf, fn, rays_d, n):
    out = devbuf(n * 32)
    fn(rays_d.ptr, n, out.ptr)
    gpu_ctx.synchronize()
    return out.download(pt.abi.HIT, n)


@pytest.fixture(params=[0, 2], ids=["binary", "four_wide"])
def trace_mode(request, gpu_ctx):
    """Both node layouts: each has its own kernel behind the entry point."""
    gpu_ctx.set_trace_mode(request.param)
    yield request.param
    gpu_ctx.set_trace_mode(0)


def test_static_scratch_entry_point_equals_trace_closest(pt, gpu_ctx, devbuf, scene_factory, trace_mode):
    hs = scene_factory("helmet", 128, 96, 5)
    gpu_ctx.upload(hs.desc())
    for n, seed in ((300000, 11), (70001, 12)):  # the second call is smaller and reuses the array's last contents
        rays_d = devbuf(n * 32)
        gpu_ctx.generate_incoherent_rays(rays_d.ptr, n, seed)
        ref = _hits(pt, gpu_ctx, devbuf, gpu_ctx.trace_closest, rays_d, n)
        got = _hits(pt, gpu_ctx, devbuf, gpu_ctx.trace_closest_static, rays_d, n)
        assert (ref["inst"] >= 0).sum() > n // 50
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))


def test_static_scratch_deep_stacks_use_the_static_spill_area(pt, gpu_ctx, devbuf, trace_mode):
    """A vine-shaped BLAS 100 levels deep pushes the traversal stack far beyond its LDS part."""
    from conftest import make_vine_scene, rays_from_arrays

    desc, keep = make_vine_scene(pt, 100, True)
    gpu_ctx.upload(desc)
    n = 4096
    rng = np.random.default_rng(3)
    o = np.zeros((n, 3), np.float32)
    o[:, 0] = -10.0
    o[:, 1:] = rng.uniform(0.05, 0.45, (n, 2)).astype(np.float32)
    d = np.tile(np.array([[1, 0, 0]], np.float32), (n, 1))
    rays = rays_from_arrays(pt, o, d)
    import torch

    rays_d = devbuf(n * 32)
    rays_d.upload(torch, rays)
    ref = _hits(pt, gpu_ctx, devbuf, gpu_ctx.trace_closest, rays_d, n)
    got = _hits(pt, gpu_ctx, devbuf, gpu_ctx.trace_closest_static, rays_d, n)
    assert (ref["inst"] >= 0).all()
    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))


def test_static_scratch_size_limit(pt, gpu_ctx, devbuf, scene_factory):
    hs = scene_factory("teapot", 64, 64, 4)
    gpu_ctx.upload(hs.desc())
    buf = devbuf(64)
    with pytest.raises(pt.PtError) as e:
        gpu_ctx.trace_closest_static(buf.ptr, (1 << 26) + 1, buf.ptr)
    assert e.value.code == pt.abi.PT_ERR_LIMIT and "limit" in str(e.value)
    gpu_ctx.trace_closest_static(buf.ptr, 0, buf.ptr)  # empty input: nothing to do
