"""Render windows (pt_set_render_window) and the image-tile sharding plan built on them, through the C ABI.

A window's pixels must be bit-identical to the same pixels of a full-frame render (same sampler seeds, same
per-pixel order of film additions), pixels outside it stay untouched, and the films of disjoint windows
add up to the full film exactly: each pixel has one non-zero addend."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _full_film(ctx, w, h, passes):
    ctx.set_render_window()
    ctx.integrator_reset(w, h)
    for first, count in passes:
        ctx.render_batch(w, h, first, count)
    ctx.synchronize()
    return ctx.film_download(w, h)


@pytest.mark.parametrize("name,w,h,depth", [("teapot", 200, 120, 6), ("helmet", 203, 117, 8)])
def test_windows_reproduce_the_full_frame_bit_for_bit(pt, gpu_ctx, scene_factory, name, w, h, depth):
    hs = scene_factory(name, w, h, depth)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    passes = [(0, 3), (3, 2)]
    full = _full_film(gpu_ctx, w, h, passes)
    assert full.mean() > 1e-3
    try:
        # windows that start and end anywhere, not only on the 8x8 tiles the path slots are laid out in
        for x0, y0, ww, hh in [(13, 7, 50, 33), (0, 0, w, 8), (w - 9, h - 5, 9, 5), (64, 40, 1, 1)]:
            gpu_ctx.set_render_window(x0, y0, ww, hh)
            gpu_ctx.integrator_reset(w, h)
            gpu_ctx.reset_stats()
            for first, count in passes:
                gpu_ctx.render_batch(w, h, first, count)
            gpu_ctx.synchronize()
            st = gpu_ctx.stats()
            assert st.paths == passes[-1][1] * ww * hh and st.nan_pixels == 0 and st.stack_drops == 0
            film = gpu_ctx.film_download(w, h)
            inside = np.zeros((h, w), bool)
            inside[y0:y0 + hh, x0:x0 + ww] = True
            assert np.array_equal(film[inside].view(np.uint32), full[inside].view(np.uint32))
            assert (film[~inside] == 0).all()
        # the bands of the tile plan, three ranks: their films sum to the full film exactly
        total = np.zeros_like(full)
        for x0, y0, ww, hh in pt.sharding.tile_windows(3, w, h):
            gpu_ctx.set_render_window(x0, y0, ww, hh)
            gpu_ctx.integrator_reset(w, h)
            for first, count in passes:
                gpu_ctx.render_batch(w, h, first, count)
            total += gpu_ctx.film_download(w, h)
        assert np.array_equal(total.view(np.uint32), full.view(np.uint32))
    finally:
        gpu_ctx.set_render_window()
    # and the whole frame again afterwards
    again = _full_film(gpu_ctx, w, h, passes)
    assert np.array_equal(again.view(np.uint32), full.view(np.uint32))


def test_render_job_with_tiles_equals_samples_on_one_rank(pt, gpu_ctx, scene_factory, devbuf):
    w, h = 160, 96
    hs = scene_factory("cornell", w, h, 8)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    done = pt.sharding.render_job(gpu_ctx, w, h, 6, 4, 0, 1)
    by_samples = gpu_ctx.film_download(w, h)
    assert done == 6
    # rank 1 of 2 in the tile plan renders the lower band only; rank 0 the upper: together the same film
    films = []
    for rank in range(2):
        assert pt.sharding.render_job(gpu_ctx, w, h, 6, 4, rank, 2, shard="tiles") == 6
        films.append(gpu_ctx.film_download(w, h))
    assert (films[0][48:] == 0).all() and (films[1][:48] == 0).all()
    assert np.array_equal((films[0] + films[1]).view(np.uint32), by_samples.view(np.uint32))
    with pytest.raises(ValueError):
        pt.sharding.render_job(gpu_ctx, w, h, 1, 1, 0, 1, shard="rows")


def test_window_arguments_are_checked(pt, gpu_ctx, scene_factory):
    w, h = 64, 48
    hs = scene_factory("teapot", w, h, 4)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    for bad in [(-1, 0, 8, 8), (0, 0, w + 1, 8), (60, 0, 8, 8), (0, 44, 8, 8), (0, 0, 0, 8), (4, 4, 0, 0)]:
        with pytest.raises(pt.PtError, match="render window"):
            gpu_ctx.set_render_window(*bad)
    gpu_ctx.set_render_window(8, 8, 16, 16)
    gpu_ctx.integrator_init(w, h)  # a new integrator starts with the whole frame
    gpu_ctx.reset_stats()
    gpu_ctx.render(w, h, 0)
    assert gpu_ctx.stats().paths == w * h
