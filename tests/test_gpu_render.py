"""GPU wavefront path tracer against the CPU oracle's scalar path loop, through the C ABI and
the host facade (pt_render, film access, tone mapping, PathIntegrator).

Same-seed parity: both sides draw the same PCG streams in the same order, traversal is bit-exact, every
arithmetic operation of sampling and shading is a correctly rounded fp32 operation in the same order on both
sides (-ffp-contract=off), and the one pair of library functions on the path, sinf / cosf of a sampled azimuth,
is defined as a fixed sequence of such operations (sincos_azimuth in csrc/dmath.h, restated in the oracle).
So the film is **bit-identical** to the oracle's: every channel of every pixel, after one sample and after
sixteen (tests/parity_report.py, tools/film_exactness.py: 0 of 129 600 pixels differ on seven scenes in trace
modes 0 and 1).  The one exception is the sRGB texture option, which calls powf."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _films(pt, ora, gpu_ctx, hs, w, h, iterations):
    desc = hs.desc()
    gpu_ctx.upload(desc)
    gpu_ctx.integrator_init(w, h)
    gpu_ctx.reset_stats()  # the context is shared with tests that overflow the stack on purpose
    ref = None
    for it in range(iterations):
        gpu_ctx.render(w, h, it)
        ref, _ = ora.render(desc, w, h, it, ref, nthreads=8)
    return gpu_ctx.film_download(w, h), ref, desc


@pytest.mark.parametrize("name,w,h,depth", [("tiny", 64, 48, 6), ("teapot", 96, 96, 6), ("cornell", 80, 80, 8),
                                            ("pbr_metallic", 96, 54, 8), ("pbr_ior", 96, 54, 8), ("helmet", 128, 72, 8),
                                            ("instancing_stress", 64, 64, 5)])
def test_one_sample_matches_oracle_per_pixel(pt, ora, gpu_ctx, scene_factory, tiny_scene, name, w, h, depth):
    hs = tiny_scene if name == "tiny" else scene_factory(name, w, h, depth)
    film, ref, _ = _films(pt, ora, gpu_ctx, hs, w, h, 1)
    assert np.isfinite(film).all()
    assert ref.mean() > 1e-3
    differ = (film.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    assert not differ.any(), f"{name}: {int(differ.sum())} of {w * h} pixels differ from the oracle's film"
    st = gpu_ctx.stats()
    assert st.paths == w * h and st.closest_rays >= w * h and st.nan_pixels == 0 and st.stack_drops == 0


def test_ray_counts_match_oracle(pt, ora, gpu_ctx, scene_factory):
    """Extension-ray counts are the oracle's; the GPU skips shadow rays whose
    contribution is zero by construction (cos <= 0), so its shadow count is a lower bound."""
    w, h = 96, 96
    hs = scene_factory("teapot", w, h, 6)
    desc = hs.desc()
    gpu_ctx.upload(desc)
    gpu_ctx.integrator_init(w, h)
    gpu_ctx.render(w, h, 0)
    st = gpu_ctx.stats()
    _, (closest, shadow) = ora.render(desc, w, h, 0, nthreads=8)
    assert int(st.closest_rays) == closest
    assert 0.5 * shadow <= st.shadow_rays <= shadow
    assert st.queue_sizes[0] == w * h and st.queue_sizes[1] < st.queue_sizes[0]


@pytest.mark.parametrize("name", ["teapot", "pbr_metallic"])
def test_accumulated_film_is_bit_identical(pt, ora, gpu_ctx, scene_factory, name):
    w, h, n = 96, 96, 16
    hs = scene_factory(name, w, h, 6)
    film, ref, _ = _films(pt, ora, gpu_ctx, hs, w, h, n)
    assert ref.mean() > 1e-2
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))


def test_render_is_bitwise_reproducible(pt, gpu_ctx, scene_factory):
    """Queue order varies from run to run (atomics); the film must not."""
    w, h = 128, 72
    hs = scene_factory("helmet", w, h, 8)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    films = []
    for _ in range(3):
        for it in range(3):
            gpu_ctx.render(w, h, it)
        films.append(gpu_ctx.film_download(w, h))
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))
    assert np.array_equal(films[0].view(np.uint32), films[2].view(np.uint32))


def test_iteration_zero_resets_and_iterations_accumulate(pt, gpu_ctx, tiny_scene):
    w, h = 64, 48
    gpu_ctx.upload(tiny_scene.desc())
    gpu_ctx.integrator_init(w, h)
    gpu_ctx.render(w, h, 0)
    a = gpu_ctx.film_download(w, h)
    gpu_ctx.render(w, h, 1)
    ab = gpu_ctx.film_download(w, h)
    gpu_ctx.integrator_reset(w, h)
    assert not gpu_ctx.film_download(w, h).any()
    gpu_ctx.render(w, h, 1)
    b = gpu_ctx.film_download(w, h)
    assert np.array_equal((a + b).view(np.uint32), ab.view(np.uint32))
    gpu_ctx.render(w, h, 0)  # iteration 0 clears the film first (path_trace.cu:438-439)
    assert np.array_equal(gpu_ctx.film_download(w, h).view(np.uint32), a.view(np.uint32))


def test_sample_sharded_film_sum_equals_single_film(pt, gpu_ctx, tiny_scene, devbuf):
    """What the multi-GPU path does (rank r renders iterations r, r+R, ...; films are summed):
    emulated on one GPU with two contexts and a rank-ordered sum."""
    w, h, total = 64, 48, 6
    desc = tiny_scene.desc()
    gpu_ctx.upload(desc)
    gpu_ctx.integrator_init(w, h)
    for it in range(total):
        gpu_ctx.render(w, h, it)
    single = gpu_ctx.film_download(w, h).astype(np.float64)
    shards = []
    for r in range(2):
        c = pt.Context(0)
        c.upload(desc)
        c.integrator_init(w, h)
        for it in range(r, total, 2):
            c.render(w, h, it) if it > 0 else c.render(w, h, 0)
        shards.append(c.film_download(w, h).astype(np.float64))
        c.close()
    assert np.allclose(shards[0] + shards[1], single, rtol=1e-6, atol=1e-6)


def test_tonemap_matches_oracle_and_reference_quantisation(pt, ora, gpu_ctx, tiny_scene, devbuf):
    w, h = 64, 48
    gpu_ctx.upload(tiny_scene.desc())
    gpu_ctx.integrator_init(w, h)
    rgba = devbuf(w * h * 4)
    for it in range(4):
        gpu_ctx.render(w, h, it, rgba.ptr)
    film = gpu_ctx.film_download(w, h)
    img = rgba.download(np.uint8, w * h * 4).reshape(h, w, 4)
    ref = ora.tonemap(film, iteration=3)
    assert (img[..., 3] == 255).all()
    diff = np.abs(img.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01  # fp division rounding at a quantisation step
    # explicit tone mapping of an arbitrary device film
    film_d = devbuf(w * h * 12)
    import torch
    film_d.upload(torch, film)
    gpu_ctx.tonemap(film_d.ptr, w, h, 4, rgba.ptr)
    assert np.array_equal(rgba.download(np.uint8, w * h * 4).reshape(h, w, 4), img)


def test_film_device_copies_roundtrip(pt, gpu_ctx, tiny_scene, devbuf):
    w, h = 64, 48
    gpu_ctx.upload(tiny_scene.desc())
    gpu_ctx.integrator_init(w, h)
    gpu_ctx.render(w, h, 0)
    film = gpu_ctx.film_download(w, h)
    buf = devbuf(w * h * 12)
    gpu_ctx.film_copy_to_device(buf.ptr)
    assert np.array_equal(buf.download(np.float32, w * h * 3).reshape(h, w, 3), film)
    gpu_ctx.integrator_reset(w, h)
    gpu_ctx.film_copy_from_device(buf.ptr)
    assert np.array_equal(gpu_ctx.film_download(w, h), film)


def test_non_multiple_of_eight_resolution_and_resize(pt, ora, gpu_ctx, tiny_scene):
    """Path slots are 8x8 pixel tiles; partial tiles at the right / bottom edge must not leak."""
    hs = tiny_scene
    desc = hs.desc()
    cam = pt.camera_make([0.0, 2.0, 6.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0], 50.0, 61, 37)
    gpu_ctx.upload(desc)
    gpu_ctx.camera_update(cam)
    gpu_ctx.integrator_init(64, 48)
    gpu_ctx.integrator_resize(61, 37)
    gpu_ctx.render(61, 37, 0)
    film = gpu_ctx.film_download(61, 37)
    import ctypes as C
    d2 = pt.abi.SceneDesc.from_buffer_copy(desc)
    d2.camera = cam
    ref, _ = ora.render(d2, 61, 37, 0, nthreads=4)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    assert gpu_ctx.stats().paths == 61 * 37
    del C


def test_path_integrator_facade(pt, ora, gpu_ctx, tiny_scene, devbuf):
    """The C++ PathIntegrator / Scene::UpdateDevice route gives the same film as the raw ABI."""
    w, h = 64, 48
    integ = pt.PathIntegrator(gpu_ctx)
    integ.init(w, h)
    rgba = devbuf(w * h * 4)
    integ.render(w, h, tiny_scene, rgba.ptr, 0)
    integ.render(w, h, tiny_scene, rgba.ptr, 1)
    a = gpu_ctx.film_download(w, h)
    gpu_ctx.upload(tiny_scene.desc())
    gpu_ctx.integrator_init(w, h)
    gpu_ctx.render(w, h, 0)
    gpu_ctx.render(w, h, 1)
    assert np.array_equal(a.view(np.uint32), gpu_ctx.film_download(w, h).view(np.uint32))
    integ.reset(w, h)
    assert not gpu_ctx.film_download(w, h).any()
    integ.close()


def test_deep_path_limit_terminates_early(pt, gpu_ctx, scene_factory):
    """The teapot scene's own depth limit is 100; the bounce loop stops launching once the
    extension queue has drained instead of running all 100 rounds."""
    w, h = 96, 96
    hs = scene_factory("teapot", w, h, None)
    assert hs.max_depth == 100
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    gpu_ctx.render(w, h, 0)
    st = gpu_ctx.stats()
    assert 3 < st.bounces_run < 100
    assert np.isfinite(gpu_ctx.film_download(w, h)).all()


def test_batched_samples_equal_separate_renders(pt, gpu_ctx, scene_factory):
    """pt_render_batch keeps several iterations in flight in one wavefront pass; the film must
    be bit-identical to rendering the same iterations one call at a time."""
    w, h = 96, 72
    hs = scene_factory("teapot", w, h, 6)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    for it in range(7):
        gpu_ctx.render(w, h, it)
    single = gpu_ctx.film_download(w, h)
    gpu_ctx.render_batch(w, h, 0, 4)
    gpu_ctx.render_batch(w, h, 4, 3)
    gpu_ctx.synchronize()
    batched = gpu_ctx.film_download(w, h)
    assert np.array_equal(single.view(np.uint32), batched.view(np.uint32))
    st = gpu_ctx.stats()
    assert st.paths == 3 * w * h
    with pytest.raises(pt.PtError, match="samples per pass"):
        gpu_ctx.render_batch(w, h, 0, 0)


def test_film_access_waits_for_passes_in_flight(pt, gpu_ctx, scene_factory, devbuf):
    """Film reads, copies and tone mapping are ordered after the passes still running on the
    library's own streams: no pt_synchronize between pt_render_batch and the film access."""
    import torch

    w, h = 160, 120
    hs = scene_factory("helmet", w, h, 8)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    for k in range(3):
        gpu_ctx.render_batch(w, h, 8 * k, 8)
    gpu_ctx.synchronize()
    expected = gpu_ctx.film_download(w, h)
    expected_rgba = gpu_ctx.display_download(w, h, 24)
    for access in ("download", "copy", "tonemap"):
        for k in range(3):
            gpu_ctx.render_batch(w, h, 8 * k, 8)
        if access == "download":
            got = gpu_ctx.film_download(w, h)
        elif access == "copy":
            dev = devbuf(w * h * 12)
            gpu_ctx.film_copy_to_device(dev.ptr)
            got = dev.download(np.float32, w * h * 3).reshape(h, w, 3)
        else:
            dev = devbuf(w * h * 4)
            gpu_ctx.tonemap(0, w, h, 24, dev.ptr)
            assert np.array_equal(dev.download(np.uint8, w * h * 4).reshape(h, w, 4), expected_rgba)
            continue
        assert np.array_equal(got.view(np.uint32), expected.view(np.uint32)), access
    del torch


def test_light_without_geometry_samples_black(pt, ora, gpu_ctx, tiny_scene):
    """An AreaLight whose geometry_instance_index is -1 is valid input: picking it costs one random
    draw and yields a black sample (kernel/light.h:18-19).  Same film as the oracle's."""
    import ctypes as C

    w, h = 64, 48
    arr = tiny_scene.arrays()
    d = pt.abi.SceneDesc.from_buffer_copy(tiny_scene.desc())
    lights = np.zeros(len(arr["lights"]) + 1, pt.abi.AREA_LIGHT)
    lights[:-1] = arr["lights"]
    lights[-1]["power"] = arr["lights"]["power"].sum() * 0.5
    lights[-1]["geometry_instance_index"] = -1
    prob, alias = pt.alias_build(lights["power"].astype(np.float32))
    table = pt.abi.AliasTable()
    table.alias_table = alias.ctypes.data_as(C.POINTER(C.c_uint32))
    table.probability_table = prob.ctypes.data_as(C.POINTER(C.c_float))
    table.size = len(lights)
    d.lights, d.light_count = lights.ctypes.data, len(lights)
    d.light_sampler = C.pointer(table)
    d.total_light_power = float(np.float32(lights["power"].sum()))
    gpu_ctx.upload(d)
    gpu_ctx.integrator_init(w, h)
    ref = None
    for it in range(2):
        gpu_ctx.render(w, h, it)
        ref, _ = ora.render(d, w, h, it, ref, nthreads=8)
    film = gpu_ctx.film_download(w, h)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32)) and ref.mean() > 1e-3
    # the extra light takes a third of the picks: the film is darker than the plain scene's
    gpu_ctx.upload(tiny_scene.desc())
    gpu_ctx.integrator_init(w, h)
    for it in range(2):
        gpu_ctx.render(w, h, it)
    assert film.mean() < gpu_ctx.film_download(w, h).mean()


@pytest.mark.parametrize("options", [1, 2, 3])
def test_texture_filter_options_match_oracle(pt, ora, gpu_ctx, scene_factory, options):
    """pt_set_texture_filter: 8-bit fixed-point bilinear weights and / or sRGB decode of tagged images
    change the textured helmet's film, and the oracle with the same options agrees with it."""
    w, h = 128, 72
    hs = scene_factory("helmet", w, h, 8)
    desc = hs.desc()
    gpu_ctx.upload(desc)
    gpu_ctx.integrator_init(w, h)
    gpu_ctx.render(w, h, 0)
    default_film = gpu_ctx.film_download(w, h)
    try:
        gpu_ctx.set_texture_filter(options)
        ora.set_texture_filter(options)
        gpu_ctx.upload(desc)  # the options take effect at upload
        gpu_ctx.render(w, h, 0)
        film = gpu_ctx.film_download(w, h)
        ref, _ = ora.render(desc, w, h, 0, nthreads=8)
    finally:
        gpu_ctx.set_texture_filter(0)
        ora.set_texture_filter(0)
        gpu_ctx.upload(desc)
    assert (film != default_film).any(axis=2).mean() > 0.005  # most of this small frame is background
    if options == 1:  # fixed-point weights: exact arithmetic on both sides
        assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    else:  # the sRGB decode calls powf, which the device library and libm round differently
        close = (np.abs(film - ref) <= 1e-4 + 1e-3 * np.abs(ref)).all(axis=2)
        assert close.mean() >= 0.99, close.mean()
    with pytest.raises(pt.PtError, match="texture filter"):
        gpu_ctx.set_texture_filter(4)


def test_passes_in_flight_shrink_when_path_state_does_not_fit(pt, gpu_ctx, scene_factory, monkeypatch):
    """With room for the path state of two passes in flight but not three, the third lane is
    retired and rendering goes on: same film, bit for bit.  A pass that does not fit even once is an
    error that names what it needed, and leaves the context usable."""
    w, h = 96, 72
    hs = scene_factory("teapot", w, h, 6)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    for k in range(6):
        gpu_ctx.render_batch(w, h, 4 * k, 4)
    gpu_ctx.synchronize()
    expected = gpu_ctx.film_download(w, h)
    # 6912 path slots per sample x 184 B = 1.3 MB: 12 samples in flight are 15.6 MB, 8 are 10.4 MB
    monkeypatch.setenv("PTAMD_PATH_STATE_LIMIT_MB", "13")
    small = pt.Context(0)
    try:
        small.upload(hs.desc())
        small.integrator_init(w, h)
        for k in range(6):  # the third pass asks for a third lane's state: refused by the limit, so it must fall back
            small.render_batch(w, h, 4 * k, 4)
        small.synchronize()
        assert np.array_equal(small.film_download(w, h).view(np.uint32), expected.view(np.uint32))
        with pytest.raises(pt.PtError, match="does not fit"):
            small.render_batch(w, h, 24, 16)
        small.render_batch(w, h, 0, 4)  # iteration 0 clears the film
        for k in range(1, 6):
            small.render_batch(w, h, 4 * k, 4)
        small.synchronize()
        assert np.array_equal(small.film_download(w, h).view(np.uint32), expected.view(np.uint32))
    finally:
        small.close()


def test_large_frame_is_deterministic_and_counts_match_oracle(pt, ora, gpu_ctx, scene_factory):
    """Queue staging, grouped work fetches and lane refill only show their corner cases with many
    waves in flight: at 960x540 the per-bounce ray counts must repeat exactly from call to call and
    agree with the oracle's scalar path loop, and the film must repeat bit for bit."""
    import torch

    w, h = 960, 540
    hs = scene_factory("helmet", w, h, 8)
    desc = hs.desc()
    gpu_ctx.upload(desc)
    gpu_ctx.integrator_init(w, h)
    buf = torch.empty(w * h * 8, dtype=torch.float32, device="cuda:0")
    counts = [[gpu_ctx.capture_bounce_rays(w, h, 0, b, buf.data_ptr(), w * h) for b in (1, 2, 3)] for _ in range(3)]
    assert counts[0] == counts[1] == counts[2], counts
    ref = [len(ora.capture_bounce_rays(desc, w, h, 0, b)) for b in (1, 2, 3)]
    assert counts[0] == ref, (counts[0], ref)  # every bounce's ray count is the oracle's exactly
    films = []
    for _ in range(2):
        gpu_ctx.render_batch(w, h, 0, 3)
        gpu_ctx.render_batch(w, h, 3, 2)
        gpu_ctx.synchronize()
        films.append(gpu_ctx.film_download(w, h))
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))
    st = gpu_ctx.stats()
    assert st.paths == 2 * w * h and st.stack_drops == 0
