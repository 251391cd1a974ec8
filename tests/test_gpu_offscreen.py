"""The headless render loop (Application::Run's offscreen branch, application.cpp:58-76, plus
progressive images and film checkpoint / resume) through the host facade."""
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import make_tiny_scene

pytestmark = pytest.mark.gpu


def _read_png_rgb(path):
    """Minimal PNG reader for the writer's own output (8-bit RGB, filter 0 rows)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(kind + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if kind == b"IHDR":
            w, h, depth, colour = struct.unpack(">IIBB", body[:10])
            assert depth == 8 and colour == 2
        elif kind == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 3)


def test_resumed_render_equals_uninterrupted_render(pt, gpu_ctx, tmp_path):
    w, h = 64, 48
    scene = make_tiny_scene(pt)
    # uninterrupted: 12 samples, images every 4
    a = pt.render_offscreen(gpu_ctx, scene, iterations=12, batch=3, save_every=4, output_prefix=str(tmp_path / "full"),
                            checkpoint_path=str(tmp_path / "full.film"))
    assert a == {"first_iteration": 0, "iterations_done": 12, "images_written": 3, "seconds": a["seconds"]} and a["seconds"] > 0
    film_full = gpu_ctx.film_download(w, h)
    for n in (4, 8, 12):
        assert os.path.exists(tmp_path / f"full.{n}samp.png")
    # the same render, killed after 8 samples and resumed in a new context
    b = pt.render_offscreen(gpu_ctx, scene, iterations=8, batch=8, save_every=0, output_prefix=str(tmp_path / "part"),
                            checkpoint_path=str(tmp_path / "part.film"))
    assert b["iterations_done"] == 8 and b["images_written"] == 1
    saved, it = pt.checkpoint_read(str(tmp_path / "part.film"))
    assert it == 8 and saved.shape == (h, w, 3)
    other = pt.Context(0)
    try:
        c = pt.render_offscreen(other, scene, iterations=12, batch=5, output_prefix=str(tmp_path / "part"),
                                checkpoint_path=str(tmp_path / "part.film"), resume=True)
        assert c["first_iteration"] == 8 and c["iterations_done"] == 12 and c["images_written"] == 1
        film_resumed = other.film_download(w, h)
        shown = other.display_download(w, h, 12)
    finally:
        other.close()
    assert np.array_equal(film_full.view(np.uint32), film_resumed.view(np.uint32))
    final, it = pt.checkpoint_read(str(tmp_path / "part.film"))
    assert it == 12 and np.array_equal(final.view(np.uint32), film_full.view(np.uint32))
    # the written image is the display image of the film (truncated to 8 bits once more by nothing: same bytes)
    img = _read_png_rgb(str(tmp_path / "part.12samp.png"))
    assert np.array_equal(img, shown[..., :3])
    assert np.array_equal(img, _read_png_rgb(str(tmp_path / "full.12samp.png")))
    # nothing left to do: a second resume renders no sample and rewrites nothing but the final image
    d = pt.render_offscreen(gpu_ctx, scene, iterations=12, checkpoint_path=str(tmp_path / "part.film"), resume=True)
    assert d["first_iteration"] == 12 and d["iterations_done"] == 12 and d["seconds"] == 0.0


def test_offscreen_argument_and_checkpoint_errors(pt, gpu_ctx, tmp_path):
    scene = make_tiny_scene(pt)
    with pytest.raises(pt.PtError, match="1024"):
        pt.render_offscreen(gpu_ctx, scene, iterations=4, batch=2000)
    bad = tmp_path / "bad.film"
    bad.write_bytes(b"not a film checkpoint at all")
    with pytest.raises(pt.PtError, match="not a film checkpoint"):
        pt.render_offscreen(gpu_ctx, scene, iterations=4, checkpoint_path=str(bad), resume=True)
    # a checkpoint of another resolution is refused
    pt.render_offscreen(gpu_ctx, scene, iterations=2, checkpoint_path=str(tmp_path / "small.film"))
    big = make_tiny_scene(pt)
    big.set_camera([0.0, 2.0, 6.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0], 50.0, 80, 48)
    with pytest.raises(pt.PtError, match="another resolution"):
        pt.render_offscreen(gpu_ctx, big, iterations=4, checkpoint_path=str(tmp_path / "small.film"), resume=True)
    # a missing checkpoint with resume is a fresh start
    r = pt.render_offscreen(gpu_ctx, scene, iterations=2, checkpoint_path=str(tmp_path / "new.film"), resume=True)
    assert r["first_iteration"] == 0 and r["iterations_done"] == 2
