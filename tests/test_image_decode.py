"""The library's own image decoder (host/image_decode.cpp, pth_decode_image: the pth_image_decoder that the C route
hands to pth_scene_load) against PIL on files made here, against the reference's decoder where its output is on
record (tests/golden/ref_pins.npz: 64x64 crops of stb_image's decode of the helmet textures, whose files the cooked
helmet scene carries), and against damaged files.  JPEG: within 3 / 255 per channel of the reference's decoder and of
libjpeg on the reference's files (4 on synthetic 4:2:2 files, see there); PNG: exact."""
import ctypes as C
import hashlib
import io
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Image = pytest.importorskip("PIL.Image")


def _decode(pt, data, want=0):
    lib = pt.abi.load_library()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    if lib.pth_decode_image(None, buf, len(data), want, C.byref(w), C.byref(h), C.byref(c), None, 0):
        raise ValueError(lib.pth_decode_image_error().decode())
    out = np.zeros((h.value, w.value, c.value), np.uint8)
    if lib.pth_decode_image(None, buf, len(data), want, C.byref(w), C.byref(h), C.byref(c), out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size):
        raise ValueError(lib.pth_decode_image_error().decode())
    return out


def _picture(w, h, seed=5):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 7.0 + y / 13.0), 127 + 120 * np.cos(x / 11.0) * np.sin(y / 5.0), (x * 3 + y * 2) % 256], -1).astype(np.float64)
    img += rng.normal(0, 12, img.shape)
    img[h // 3: h // 3 + 9, :, :] = [255, 0, 0]  # hard chroma edges
    img[:, w // 2: w // 2 + 5, :] = [0, 0, 255]
    return np.clip(img, 0, 255).astype(np.uint8)


def _pil(data):
    ref = np.asarray(Image.open(io.BytesIO(data)))
    return ref[:, :, None] if ref.ndim == 2 else ref


JPEG_VARIANTS = [dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, progressive=True), dict(subsampling=0, progressive=True),
                 dict(subsampling=1, progressive=True, optimize=True), dict(subsampling=2, restart_marker_blocks=3), dict(subsampling=2, quality=35),
                 dict(subsampling=2, quality=98, progressive=True), dict(subsampling=2, restart_marker_rows=1, progressive=True)]


@pytest.mark.parametrize("size", [(64, 48), (37, 23), (129, 67), (16, 16), (1, 1), (300, 201)])
def test_jpeg_baseline_and_progressive_against_libjpeg(pt, size):
    """Noisy pictures with hard chroma edges, every sampling / scan layout PIL writes.  Bound 4, not 3: libjpeg rounds its
    (3, 1) / 4 chroma interpolation of 4:2:2 files with an alternating bias where this decoder (like stb_image) rounds to
    nearest, which shows as one more unit on a pixel in ten thousand; the files of the reference's scenes stay within 3."""
    im = Image.fromarray(_picture(*size))
    for kw in JPEG_VARIANTS:
        for mode in ("RGB", "L"):
            b = io.BytesIO()
            im.convert(mode).save(b, format="JPEG", quality=kw.get("quality", 85), **{k: v for k, v in kw.items() if k != "quality"})
            ref, got = _pil(b.getvalue()), _decode(pt, b.getvalue())
            assert got.shape == ref.shape, (kw, mode)
            diff = np.abs(got.astype(int) - ref.astype(int))
            assert diff.max() <= 4 and (diff > 3).mean() <= 1e-3, (kw, mode)
    rgba = _decode(pt, b.getvalue(), 4)  # the last file was grey: expanded to RGBA on request
    assert rgba.shape[2] == 4 and (rgba[:, :, 3] == 255).all() and np.array_equal(rgba[:, :, 0], rgba[:, :, 2])


@pytest.mark.parametrize("size", [(37, 23), (64, 48), (5, 3), (1, 1)])
def test_png_of_every_colour_type_is_exact(pt, size):
    base = _picture(*size)
    cases = {"RGB": Image.fromarray(base), "RGBA": Image.fromarray(np.dstack([base, base[:, :, :1]])), "L": Image.fromarray(base[:, :, 0]),
             "LA": Image.fromarray(np.dstack([base[:, :, 0], base[:, :, 1]]), "LA"), "P": Image.fromarray(base).quantize(17),
             "1": Image.fromarray(base[:, :, 0]).convert("1"), "I;16": Image.fromarray(base[:, :, 0].astype(np.uint16) * 257)}
    for name, im in cases.items():
        for kw in (dict(), dict(compress_level=0), dict(compress_level=9, optimize=True)):
            b = io.BytesIO()
            im.save(b, format="PNG", **kw)
            r = Image.open(io.BytesIO(b.getvalue()))
            if name == "P":
                ref = np.asarray(r.convert("RGB"))
            elif name == "1":
                ref = np.asarray(r.convert("L"))[:, :, None]
            elif name == "I;16":
                ref = (np.asarray(r).astype(np.uint16) >> 8).astype(np.uint8)[:, :, None]
            else:
                ref = _pil(b.getvalue())
            got = _decode(pt, b.getvalue())
            assert got.shape == ref.shape and np.array_equal(got, ref), (name, kw)
        assert _decode(pt, b.getvalue(), 3).shape[2] == 3 and _decode(pt, b.getvalue(), 4).shape[2] == 4


def test_png_interlaced_and_transparent_palette(pt):
    """Adam7 files and tRNS chunks, which PIL cannot write on request: assembled by hand from zlib streams."""
    import struct
    import zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    w, h = 11, 9
    img = _picture(w, h)
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    raw = b"".join(b"".join(b"\x00" + img[y, x0::dx].tobytes() for y in range(y0, h, dy)) for x0, y0, dx, dy in passes if x0 < w and y0 < h)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    assert np.array_equal(_decode(pt, png), img)
    assert np.array_equal(_decode(pt, png), _pil(png))
    idx = (np.arange(w * h).reshape(h, w) % 5).astype(np.uint8)
    palette = bytes(range(15))
    raw = b"".join(b"\x00" + idx[y].tobytes() for y in range(h))
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 3, 0, 0, 0)) + chunk(b"PLTE", palette) + chunk(b"tRNS", bytes([0, 128])) +
           chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    got = _decode(pt, png)
    assert got.shape == (h, w, 4) and np.array_equal(got, np.asarray(Image.open(io.BytesIO(png)).convert("RGBA")))


def _helmet_files(pt):
    spec = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz"))
    return {hashlib.sha1(spec.arrays[o["data"]].tobytes()).digest(): spec.arrays[o["data"]].tobytes() for o in spec.ops if o["op"] == "image" and o.get("encoded")}


def test_helmet_textures_within_3_of_libjpeg_and_of_the_reference_decoder(pt):
    """The 2048x2048 JPEGs the helmet scene loads (albedo: baseline; metallic-roughness: progressive), full frame against
    PIL's libjpeg, and the 64x64 crops that the reference's stb_image produced from the same files (ref_pins.npz)."""
    pins = np.load(os.path.join(ROOT, "tests", "golden", "ref_pins.npz"))
    files = _helmet_files(pt)
    assert len(files) >= 2
    progressive = 0
    for raw in files.values():
        got = _decode(pt, raw)
        assert np.abs(got.astype(int) - _pil(raw).astype(int)).max() <= 3
        progressive += b"\xff\xc2" in raw[:4096]
    assert progressive >= 1
    checked = 0
    for name in ("Default_albedo", "Default_metalRoughness", "Default_emissive", "Default_AO", "Default_normal"):
        raw = files.get(pins["jpg_" + name + "_sha1"].tobytes())
        if raw is None:
            continue
        w, h, c = pins["jpg_" + name + "_size"]
        got = _decode(pt, raw, 3)
        assert got.shape == (h, w, 3)
        assert np.abs(got[1000:1064, 1000:1064].astype(int) - pins["jpg_" + name + "_crop"].astype(int)).max() <= 3, name
        checked += 1
    assert checked >= 2


def test_damaged_files_end_in_an_error_or_an_image_never_in_a_crash(pt):
    rng = np.random.default_rng(11)
    im = Image.fromarray(_picture(67, 41))
    files = []
    for kw in (dict(format="JPEG", subsampling=2), dict(format="JPEG", subsampling=2, progressive=True), dict(format="PNG"), dict(format="PNG", compress_level=0)):
        b = io.BytesIO()
        im.save(b, **kw)
        files.append(b.getvalue())
    outcomes = {"image": 0, "error": 0}
    for data in files:
        for trial in range(150):
            d = bytearray(data)
            kind = trial % 3
            if kind == 0:
                d = d[: rng.integers(0, len(d))]
            elif kind == 1:
                for _ in range(rng.integers(1, 6)):
                    d[rng.integers(0, len(d))] = rng.integers(0, 256)
            else:
                a = rng.integers(0, len(d))
                d[a: a + rng.integers(1, 40)] = b""
            try:
                _decode(pt, bytes(d))
                outcomes["image"] += 1
            except ValueError:
                outcomes["error"] += 1
    assert outcomes["error"] > 100 and outcomes["image"] + outcomes["error"] == 600
    with pytest.raises(ValueError, match="neither"):
        _decode(pt, b"GIF89a" + bytes(100))
    with pytest.raises(ValueError, match="want_channels"):
        _decode(pt, files[0], 2)
