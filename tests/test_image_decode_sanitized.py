"""The image decoder (host/image_decode.cpp) reads untrusted bytes: a few thousand damaged JPEG and PNG files go through a
copy of it built with the address and undefined-behaviour sanitizers (tests/harness/image_decode_fuzz.cpp).  A finding
aborts the harness; every file has to end in an image or an error."""
import io
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

Image = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _picture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 7.0 + y / 13.0), 127 + 120 * np.cos(x / 11.0) * np.sin(y / 5.0), (x * 3 + y * 2) % 256], -1)
    return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)


def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))


def _seed_files():
    files = []
    rgb = _picture(45, 31, 3)
    for kw in (dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, progressive=True), dict(subsampling=0, progressive=True),
               dict(subsampling=2, restart_marker_blocks=3), dict(subsampling=2, restart_marker_rows=1, progressive=True), dict(subsampling=1, optimize=True, quality=30)):
        b = io.BytesIO()
        Image.fromarray(rgb).save(b, format="JPEG", **kw)
        files.append(b.getvalue())
    for mode in ("L", "CMYK"):
        b = io.BytesIO()
        Image.fromarray(rgb).convert(mode).save(b, format="JPEG", progressive=(mode == "L"))
        files.append(b.getvalue())
    for mode in ("RGB", "RGBA", "L", "LA", "P", "1", "I;16"):
        b = io.BytesIO()
        im = Image.fromarray(rgb).convert(mode) if mode != "I;16" else Image.fromarray((rgb[:, :, 0].astype(np.uint16) * 257))
        im.save(b, format="PNG", compress_level=(0 if mode == "L" else 6))
        files.append(b.getvalue())
    # Adam7 and tRNS, assembled by hand
    w, h = 11, 9
    img = _picture(w, h, 4)
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    raw = b"".join(b"".join(b"\x00" + img[y, x0::dx].tobytes() for y in range(y0, h, dy)) for x0, y0, dx, dy in passes if x0 < w and y0 < h)
    files.append(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b""))
    idx = (np.arange(w * h).reshape(h, w) % 5).astype(np.uint8)
    raw = b"".join(b"\x00" + idx[y].tobytes() for y in range(h))
    files.append(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 3, 0, 0, 0)) + _chunk(b"PLTE", bytes(range(15))) +
                 _chunk(b"tRNS", bytes([0, 128])) + _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b""))
    return files


def _damage(data, rng):
    d = bytearray(data)
    kind = int(rng.integers(0, 6))
    if kind == 0:
        d = d[: int(rng.integers(0, len(d)))]
    elif kind == 1:
        for _ in range(int(rng.integers(1, 6))):
            d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
    elif kind == 2:
        a = int(rng.integers(0, len(d)))
        del d[a: a + int(rng.integers(1, 40))]
    elif kind == 3:  # a big-endian 16-bit field (segment lengths, dimensions) set to an extreme
        a = int(rng.integers(0, max(1, len(d) - 2)))
        d[a: a + 2] = bytes(rng.choice([b"\x00\x00", b"\xff\xff", b"\x00\x01", b"\x7f\xff", b"\x80\x00"]))
    elif kind == 4:  # a span repeated
        a = int(rng.integers(0, len(d)))
        span = d[a: a + int(rng.integers(1, 64))]
        d[a:a] = span
    else:  # a 32-bit field (PNG chunk lengths and dimensions)
        a = int(rng.integers(0, max(1, len(d) - 4)))
        d[a: a + 4] = bytes(rng.choice([b"\x00\x00\x00\x00", b"\xff\xff\xff\xff", b"\x7f\xff\xff\xff", b"\x00\x01\x00\x00", b"\x00\x00\x40\x00"]))
    return bytes(d)


def _refix_png_crcs(data):
    """Damaged PNGs mostly die at the chunk checksum; half of them get their checksums recomputed so the damage reaches the decoder."""
    if not data.startswith(b"\x89PNG\r\n\x1a\n"):
        return data
    out, at = bytearray(data[:8]), 8
    while at + 8 <= len(data):
        (n,) = struct.unpack(">I", data[at: at + 4])
        kind = data[at + 4: at + 8]
        body = data[at + 8: at + 8 + n]
        if at + 12 + n > len(data):
            out += data[at:]
            return bytes(out)
        out += data[at: at + 8] + body + struct.pack(">I", zlib.crc32(kind + body))
        at += 12 + n
    out += data[at:]
    return bytes(out)


def test_damaged_images_through_a_sanitizer_build_of_the_decoder(tmp_path):
    exe = tmp_path / "image_decode_fuzz"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                            os.path.join(ROOT, "tests", "harness", "image_decode_fuzz.cpp"), os.path.join(ROOT, "cuda-path-tracer_amd", "host", "image_decode.cpp"),
                            "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    rng = np.random.default_rng(2024)
    seeds = _seed_files()
    blobs = list(seeds)
    for data in seeds:
        for trial in range(120):
            d = _damage(data, rng)
            if trial % 3 == 0:
                d = _damage(d, rng)
            if trial % 2 == 0:
                d = _refix_png_crcs(d)
            blobs.append(d)
    corpus = tmp_path / "corpus.bin"
    with open(corpus, "wb") as f:
        for b in blobs:
            f.write(struct.pack("<I", len(b)) + b)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe), str(corpus)], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-6000:])
    words = run.stdout.split()
    images, errors = int(words[1]), int(words[3])
    print(run.stdout)
    assert images >= 4 * len(seeds) - 8 and errors > 500, run.stdout  # the intact files decode in every channel count they allow
