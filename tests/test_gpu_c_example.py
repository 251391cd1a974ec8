"""The compiled plain-C host program (examples/headless_render.c) against the C ABI alone: scene file in, PNG and
film checkpoint out, no Python between the file and the GPU.  Its film must equal, bit for bit, the film the same
scene renders through the ctypes bindings, and two tile-sharded runs of it must add up to the whole frame."""
import os
import subprocess

import numpy as np
import pytest

from test_scene_io import _current_schema_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "headless_render")


def _run(*args):
    out = subprocess.run([EXE, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert out.returncode == 0, out.stdout.decode()
    return out.stdout.decode()


def test_c_program_renders_the_same_film_as_the_bindings(pt, gpu_ctx, tmp_path):
    if not os.path.exists(EXE):  # normally built by cuda-path-tracer_amd/build.sh together with the library
        subprocess.check_call(["gcc", "-std=c11", "-O1", "-I" + os.path.join(ROOT, "include"), EXE + ".c", "-o", EXE,
                               "-L" + os.path.join(ROOT, "cuda-path-tracer_amd"), "-lptamd", "-Wl,-rpath," + os.path.join(ROOT, "cuda-path-tracer_amd"),
                               "-Wl,-rpath-link,/opt/rocm/lib"])
    path = _current_schema_scene(tmp_path)
    iterations, batch = 6, 4
    text = _run(path, tmp_path / "whole", iterations, batch)
    assert "32x24, 6 iterations" in text and "1 image(s)" in text
    film, done = pt.checkpoint_read(str(tmp_path / "whole.film"))
    assert done == iterations and os.path.getsize(tmp_path / "whole.6samp.png") > 100

    hs = pt.HostScene.load(path, ctx=gpu_ctx)
    w, h = hs.width, hs.height
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    for it in range(iterations):
        gpu_ctx.render(w, h, it)
    expected = gpu_ctx.film_download(w, h)
    assert expected.mean() > 1e-3
    assert np.array_equal(film.view(np.uint32), expected.view(np.uint32))

    # two ranks of an image-tile sharded job, each its own process
    total = np.zeros_like(film)
    for rank in range(2):
        _run(path, tmp_path / f"tile{rank}", iterations, batch, rank, 2)
        part, done = pt.checkpoint_read(str(tmp_path / f"tile{rank}.film"))
        assert done == iterations
        rows = slice(0, 16) if rank == 0 else slice(16, 24)
        outside = np.ones(h, bool)
        outside[rows] = False
        assert (part[outside] == 0).all() and part[rows].any()
        total += part
    assert np.array_equal(total.view(np.uint32), expected.view(np.uint32))


def test_c_program_loads_image_textures_with_the_library_decoder(pt, gpu_ctx, tmp_path):
    """A scene with PATH textures and a glTF with embedded images, plus a JPEG of the helmet scene as a further
    texture: the plain-C host decodes them with pth_decode_image and renders the film the bindings render."""
    import json

    from test_scene_loader_cpp import _textured_scene

    path = _textured_scene(tmp_path)
    spec = pt.SceneSpec.load(os.path.join(ROOT, "scenes", "helmet.npz"))
    jpeg = next(spec.arrays[o["data"]].tobytes() for o in spec.ops if o["op"] == "image" and o.get("encoded"))
    (tmp_path / "helmet_albedo.jpg").write_bytes(jpeg)
    scene = json.loads(open(path).read())
    first = next(iter(scene["materials"]))
    scene["materials"]["zz_jpeg"] = {"TYPE": "lambertian", "ALBEDO": {"PATH": "helmet_albedo.jpg"}}
    scene["objects"].append({"TYPE": "obj", "PATH": "quad.obj", "MATERIAL": "zz_jpeg", "TRANS": [-0.6, 0.2, 0.1], "ROTAT": [0, 20, 0], "SCALE": [0.8, 0.8, 0.8]})
    path2 = str(tmp_path / "textured_jpeg.json")
    open(path2, "w").write(json.dumps(scene))
    assert first
    iterations = 4
    _run(path2, tmp_path / "tex", iterations, 2)
    film, done = pt.checkpoint_read(str(tmp_path / "tex.film"))
    assert done == iterations
    hs = pt.HostScene.load(path2, ctx=gpu_ctx)  # the same loader and decoder through the bindings
    w, h = hs.width, hs.height
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    for it in range(iterations):
        gpu_ctx.render(w, h, it)
    expected = gpu_ctx.film_download(w, h)
    assert expected.mean() > 1e-3
    assert np.array_equal(film.view(np.uint32), expected.view(np.uint32))
