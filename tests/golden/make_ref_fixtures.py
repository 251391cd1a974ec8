#!/usr/bin/env python3
"""Generates tests/golden/ref_pins.npz from the reference's own code, in the build container only.

`make -C oracle ref` compiles oracle/ref_probe.cpp against the UNMODIFIED reference sources that need
nothing this image lacks (src/math/hash.h, src/utils/stack.h, third_party/tinyobjloader,
third_party/stb) into oracle/_ref/ref_probe; this script runs it and stores inputs and outputs:

  hash_in / hash_out        Math::PCG_Hash known answers (edge values + an LCG sequence)
  stack_script / stack_pops / stack_top   Stack<uint32_t, 64>: overflow (70 pushes), drain, pop on empty
  obj_bytes + obj_*         Scenes/teapot/plane.obj as tinyobj::ObjReader parses it
  jpg_* per helmet texture  stbi_load(..., 3) against PIL's decode of the same file (what the cook
                            step uses): size, histogram of |stb - PIL|, a 64x64 crop of stb's pixels
  png_bytes / png_pixels    stbi_write_png of a test pattern and the pattern itself

The fixture is data (inputs and expected outputs); no reference source text is stored."""
import io
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
REF = "/root/reference"


def run(*args):
    return subprocess.check_output([PROBE, *map(str, args)]).decode()


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    out = {}
    pairs = np.array([[int(x) for x in ln.split()] for ln in run("hash", 4096).splitlines()], np.uint64)
    out["hash_in"], out["hash_out"] = pairs[:, 0].astype(np.uint32), pairs[:, 1].astype(np.uint32)

    # SMALL_EPSILON, EPSILON, ONE_MINUS_EPSILON, PI, HALF_PI, TWO_PI, INV_PI, SQRT3, INV_SQRT3 (src/math/constants.h)
    out["constants_bits"] = np.array([int(x) for x in run("constants").split()], np.uint32)

    script = list(range(1, 71)) + [0] * 66 + [7, 8, 0, 9, 0, 0, 0]  # 70 pushes into 64 slots, 66 pops, then a few more
    lines = run("stack", *script).splitlines()
    out["stack_script"] = np.array(script, np.uint32)
    out["stack_pops"] = np.array([[int(x) for x in ln.split()] for ln in lines[:-1]], np.uint32)
    out["stack_top"] = np.array([int(lines[-1].split()[1])], np.uint32)

    obj = os.path.join(REF, "Scenes", "teapot", "plane.obj")
    out["obj_bytes"] = np.frombuffer(open(obj, "rb").read(), np.uint8)
    for ln in run("obj", obj).splitlines():
        key, n, *vals = ln.split()
        if key in ("vertices", "normals", "texcoords"):
            out["obj_" + key] = np.array(vals, np.float64).astype(np.float32)
        elif key == "shape":
            out["obj_face_vertices"] = np.array(vals, np.int32)
        elif key == "indices":
            out["obj_indices"] = np.array(vals, np.int32).reshape(-1, 3)  # vertex, normal, texcoord per corner

    from PIL import Image
    names = ["Default_albedo.jpg", "Default_metalRoughness.jpg", "Default_emissive.jpg", "Default_AO.jpg", "Default_normal.jpg"]
    with tempfile.TemporaryDirectory() as tmp:
        for name in names:
            path = os.path.join(REF, "Scenes", "helmet", "helmet", name)
            raw = os.path.join(tmp, "px.raw")
            w, h, c = (int(x) for x in run("image", path, raw).split())
            stb = np.fromfile(raw, np.uint8).reshape(h, w, 3)
            pil = np.asarray(Image.open(path).convert("RGB"), np.uint8)
            d = np.abs(stb.astype(np.int16) - pil.astype(np.int16))
            key = "jpg_" + name.split(".")[0]
            out[key + "_size"] = np.array([w, h, c], np.int32)
            out[key + "_absdiff_hist"] = np.bincount(d.reshape(-1), minlength=16)[:16].astype(np.int64)
            out[key + "_crop"] = stb[1000:1064, 1000:1064].copy()
            out[key + "_sha1"] = np.frombuffer(__import__("hashlib").sha1(open(path, "rb").read()).digest(), np.uint8)
            print(name, w, h, c, "max |stb - PIL| =", int(d.max()), "mean", float(d.mean()))
        png = os.path.join(tmp, "t.png")
        run("png", 37, 23, png)
        out["png_bytes"] = np.frombuffer(open(png, "rb").read(), np.uint8)
        x, y, k = np.meshgrid(np.arange(37), np.arange(23), np.arange(3), indexing="xy")
        out["png_pixels"] = ((x * 7 + y * 13 + k * 101 + (x ^ y)) & 0xFF).astype(np.uint8).transpose(0, 1, 2)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(out["png_bytes"].tobytes())).convert("RGB")), out["png_pixels"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_pins.npz"), **out)
    print("wrote ref_pins.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
