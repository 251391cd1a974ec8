#!/usr/bin/env python3
"""Random damage to scene files (scene JSON, glTF with its buffer, OBJ) fed to the C++ front end
(pth_scene_load): every outcome has to be a loaded scene or a PtError, never a crash.  Meant to run against a
library whose host code is built with the address and undefined-behaviour sanitizers:

    PTAMD_OUT=variant_asan.so bash cuda-path-tracer_amd/build.sh -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -g -shared-libsan
    PTAMD_LIB=$PWD/cuda-path-tracer_amd/variant_asan.so LD_PRELOAD=<libclang_rt.asan-x86_64.so> \\
        ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 python tools/fuzz_scene_loader.py [rounds] [seed]
"""
import os
import random
import re
import sys
import tempfile
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cuda_path_tracer_amd as pt  # noqa: E402
from test_scene_loader_cpp import _textured_scene  # noqa: E402  (scene JSON + glTF + textures written to a directory)

OBJ = """v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0.5\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1/1 2/2/1 3/3/1\nf 2 4 3\nf -1 -2 -3\n"""


def damage(data: bytes, rng: random.Random) -> bytes:
    b = bytearray(data)
    kind = rng.randrange(7)
    if not b:
        return bytes(b)
    if kind == 0:  # flip bytes
        for _ in range(rng.randrange(1, 8)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    elif kind == 1:  # cut a span out
        i = rng.randrange(len(b))
        del b[i:i + rng.randrange(1, 64)]
    elif kind == 2:  # truncate
        del b[rng.randrange(len(b)):]
    elif kind == 3:  # duplicate a span
        i = rng.randrange(len(b))
        span = b[i:i + rng.randrange(1, 64)]
        b[i:i] = span
    elif kind == 4:  # replace a number
        text = b.decode("latin-1")
        nums = list(re.finditer(r"-?\d+(\.\d+)?", text))
        if nums:
            m = rng.choice(nums)
            new = rng.choice(["-1", "0", "4294967295", "1e40", "-0.0", "9999999999999", "nan", "2147483648", "65536", ""])
            text = text[:m.start()] + new + text[m.end():]
        b = bytearray(text.encode("latin-1"))
    elif kind == 5:  # replace a structural character
        pos = [i for i, c in enumerate(b) if c in b'{}[],:"']
        if pos:
            b[rng.choice(pos)] = rng.choice(b'{}[],:" x')
    else:  # swap two spans
        i, j = sorted(rng.randrange(len(b)) for _ in range(2))
        n = rng.randrange(1, 32)
        b[i:i + n], b[j:j + n] = b[j:j + n], b[i:i + n]
    return bytes(b)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    with tempfile.TemporaryDirectory() as d:
        tmp = Path(d)
        scene_path = _textured_scene(tmp)
        (tmp / "m.obj").write_text(OBJ)
        targets = [p for p in tmp.iterdir() if p.suffix in (".json", ".gltf", ".obj", ".bin")]
        originals = {p: p.read_bytes() for p in targets}
        pt.HostScene.load(scene_path)
        loaded = errors = 0
        for k in range(rounds):
            victim = rng.choice(targets)
            victim.write_bytes(damage(originals[victim], rng))
            entry = scene_path if victim.suffix != ".obj" else str(victim)
            try:
                pt.HostScene.load(entry).close()
                loaded += 1
            except pt.PtError:
                errors += 1
            victim.write_bytes(originals[victim])
            if (k + 1) % 500 == 0:
                print(f"{k + 1} rounds: {loaded} loaded, {errors} reported an error", flush=True)
        print(f"done: {rounds} damaged files, {loaded} loaded, {errors} reported an error, no crash")


if __name__ == "__main__":
    main()
