#!/usr/bin/env python3
"""Random damage to the arrays of a scene description handed to pt_scene_upload: every call has to end in PT_OK
or in an error status, never in a crash or an out-of-bounds read of the caller's arrays.  Only uploads are made
(no ray is traced through a damaged scene).  Every array of the description ends right before an inaccessible
page, so a read past its end stops the process on the spot (the address sanitizer cannot be used next to the
HIP runtime on this machine).

    python tools/fuzz_upload.py [scene] [rounds] [seed]
"""
import ctypes as C
import mmap
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_path_tracer_amd as pt  # noqa: E402


PAGE = mmap.PAGESIZE
_libc = C.CDLL(None, use_errno=True)
_maps = []


def guarded(a: np.ndarray) -> np.ndarray:
    """A copy of `a` whose last byte is the last accessible byte before a PROT_NONE page."""
    nbytes = max(a.nbytes, 1)
    span = (nbytes + PAGE - 1) // PAGE * PAGE
    m = mmap.mmap(-1, span + PAGE)
    base = C.addressof(C.c_char.from_buffer(m))
    if _libc.mprotect(C.c_void_p(base + span), C.c_size_t(PAGE), 0) != 0:
        raise OSError(C.get_errno(), "mprotect")
    start = span - a.nbytes
    out = np.frombuffer(m, dtype=a.dtype, count=a.size, offset=start).reshape(a.shape)
    out[...] = a
    _maps.append(m)
    return out


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    hs = pt.SceneSpec.load(os.path.join(ROOT, "scenes", scene + ".npz")).build(width=64, height=64, max_depth=4)
    ctx = pt.Context(0)
    d = hs.desc()
    ctx.upload(d)
    arrays = {k: guarded(v) for k, v in hs.arrays().items() if isinstance(v, np.ndarray)}
    d = pt.abi.SceneDesc.from_buffer_copy(d)
    d.positions = arrays["positions"].ctypes.data_as(C.POINTER(C.c_float))
    d.normals = arrays["normals"].ctypes.data_as(C.POINTER(C.c_float))
    d.uvs = arrays["uvs"].ctypes.data_as(C.POINTER(C.c_float))
    d.indices = arrays["indices"].ctypes.data_as(C.POINTER(C.c_uint32))
    d.tri_indices = arrays["tri_indices"].ctypes.data_as(C.POINTER(C.c_uint32))
    for name in ("geometries", "tlas_nodes", "blases", "blas_nodes", "materials", "lights"):
        setattr(d, name, arrays[name].ctypes.data)
    ctx.upload(d)
    fields = {  # description member -> (array, setter)
        "indices": lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32)),
        "tri_indices": lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32)),
        "geometries": lambda a: a.ctypes.data,
        "tlas_nodes": lambda a: a.ctypes.data,
        "blases": lambda a: a.ctypes.data,
        "blas_nodes": lambda a: a.ctypes.data,
        "materials": lambda a: a.ctypes.data,
        "lights": lambda a: a.ctypes.data,
    }
    special = np.array([0, 1, 2, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFFE], dtype=np.uint32)
    ok = err = 0
    keep = []
    for k in range(rounds):
        name = list(fields)[rng.integers(len(fields))]
        if arrays[name].size == 0:
            continue
        del _maps[len(arrays):]  # the guarded copies of earlier rounds
        a = guarded(arrays[name])
        words = a.view(np.uint32).reshape(-1)  # every record type here is made of 32-bit words
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(words.size))
            choice = int(rng.integers(3))
            words[i] = special[rng.integers(special.size)] if choice == 0 else rng.integers(0, 1 << 32, dtype=np.uint64).astype(np.uint32) if choice == 1 else np.uint32((int(words[i]) + int(rng.integers(-3, 4))) & 0xFFFFFFFF)
        bad = pt.abi.SceneDesc.from_buffer_copy(d)
        setattr(bad, name, fields[name](a))
        keep = [a]
        if rng.integers(8) == 0:  # now and then a count that names fewer records than the other arrays refer to
            cname = ["vertex_count", "index_count", "geometry_count", "tlas_node_count", "blas_count", "blas_node_count", "tri_index_count", "material_count", "light_count"][rng.integers(9)]
            cur = getattr(bad, cname)
            if cur > 1:
                setattr(bad, cname, int(rng.integers(1, cur)))
        try:
            ctx.upload(bad)
            ok += 1
        except pt.PtError:
            err += 1
        if (k + 1) % 500 == 0:
            print(f"{k + 1} rounds: {ok} accepted, {err} refused", flush=True)
    ctx.upload(d)
    print(f"done: {rounds} damaged descriptions of {scene}, {ok} accepted, {err} refused, no crash")
    del keep


if __name__ == "__main__":
    main()
