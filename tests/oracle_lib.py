"""ctypes bindings of the CPU oracle (oracle/liboracle.so).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

import sys  # noqa: E402

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from cuda_path_tracer_amd import abi  # noqa: E402  (struct layouts only)

_lib = None


def build(force: bool = False):
    src = os.path.join(ORACLE_DIR, "oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(os.environ.get("ORACLE_LIB", LIB_PATH))  # bench.py's CPU-baseline leg names a build for the local CPU
    vp, i32, u32, u64, f32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    dp = C.POINTER(abi.SceneDesc)
    sig = {
        "ora_intersect_aabb": (i32, [fp, fp, fp, fp, f32, fp]),
        "ora_intersect_triangle": (i32, [fp, fp, fp, fp, fp, f32, fp, fp]),
        "ora_tlas_build": (vp, [fp, up, vp, u32]),
        "ora_tlas_sizes": (None, [vp, up, up, up, up]),
        "ora_tlas_get": (None, [vp, vp, vp, vp, vp]),
        "ora_tlas_free": (None, [vp]),
        "ora_alias_build": (i32, [fp, u32, fp, up]),
        "ora_light_data": (vp, [fp, up, vp, u32, vp]),
        "ora_light_count": (u32, [vp]),
        "ora_light_total_power": (f32, [vp]),
        "ora_light_get": (None, [vp, vp, fp, up]),
        "ora_light_tri_count": (u32, [vp, u32]),
        "ora_light_tri_get": (None, [vp, u32, fp, up]),
        "ora_light_free": (None, [vp]),
        "ora_camera_update": (None, [fp, fp, fp, f32, i32, i32, C.POINTER(abi.Camera)]),
        "ora_transform_matrices": (None, [fp, fp, fp, fp, fp, fp]),
        "ora_matrix_derive": (None, [fp, fp, fp]),
        "ora_matrix_mul": (None, [fp, fp, fp]),
        "ora_trace_closest": (i32, [dp, vp, sz, vp, vp, C.POINTER(u64), i32]),
        "ora_trace_any": (i32, [dp, vp, sz, vp, i32]),
        "ora_trace_naive": (i32, [dp, vp, sz, vp, i32]),
        "ora_generate_incoherent_rays": (None, [dp, vp, sz, u64]),
        "ora_render": (i32, [dp, i32, i32, i32, vp, C.POINTER(u64), i32]),
        "ora_capture_bounce_rays": (sz, [dp, i32, i32, i32, i32, vp, sz]),
        "ora_tonemap": (None, [vp, i32, i32, i32, vp]),
        "ora_debug_render": (i32, [dp, i32, i32, C.POINTER(abi.IntegratorState), vp, vp, i32]),
        "ora_sample_texture3": (None, [dp, i32, f32, f32, fp]),
        "ora_sample_texture1": (f32, [dp, i32, f32, f32]),
        "ora_pcg_stream": (None, [i32, i32, i32, i32, fp]),
        "ora_pcg_hash": (u32, [u32]),
        "ora_set_texture_filter": (None, [i32]),
        "ora_stack_script": (u32, [up, i32, up]),
        "ora_camera_ray": (None, [C.POINTER(abi.Camera), i32, i32, f32, f32, fp, fp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _f(a):
    return np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1))


# ---- wrappers ---------------------------------------------------------------------------
def intersect_aabb(o, d, bmin, bmax, t_max=np.float32(3.4028235e38)):
    t = np.zeros(1, np.float32)
    h = lib().ora_intersect_aabb(abi.fptr(_f(o)), abi.fptr(_f(d)), abi.fptr(_f(bmin)), abi.fptr(_f(bmax)), float(t_max), abi.fptr(t))
    return bool(h), float(t[0])


def intersect_triangle(o, d, v0, v1, v2, t_max=np.float32(3.4028235e38)):
    uv = np.zeros(2, np.float32)
    t = np.zeros(1, np.float32)
    h = lib().ora_intersect_triangle(abi.fptr(_f(o)), abi.fptr(_f(d)), abi.fptr(_f(v0)), abi.fptr(_f(v1)), abi.fptr(_f(v2)), float(t_max),
                                     abi.fptr(uv), abi.fptr(t))
    return bool(h), float(t[0]), float(uv[0]), float(uv[1])


def tlas_build(positions, indices, geometries):
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1)
    idx = np.ascontiguousarray(indices, np.uint32)
    geo = np.ascontiguousarray(geometries, abi.GEOMETRY)
    h = lib().ora_tlas_build(abi.fptr(pos), abi.uptr(idx), abi.vptr(geo), len(geo))
    n = [C.c_uint32() for _ in range(4)]
    lib().ora_tlas_sizes(h, *[C.byref(x) for x in n])
    out = {"tlas_nodes": np.zeros(n[0].value, abi.TLAS_NODE), "blases": np.zeros(n[1].value, abi.BLAS),
           "blas_nodes": np.zeros(n[2].value, abi.BVH_NODE), "tri_indices": np.zeros(n[3].value, np.uint32)}
    lib().ora_tlas_get(h, abi.vptr(out["tlas_nodes"]), abi.vptr(out["blases"]), abi.vptr(out["blas_nodes"]), abi.vptr(out["tri_indices"]))
    lib().ora_tlas_free(h)
    return out


def alias_build(weights):
    w = np.ascontiguousarray(weights, np.float32)
    prob, alias = np.zeros(len(w), np.float32), np.zeros(len(w), np.uint32)
    st = lib().ora_alias_build(abi.fptr(w), len(w), abi.fptr(prob), abi.uptr(alias))
    return st, prob, alias


def light_data(positions, indices, geometries, materials):
    """Returns (geometries with light fields filled, lights, light_prob, light_alias, tri tables, total power)."""
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1)
    idx = np.ascontiguousarray(indices, np.uint32)
    geo = np.ascontiguousarray(geometries, abi.GEOMETRY).copy()
    mats = np.ascontiguousarray(materials, abi.MATERIAL)
    h = lib().ora_light_data(abi.fptr(pos), abi.uptr(idx), abi.vptr(geo), len(geo), abi.vptr(mats))
    assert h
    n = lib().ora_light_count(h)
    lights = np.zeros(n, abi.AREA_LIGHT)
    lp, la = np.zeros(n, np.float32), np.zeros(n, np.uint32)
    lib().ora_light_get(h, abi.vptr(lights), abi.fptr(lp), abi.uptr(la))
    tris = []
    for i in range(n):
        m = lib().ora_light_tri_count(h, i)
        p, a = np.zeros(m, np.float32), np.zeros(m, np.uint32)
        lib().ora_light_tri_get(h, i, abi.fptr(p), abi.uptr(a))
        tris.append((p, a))
    total = lib().ora_light_total_power(h)
    lib().ora_light_free(h)
    return geo, lights, lp, la, tris, total


def camera_update(eye, look_at, up, fovy, w, h) -> abi.Camera:
    cam = abi.Camera()
    lib().ora_camera_update(abi.fptr(_f(eye)), abi.fptr(_f(look_at)), abi.fptr(_f(up)), float(fovy), w, h, C.byref(cam))
    return cam


def transform_matrices(t, r, s):
    T, I, IT = (np.zeros(16, np.float32) for _ in range(3))
    lib().ora_transform_matrices(abi.fptr(_f(t)), abi.fptr(_f(r)), abi.fptr(_f(s)), abi.fptr(T), abi.fptr(I), abi.fptr(IT))
    return T, I, IT


def trace_closest(desc, rays, nthreads=1, counters=False, full=False):
    r = np.ascontiguousarray(rays, abi.RAY)
    hits = np.zeros(len(r), abi.HIT)
    fl = np.zeros(len(r), abi.INTERSECTION) if full else None
    ctr = (C.c_uint64 * 6)() if counters else None
    lib().ora_trace_closest(C.byref(desc), abi.vptr(r), len(r), abi.vptr(hits), abi.vptr(fl) if full else None, ctr, nthreads)
    out = [hits]
    if full:
        out.append(fl)
    if counters:
        out.append(np.array(list(ctr), np.uint64))
    return out[0] if len(out) == 1 else tuple(out)


def trace_any(desc, rays, nthreads=1):
    r = np.ascontiguousarray(rays, abi.RAY)
    occ = np.zeros(len(r), np.uint8)
    lib().ora_trace_any(C.byref(desc), abi.vptr(r), len(r), abi.vptr(occ), nthreads)
    return occ


def trace_naive(desc, rays, nthreads=1):
    r = np.ascontiguousarray(rays, abi.RAY)
    hits = np.zeros(len(r), abi.HIT)
    lib().ora_trace_naive(C.byref(desc), abi.vptr(r), len(r), abi.vptr(hits), nthreads)
    return hits


def generate_incoherent_rays(desc, n, first_index=0):
    rays = np.zeros(n, abi.RAY)
    lib().ora_generate_incoherent_rays(C.byref(desc), abi.vptr(rays), n, first_index)
    return rays


def render(desc, w, h, iteration, film=None, nthreads=1):
    if film is None:
        film = np.zeros((h, w, 3), np.float32)
    stats = (C.c_uint64 * 2)()
    lib().ora_render(C.byref(desc), w, h, iteration, abi.vptr(film), stats, nthreads)
    return film, (int(stats[0]), int(stats[1]))


def capture_bounce_rays(desc, w, h, iteration, bounce):
    rays = np.zeros(w * h, abi.RAY)
    n = lib().ora_capture_bounce_rays(C.byref(desc), w, h, iteration, bounce, abi.vptr(rays), len(rays))
    return rays[:n].copy()


def tonemap(film, iteration):
    h, w = film.shape[:2]
    out = np.zeros((h, w, 4), np.uint8)
    lib().ora_tonemap(abi.vptr(np.ascontiguousarray(film, np.float32)), w, h, iteration, abi.vptr(out))
    return out


def debug_render(desc, w, h, state, nthreads=1):
    """DebugIntegrator::Render: (float image HxWx3, rgba HxWx4); both untouched (zeros) when no view is selected."""
    image = np.zeros((h, w, 3), np.float32)
    rgba = np.zeros((h, w, 4), np.uint8)
    lib().ora_debug_render(C.byref(desc), w, h, C.byref(state), abi.vptr(image), abi.vptr(rgba), nthreads)
    return image, rgba


def sample_texture3(desc, texture_id, u, v):
    out = np.zeros(3, np.float32)
    lib().ora_sample_texture3(C.byref(desc), int(texture_id), float(u), float(v), abi.fptr(out))
    return out


def set_texture_filter(options: int):
    lib().ora_set_texture_filter(int(options))


def sincos_azimuth(phi):
    """The fixed-sequence sin / cos both sides use for sampled azimuths (SinCosAzimuth in oracle.cpp)."""
    p = np.ascontiguousarray(phi, np.float32)
    s, c = np.zeros_like(p), np.zeros_like(p)
    fn = lib().ora_sincos_azimuth
    fn.restype = None
    fn(abi.vptr(p), C.c_size_t(len(p)), abi.vptr(s), abi.vptr(c))
    return s, c


def pcg_hash(values):
    L = lib()
    return np.array([L.ora_pcg_hash(int(v)) for v in values], np.uint32)


def stack_script(script):
    """Runs push (v > 0) / pop (0) operations on the oracle's 64-entry stack -> (pops [k,2], final top)."""
    s = np.ascontiguousarray(script, np.uint32)
    out = np.zeros(2 * int((s == 0).sum()) + 2, np.uint32)
    top = lib().ora_stack_script(abi.uptr(s), len(s), abi.uptr(out))
    return out[: 2 * int((s == 0).sum())].reshape(-1, 2), int(top)


def pcg_stream(px, py, iteration, n):
    out = np.zeros(n, np.float32)
    lib().ora_pcg_stream(px, py, iteration, n, abi.fptr(out))
    return out
