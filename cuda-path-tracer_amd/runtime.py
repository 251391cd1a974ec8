"""Thin Python handles over the C ABI: Context (pt_ctx), HostScene (pth_scene), Integrator.

Mirrors the reference's object model for the hot path: a Scene that owns host arrays and
uploads them (Scene::UpdateDevice, scene.cpp:52-96) and an Integrator with
Init/Resize/Reset/Render (kernel/integrators/integrator.h:11-24).  Device buffers are passed
as integer addresses (e.g. ``tensor.data_ptr()``); nothing here allocates GPU memory itself.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os

import numpy as np

from . import abi


class PtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[ptamd status {code}] {message}")
        self.code = code


def _host_check(status: int):
    if status != abi.PT_OK:
        raise PtError(status, abi.load_library().pth_last_error().decode(errors="replace"))


def _f3(v):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))
    return a


# --------------------------------------------------------------------------- host helpers
def transform_matrix(translate, rotate_deg, scale) -> np.ndarray:
    """Math::GetTransformMatrix (math/transform.cpp:5-14) -> float32[16], column-major."""
    lib = abi.load_library()
    out = np.zeros(16, np.float32)
    _host_check(lib.pth_transform_matrix(abi.fptr(_f3(translate)), abi.fptr(_f3(rotate_deg)), abi.fptr(_f3(scale)), abi.fptr(out)))
    return out


def matrix_derive(transform: np.ndarray):
    """glm::inverse and glm::inverseTranspose of a column-major mat4 (scene.cpp:415-417)."""
    lib = abi.load_library()
    t = np.ascontiguousarray(transform, np.float32).reshape(16)
    inv, invt = np.zeros(16, np.float32), np.zeros(16, np.float32)
    _host_check(lib.pth_matrix_derive(abi.fptr(t), abi.fptr(inv), abi.fptr(invt)))
    return inv, invt


def matrix_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    lib = abi.load_library()
    out = np.zeros(16, np.float32)
    _host_check(lib.pth_matrix_mul(abi.fptr(np.ascontiguousarray(a, np.float32).reshape(16)),
                                   abi.fptr(np.ascontiguousarray(b, np.float32).reshape(16)), abi.fptr(out)))
    return out


def gltf_local_matrix(translation, rotation_xyzw, scale, matrix) -> np.ndarray:
    lib = abi.load_library()
    out = np.zeros(16, np.float32)
    q = np.ascontiguousarray(np.asarray(rotation_xyzw, np.float32).reshape(4))
    m = np.ascontiguousarray(np.asarray(matrix, np.float32).reshape(16))
    _host_check(lib.pth_gltf_local_matrix(abi.fptr(_f3(translation)), abi.fptr(q), abi.fptr(_f3(scale)), abi.fptr(m), abi.fptr(out)))
    return out


def alias_build(weights: np.ndarray):
    """DiscreteSampler1D::UpdateWeights (discrete_sampler.cpp:14-76) -> (probability, alias)."""
    lib = abi.load_library()
    w = np.ascontiguousarray(weights, np.float32)
    prob = np.zeros(len(w), np.float32)
    alias = np.zeros(len(w), np.uint32)
    _host_check(lib.pth_alias_build(abi.fptr(w), len(w), abi.fptr(prob), abi.uptr(alias)))
    return prob, alias


def camera_make(eye, look_at, up, fovy_deg: float, width: int, height: int) -> abi.Camera:
    """Camera::Update + UpdateDevice (camera.cpp:6-51)."""
    lib = abi.load_library()
    cam = abi.Camera()
    _host_check(lib.pth_camera_make(abi.fptr(_f3(eye)), abi.fptr(_f3(look_at)), abi.fptr(_f3(up)), float(fovy_deg), width, height, C.byref(cam)))
    return cam


def bvh_build(positions: np.ndarray, indices: np.ndarray, geometries: np.ndarray):
    """TLAS::Build (bvh.cpp:206-332) -> dict of reference-layout arrays."""
    lib = abi.load_library()
    pos = np.ascontiguousarray(positions, np.float32)
    idx = np.ascontiguousarray(indices, np.uint32)
    geo = np.ascontiguousarray(geometries, abi.GEOMETRY)
    h = C.c_void_p()
    _host_check(lib.pth_bvh_build(abi.fptr(pos.reshape(-1)), abi.uptr(idx), abi.vptr(geo), len(geo), C.byref(h)))
    try:
        n = [C.c_uint32() for _ in range(4)]
        _host_check(lib.pth_bvh_sizes(h, *[C.byref(x) for x in n]))
        out = {"tlas_nodes": np.zeros(n[0].value, abi.TLAS_NODE), "blases": np.zeros(n[1].value, abi.BLAS),
               "blas_nodes": np.zeros(n[2].value, abi.BVH_NODE), "tri_indices": np.zeros(n[3].value, np.uint32)}
        _host_check(lib.pth_bvh_get(h, abi.vptr(out["tlas_nodes"]), abi.vptr(out["blases"]), abi.vptr(out["blas_nodes"]),
                                    abi.vptr(out["tri_indices"])))
        return out
    finally:
        lib.pth_bvh_free(h)


def render_offscreen(ctx: "Context", scene: "HostScene", iterations: int = 0, batch: int = 0, save_every: int = 0,
                     output_prefix: str = None, checkpoint_path: str = None, resume: bool = False, window=(0, 0, 0, 0)) -> dict:
    """Application::Run's offscreen branch (application.cpp:58-76): progressive PNGs, film checkpoint, resume."""
    lib = abi.load_library()
    opt = abi.OffscreenOptions(iterations, batch, save_every, int(resume),
                               output_prefix.encode() if output_prefix else None,
                               checkpoint_path.encode() if checkpoint_path else None, (C.c_int32 * 4)(*window))
    res = abi.OffscreenResult()
    _host_check(lib.pth_render_offscreen(ctx.handle, scene.handle, C.byref(opt), C.byref(res)))
    return {"first_iteration": res.first_iteration, "iterations_done": res.iterations_done,
            "images_written": res.images_written, "seconds": res.seconds}


def checkpoint_read(path: str):
    """(film HxWx3 fp32, iterations) of a film checkpoint written by render_offscreen."""
    lib = abi.load_library()
    w, h, it = C.c_int(), C.c_int(), C.c_uint32()
    _host_check(lib.pth_checkpoint_read(path.encode(), C.byref(w), C.byref(h), C.byref(it), None))
    film = np.zeros((h.value, w.value, 3), np.float32)
    _host_check(lib.pth_checkpoint_read(path.encode(), C.byref(w), C.byref(h), C.byref(it), abi.vptr(film)))
    return film, it.value


def save_png(path: str, rgba: np.ndarray):
    lib = abi.load_library()
    a = np.ascontiguousarray(rgba, np.uint8)
    h, w = a.shape[0], a.shape[1]
    _host_check(lib.pth_save_png(path.encode(), abi.vptr(a), w, h))


def save_hdr(path: str, film: np.ndarray, iterations: int):
    lib = abi.load_library()
    a = np.ascontiguousarray(film, np.float32)
    h, w = a.shape[0], a.shape[1]
    _host_check(lib.pth_save_hdr(path.encode(), abi.vptr(a), w, h, int(iterations)))


# --------------------------------------------------------------------------- HostScene
def _pil_decoder():
    """A pth_image_decoder backed by PIL: the C++ scene loader has no image decoder of its own."""
    cache = {}

    def decode(_user, data, n, want_channels, w, h, c, pixels, capacity):
        try:
            import io

            from PIL import Image

            # The loader asks twice per image (size query, then fill) and frees the file's bytes in between
            # files, so the address of the bytes says nothing about their content: the entry that spans the two
            # calls is keyed on a digest of the bytes and dropped as soon as the pixels have been copied out.
            raw = C.string_at(data, n)
            key = (hashlib.blake2b(raw, digest_size=16).digest(), n, want_channels)
            if key not in cache:
                cache.clear()
                img = Image.open(io.BytesIO(raw))
                if want_channels in (3, 4):
                    img = img.convert("RGB" if want_channels == 3 else "RGBA")
                px = np.asarray(img, dtype=np.uint8)
                cache[key] = np.ascontiguousarray(px[:, :, None] if px.ndim == 2 else px)
            px = cache[key]
            w[0], h[0], c[0] = px.shape[1], px.shape[0], px.shape[2]
            if pixels:
                if capacity < px.size:
                    return 2
                C.memmove(pixels, px.ctypes.data, px.size)
                cache.clear()
            return 0
        except Exception:  # noqa: BLE001 - an exception must not cross the C boundary
            return 1

    return abi.IMAGE_DECODER(decode)


class HostScene:
    """pth_scene: the host-side Scene of the facade (scene.h:23-186), filled programmatically or
    read from a scene file by the C++ loader (HostScene.load)."""

    def __init__(self, _handle=None):
        self._lib = abi.load_library()
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            _host_check(self._lib.pth_scene_create(C.byref(self._h)))
        self.width = self.height = 0
        self.max_depth, self.max_iterations = 4, 1000

    @classmethod
    def load(cls, path: str, width: int = 0, height: int = 0, max_depth: int = 0, ctx: "Context" = None, decode_images: bool = True,
             decoder: str = "builtin") -> "HostScene":
        """Scene::LoadScene (scene.cpp:649-687) through the C++ front end: scene JSON, glTF, OBJ.  Images are decoded by
        the library's own JPEG / PNG decoder (`decoder="builtin"`, pth_decode_image) or by PIL through a callback
        (`decoder="pil"`, any format PIL reads).  The scene comes back finalised (on the GPU behind `ctx` when given)."""
        lib = abi.load_library()
        h = C.c_void_p()
        if not decode_images:
            decoder = abi.IMAGE_DECODER()
        elif decoder == "pil":
            decoder = _pil_decoder()
        elif decoder == "builtin":
            decoder = C.cast(lib.pth_decode_image, abi.IMAGE_DECODER)
        else:
            raise ValueError("decoder must be 'builtin' or 'pil'")
        _host_check(lib.pth_scene_load(os.fspath(path).encode(), decoder, None, int(width), int(height), int(max_depth), C.byref(h)))
        hs = cls(_handle=h)
        hs.finalize(ctx)
        d = hs.desc()
        hs.width, hs.height, hs.max_depth = d.camera.resolution[0], d.camera.resolution[1], d.max_depth
        return hs

    def close(self):
        if self._h:
            self._lib.pth_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def set_integrator(self, max_depth: int, max_iterations: int):
        _host_check(self._lib.pth_scene_set_integrator(self._h, int(max_depth), int(max_iterations)))
        self.max_depth, self.max_iterations = int(max_depth), int(max_iterations)

    def set_camera(self, eye, look_at, up, fovy_deg, width, height):
        _host_check(self._lib.pth_scene_set_camera(self._h, abi.fptr(_f3(eye)), abi.fptr(_f3(look_at)), abi.fptr(_f3(up)),
                                                   float(fovy_deg), int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def add_mesh(self, positions, normals, uvs, indices) -> int:
        p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
        i = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        assert len(p) == len(n) == len(t)
        first = C.c_uint32()
        _host_check(self._lib.pth_scene_add_mesh(self._h, abi.fptr(p.reshape(-1)), abi.fptr(n.reshape(-1)), abi.fptr(t.reshape(-1)),
                                                 len(p), abi.uptr(i), len(i), C.byref(first)))
        return first.value

    def add_material(self, mtype: int, tex0: int = 0, w1=0, w2: int = 0) -> int:
        m = abi.Material()
        m.type = mtype
        m.tex0 = int(tex0)
        m.w1 = int(np.float32(w1).view(np.uint32)) if isinstance(w1, (float, np.floating)) else int(np.int32(w1).view(np.uint32))
        m.w2 = int(w2)
        out = C.c_int()
        _host_check(self._lib.pth_scene_add_material(self._h, C.byref(m), C.byref(out)))
        return out.value

    def add_constant_texture3(self, rgb) -> int:
        t = abi.Texture3()
        t.type = abi.PT_TEX_CONSTANT
        for k in range(3):
            t.u.constant[k] = float(rgb[k])
        out = C.c_int()
        _host_check(self._lib.pth_scene_add_texture3(self._h, C.byref(t), C.byref(out)))
        return out.value

    def add_constant_texture1(self, value: float) -> int:
        t = abi.Texture1()
        t.type = abi.PT_TEX_CONSTANT
        t.u.constant = float(value)
        out = C.c_int()
        _host_check(self._lib.pth_scene_add_texture1(self._h, C.byref(t), C.byref(out)))
        return out.value

    def add_checker_texture3(self, c0, c1, scale=(1.0, 1.0), offset=(0.0, 0.0)) -> int:
        t = abi.Texture3()
        t.type = abi.PT_TEX_CHECKERBOARD
        for k in range(3):
            t.u.checkerboard.color0[k] = float(c0[k])
            t.u.checkerboard.color1[k] = float(c1[k])
        for k in range(2):
            t.u.checkerboard.scale[k] = float(scale[k])
            t.u.checkerboard.offset[k] = float(offset[k])
        out = C.c_int()
        _host_check(self._lib.pth_scene_add_texture3(self._h, C.byref(t), C.byref(out)))
        return out.value

    def add_image(self, texels: np.ndarray, srgb: bool = False) -> int:
        a = np.ascontiguousarray(texels, np.float32)
        if a.ndim == 2:
            a = a[:, :, None]
        h, w, ch = a.shape
        out = C.c_int()
        _host_check(self._lib.pth_scene_add_image(self._h, abi.fptr(a.reshape(-1)), w, h, ch, int(srgb), C.byref(out)))
        return out.value

    def add_image_texture(self, image_id: int, width: int, height: int, channels: int, scale=(1.0, 1.0), offset=(0.0, 0.0)) -> int:
        t = abi.Texture1() if channels == 1 else abi.Texture3()
        t.type = abi.PT_TEX_IMAGE
        t.u.image.width, t.u.image.height, t.u.image.image_id = int(width), int(height), int(image_id)
        for k in range(2):
            t.u.image.scale[k] = float(scale[k])
            t.u.image.offset[k] = float(offset[k])
        out = C.c_int()
        fn = self._lib.pth_scene_add_texture1 if channels == 1 else self._lib.pth_scene_add_texture3
        _host_check(fn(self._h, C.byref(t), C.byref(out)))
        return out.value

    def new_blas_index(self) -> int:
        out = C.c_int()
        _host_check(self._lib.pth_scene_new_blas_index(self._h, C.byref(out)))
        return out.value

    def add_instance(self, first_index: int, index_count: int, vertex_count: int, material_id: int, transform, blas_index: int) -> int:
        t = np.ascontiguousarray(transform, np.float32).reshape(16)
        out = C.c_int()
        _host_check(self._lib.pth_scene_add_instance(self._h, first_index, index_count, vertex_count, material_id, abi.fptr(t), blas_index, C.byref(out)))
        return out.value

    def finalize(self, ctx: "Context" = None):
        """Scene::Finalize: builds the acceleration structures (on the GPU behind `ctx` when given) and the light tables."""
        if ctx is None:
            _host_check(self._lib.pth_scene_finalize(self._h))
        else:
            _host_check(self._lib.pth_scene_finalize_device(self._h, ctx.handle))

    def desc(self) -> abi.SceneDesc:
        """Host-pointer scene description (valid while this object is alive and unchanged)."""
        d = abi.SceneDesc()
        _host_check(self._lib.pth_scene_get_desc(self._h, C.byref(d)))
        return d

    def arrays(self) -> dict:
        """Copies of the reference-layout arrays (for parity checks against the oracle)."""
        d = self.desc()

        def view(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype)
            addr = ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(addr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()

        return {
            "positions": view(d.positions, d.vertex_count * 3, np.float32).reshape(-1, 3),
            "normals": view(d.normals, d.vertex_count * 3, np.float32).reshape(-1, 3),
            "uvs": view(d.uvs, d.vertex_count * 2, np.float32).reshape(-1, 2),
            "indices": view(d.indices, d.index_count, np.uint32),
            "geometries": view(d.geometries, d.geometry_count, abi.GEOMETRY),
            "tlas_nodes": view(d.tlas_nodes, d.tlas_node_count, abi.TLAS_NODE),
            "blases": view(d.blases, d.blas_count, abi.BLAS),
            "blas_nodes": view(d.blas_nodes, d.blas_node_count, abi.BVH_NODE),
            "tri_indices": view(d.tri_indices, d.tri_index_count, np.uint32),
            "materials": view(d.materials, d.material_count, abi.MATERIAL),
            "lights": view(d.lights, d.light_count, abi.AREA_LIGHT),
            "total_light_power": float(d.total_light_power),
        }

    def update_device(self, ctx: "Context"):
        _host_check(self._lib.pth_scene_update_device(self._h, ctx.handle))


# --------------------------------------------------------------------------- Context
class Context:
    """pt_ctx: one GPU, one stream, the uploaded scene and the integrator state."""

    def __init__(self, device: int = 0):
        self._lib = abi.load_library()
        self._h = C.c_void_p()
        st = self._lib.pt_ctx_create(int(device), C.byref(self._h))
        if st != abi.PT_OK:
            raise PtError(st, self._lib.pt_last_error(None).decode())
        self.device = int(device)

    def close(self):
        if self._h:
            self._lib.pt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _check(self, status: int):
        if status != abi.PT_OK:
            raise PtError(status, self._lib.pt_last_error(self._h).decode())

    def upload(self, desc: abi.SceneDesc):
        self._check(self._lib.pt_scene_upload(self._h, C.byref(desc)))

    def camera_update(self, cam: abi.Camera):
        self._check(self._lib.pt_camera_update(self._h, C.byref(cam)))

    def scene_free(self):
        self._check(self._lib.pt_scene_free(self._h))

    def set_stream(self, stream_ptr: int):
        self._check(self._lib.pt_set_stream(self._h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._check(self._lib.pt_synchronize(self._h))

    def set_trace_mode(self, mode: int):
        """Two bits. 0 = reference-exact (default); bit 0: any-hit searches skip boxes beyond the ray's
        t_max; bit 1: 4-wide BVH nodes, rebuilt top level, searches bounded across instances."""
        self._check(self._lib.pt_set_trace_mode(self._h, int(mode)))

    def set_texture_filter(self, options: int):
        """Bit 0: bilinear weights with 8 fractional bits (as the CUDA texture unit); bit 1: sRGB decode of
        images tagged sRGB.  Applied by the next upload."""
        self._check(self._lib.pt_set_texture_filter(self._h, int(options)))

    def set_profiling(self, on: bool):
        self._check(self._lib.pt_set_profiling(self._h, int(on)))

    def stats(self) -> abi.Stats:
        s = abi.Stats()
        self._check(self._lib.pt_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(self._lib.pt_reset_stats(self._h))

    # integrator
    def integrator_init(self, w, h):
        self._check(self._lib.pt_integrator_init(self._h, w, h))

    def integrator_resize(self, w, h):
        self._check(self._lib.pt_integrator_resize(self._h, w, h))

    def integrator_reset(self, w, h):
        self._check(self._lib.pt_integrator_reset(self._h, w, h))

    def integrator_destroy(self):
        self._check(self._lib.pt_integrator_destroy(self._h))

    def render(self, w, h, iteration, rgba_device_ptr: int = 0, sync: bool = True):
        fn = self._lib.pt_render if sync else self._lib.pt_render_async
        self._check(fn(self._h, w, h, iteration, C.c_void_p(rgba_device_ptr)))

    def render_batch(self, w, h, iteration, samples, rgba_device_ptr: int = 0):
        """`samples` iterations in one wavefront pass (asynchronous); film identical to separate renders."""
        self._check(self._lib.pt_render_batch(self._h, w, h, iteration, samples, C.c_void_p(rgba_device_ptr)))

    def set_render_window(self, x0: int = 0, y0: int = 0, w: int = 0, h: int = 0):
        """Restrict the render calls to a pixel window of the frame (all zero: the whole frame)."""
        self._check(self._lib.pt_set_render_window(self._h, x0, y0, w, h))

    def film_download(self, w, h) -> np.ndarray:
        out = np.zeros((h, w, 3), np.float32)
        self._check(self._lib.pt_film_download(self._h, abi.vptr(out)))
        return out

    def film_copy_to_device(self, ptr: int):
        self._check(self._lib.pt_film_copy_to_device(self._h, C.c_void_p(ptr)))

    def film_copy_from_device(self, ptr: int):
        self._check(self._lib.pt_film_copy_from_device(self._h, C.c_void_p(ptr)))

    def film_upload(self, film: np.ndarray):
        f = np.ascontiguousarray(film, np.float32)
        self._check(self._lib.pt_film_upload(self._h, abi.vptr(f)))

    # multi-GPU film reduce over RCCL (one process per GPU)
    def rccl_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._check(self._lib.pt_rccl_unique_id(self._h, buf))
        return buf.raw

    def rccl_comm_create(self, nranks: int, unique_id: bytes, rank: int) -> int:
        comm = C.c_void_p()
        self._check(self._lib.pt_rccl_comm_create(self._h, nranks, C.create_string_buffer(unique_id, 128), rank, C.byref(comm)))
        return comm.value

    def rccl_comm_destroy(self, comm: int):
        self._check(self._lib.pt_rccl_comm_destroy(self._h, C.c_void_p(comm)))

    def film_reduce(self, comm: int, root: int = 0):
        """film(root) = sum of the ranks' films (ncclReduce on the context's stream, asynchronous)."""
        self._check(self._lib.pt_film_reduce(self._h, C.c_void_p(comm), root))

    def display_download(self, w, h, iterations) -> np.ndarray:
        out = np.zeros((h, w, 4), np.uint8)
        self._check(self._lib.pt_display_download(self._h, iterations, abi.vptr(out)))
        return out

    def tonemap(self, film_ptr: int, w, h, iterations, rgba_ptr: int):
        self._check(self._lib.pt_tonemap(self._h, C.c_void_p(film_ptr), w, h, iterations, C.c_void_p(rgba_ptr)))

    def debug_render(self, w, h, state: abi.IntegratorState, image_ptr: int = 0, rgba_ptr: int = 0):
        """DebugIntegrator::Render (debug.cu:198-239): BVH heat map or attribute view; asynchronous."""
        self._check(self._lib.pt_debug_render(self._h, w, h, C.byref(state), C.c_void_p(image_ptr), C.c_void_p(rgba_ptr)))

    def bvh_build(self, positions: np.ndarray, indices: np.ndarray, geometries: np.ndarray, timing: bool = False):
        """TLAS::Build (bvh.cpp:204-332) on the GPU -> dict of reference-layout host arrays, identical to bvh_build()."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1)
        idx = np.ascontiguousarray(indices, np.uint32)
        geo = np.ascontiguousarray(geometries, abi.GEOMETRY)
        h = C.c_void_p()
        self._check(self._lib.pt_bvh_build(self._h, abi.fptr(pos), len(pos) // 3, abi.uptr(idx), len(idx), abi.vptr(geo), len(geo), C.byref(h)))
        try:
            n = [C.c_uint32() for _ in range(4)]
            self._lib.pt_bvh_sizes(h, *[C.byref(x) for x in n])
            out = {"tlas_nodes": np.zeros(n[0].value, abi.TLAS_NODE), "blases": np.zeros(n[1].value, abi.BLAS),
                   "blas_nodes": np.zeros(n[2].value, abi.BVH_NODE), "tri_indices": np.zeros(n[3].value, np.uint32)}
            self._lib.pt_bvh_get(h, abi.vptr(out["tlas_nodes"]), abi.vptr(out["blases"]), abi.vptr(out["blas_nodes"]), abi.vptr(out["tri_indices"]))
            if timing:
                ms_blas, ms_tlas, levels = C.c_float(), C.c_float(), C.c_uint32()
                self._lib.pt_bvh_timing(h, C.byref(ms_blas), C.byref(ms_tlas), C.byref(levels))
                out["timing"] = {"ms_blas": ms_blas.value, "ms_tlas": ms_tlas.value, "levels": levels.value}
            return out
        finally:
            self._lib.pt_bvh_free(h)

    # batch queries on device pointers
    def trace_closest(self, rays_ptr: int, n: int, hits_ptr: int):
        self._check(self._lib.pt_trace_closest(self._h, C.c_void_p(rays_ptr), n, C.c_void_p(hits_ptr)))

    def trace_closest_static(self, rays_ptr: int, n: int, hits_ptr: int):
        """pt_trace_closest with its scratch in the library's static device array: allocates nothing."""
        self._check(self._lib.pt_trace_closest_static(self._h, C.c_void_p(rays_ptr), n, C.c_void_p(hits_ptr)))

    def trace_any(self, rays_ptr: int, n: int, occluded_ptr: int):
        self._check(self._lib.pt_trace_any(self._h, C.c_void_p(rays_ptr), n, C.c_void_p(occluded_ptr)))

    def trace_closest_counted(self, rays_ptr: int, n: int, hits_ptr: int, counters_ptr: int):
        self._check(self._lib.pt_trace_closest_counted(self._h, C.c_void_p(rays_ptr), n, C.c_void_p(hits_ptr), C.c_void_p(counters_ptr)))

    def trace_any_counted(self, rays_ptr: int, n: int, occluded_ptr: int, counters_ptr: int):
        self._check(self._lib.pt_trace_any_counted(self._h, C.c_void_p(rays_ptr), n, C.c_void_p(occluded_ptr), C.c_void_p(counters_ptr)))

    def resolve_hits(self, rays_ptr: int, hits_ptr: int, n: int, out_ptr: int):
        self._check(self._lib.pt_resolve_hits(self._h, C.c_void_p(rays_ptr), C.c_void_p(hits_ptr), n, C.c_void_p(out_ptr)))

    def generate_incoherent_rays(self, rays_ptr: int, n: int, first_index: int = 0):
        self._check(self._lib.pt_generate_incoherent_rays(self._h, C.c_void_p(rays_ptr), n, first_index))

    def capture_bounce_rays(self, w, h, iteration, bounce, rays_ptr: int, capacity: int) -> int:
        cnt = C.c_size_t()
        self._check(self._lib.pt_capture_bounce_rays(self._h, w, h, iteration, bounce, C.c_void_p(rays_ptr), capacity, C.byref(cnt)))
        return cnt.value

    def capture_shadow_rays(self, w, h, iteration, bounce, rays_ptr: int, capacity: int) -> int:
        cnt = C.c_size_t()
        self._check(self._lib.pt_capture_shadow_rays(self._h, w, h, iteration, bounce, C.c_void_p(rays_ptr), capacity, C.byref(cnt)))
        return cnt.value

    # host-array conveniences
    def trace_closest_host(self, rays: np.ndarray) -> np.ndarray:
        r = np.ascontiguousarray(rays, abi.RAY)
        hits = np.zeros(len(r), abi.HIT)
        self._check(self._lib.pt_trace_closest_host(self._h, abi.vptr(r), len(r), abi.vptr(hits)))
        return hits

    def trace_any_host(self, rays: np.ndarray) -> np.ndarray:
        r = np.ascontiguousarray(rays, abi.RAY)
        occ = np.zeros(len(r), np.uint8)
        self._check(self._lib.pt_trace_any_host(self._h, abi.vptr(r), len(r), abi.vptr(occ)))
        return occ

    def resolve_hits_host(self, rays: np.ndarray, hits: np.ndarray) -> np.ndarray:
        r = np.ascontiguousarray(rays, abi.RAY)
        h = np.ascontiguousarray(hits, abi.HIT)
        out = np.zeros(len(r), abi.INTERSECTION)
        self._check(self._lib.pt_resolve_hits_host(self._h, abi.vptr(r), abi.vptr(h), len(r), abi.vptr(out)))
        return out


class PathIntegrator:
    """PathIntegrator (kernel/integrators/path_trace.h:82-116) through the host facade."""

    def __init__(self, ctx: Context):
        self._lib = abi.load_library()
        self._ctx = ctx
        self._h = C.c_void_p()
        _host_check(self._lib.pth_integrator_create(ctx.handle, C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.pth_integrator_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init(self, w, h):
        _host_check(self._lib.pth_integrator_init(self._h, w, h))

    def resize(self, w, h):
        _host_check(self._lib.pth_integrator_resize(self._h, w, h))

    def reset(self, w, h):
        _host_check(self._lib.pth_integrator_reset(self._h, w, h))

    def render(self, w, h, scene: HostScene, rgba_device_ptr: int, iteration: int):
        _host_check(self._lib.pth_integrator_render(self._h, w, h, scene.handle, C.c_void_p(rgba_device_ptr), iteration))


class DebugIntegrator:
    """DebugIntegrator (kernel/integrators/debug.h:6-24) through the host facade: holds no device state."""

    def __init__(self, ctx: Context):
        self._lib = abi.load_library()
        self._ctx = ctx

    def init(self, w, h):
        pass

    def resize(self, w, h):
        pass

    def reset(self, w, h):
        pass

    def render(self, w, h, scene: HostScene, rgba_device_ptr: int, iteration: int, state: abi.IntegratorState):
        _host_check(self._lib.pth_debug_render(self._ctx.handle, w, h, scene.handle, C.byref(state), C.c_void_p(rgba_device_ptr)))
