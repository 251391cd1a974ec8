"""ctypes mirror of include/ptamd.h and include/ptamd_host.h.

Only type definitions and library loading live here; nothing in this module computes.
The product library is ``libptamd.so`` next to this file; importing the package fails
loudly when it has not been built (``python -c "import __graft_entry__ as g; g.build()"``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptamd.so")

PT_OK, PT_ERR_INVALID, PT_ERR_DEVICE, PT_ERR_STATE, PT_ERR_LIMIT = 0, 1, 2, 3, 4
PT_MAT_LAMBERTIAN, PT_MAT_DIELECTRIC, PT_MAT_MIRROR, PT_MAT_METALLIC_ROUGHNESS, PT_MAT_EMISSIVE = range(5)
PT_TEX_CONSTANT, PT_TEX_IMAGE, PT_TEX_CHECKERBOARD = range(3)
PT_MAX_TRACKED_BOUNCES = 16

# ---- numpy dtypes for array-of-struct buffers (byte-identical to the C structs)
BVH_NODE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("first_primitive", "<u4"), ("primitive_count", "<u4")])
TLAS_NODE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("instance", "<u4"), ("left_right", "<u4")])
BLAS = np.dtype([("first_node", "<u4"), ("node_count", "<u4"), ("first_index", "<u4"), ("index_count", "<u4")])
GEOMETRY = np.dtype([
    ("material_id", "<i4"), ("first_index", "<u4"), ("index_count", "<u4"), ("vertex_count", "<u4"),
    ("transform", "<f4", 16), ("inv_transform", "<f4", 16), ("inv_transpose", "<f4", 16),
    ("blas_index", "<i4"), ("total_area", "<f4"), ("triangle_sampler_index", "<i4"), ("area_light_id", "<i4")])
MATERIAL = np.dtype([("type", "u1"), ("_pad", "u1", 3), ("tex0", "<i4"), ("w1", "<u4"), ("w2", "<i4")])
AREA_LIGHT = np.dtype([("power", "<f4"), ("geometry_instance_index", "<i4")])
RAY = np.dtype([("origin", "<f4", 3), ("tmin", "<f4"), ("direction", "<f4", 3), ("tmax", "<f4")])
HIT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<i4"), ("inst", "<i4"), ("material_id", "<i4"),
                ("_pad", "<u4", 2)])
INTERSECTION = np.dtype([("time", "<f4"), ("pos", "<f4", 3), ("normal", "<f4", 3), ("uv", "<f4", 2),
                         ("material_id", "<i4"), ("geometry_id", "<i4")])
assert BVH_NODE.itemsize == 32 and TLAS_NODE.itemsize == 32 and BLAS.itemsize == 16
assert GEOMETRY.itemsize == 224 and MATERIAL.itemsize == 16 and RAY.itemsize == 32 and HIT.itemsize == 32
assert INTERSECTION.itemsize == 44


class Material(C.Structure):
    _fields_ = [("type", C.c_uint8), ("_pad", C.c_uint8 * 3), ("tex0", C.c_int32), ("w1", C.c_uint32), ("w2", C.c_int32)]


class _Tex1Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("scale", C.c_float * 2), ("offset", C.c_float * 2),
                ("image_id", C.c_int32)]


class _Tex1Checker(C.Structure):
    _fields_ = [("color0", C.c_float), ("color1", C.c_float), ("scale", C.c_float * 2), ("offset", C.c_float * 2)]


class _Tex1U(C.Union):
    _fields_ = [("constant", C.c_float), ("image", _Tex1Image), ("checkerboard", _Tex1Checker)]


class Texture1(C.Structure):
    _fields_ = [("type", C.c_uint8), ("_pad", C.c_uint8 * 3), ("u", _Tex1U)]


class _Tex3Checker(C.Structure):
    _fields_ = [("color0", C.c_float * 3), ("color1", C.c_float * 3), ("scale", C.c_float * 2), ("offset", C.c_float * 2)]


class _Tex3U(C.Union):
    _fields_ = [("constant", C.c_float * 3), ("image", _Tex1Image), ("checkerboard", _Tex3Checker)]


class Texture3(C.Structure):
    _fields_ = [("type", C.c_uint8), ("_pad", C.c_uint8 * 3), ("u", _Tex3U)]


class Image(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32),
                ("srgb", C.c_int32)]


class AliasTable(C.Structure):
    _fields_ = [("alias_table", C.POINTER(C.c_uint32)), ("probability_table", C.POINTER(C.c_float)),
                ("size", C.c_uint32), ("_pad", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("resolution", C.c_int32 * 2), ("position", C.c_float * 3), ("front", C.c_float * 3),
                ("up", C.c_float * 3), ("right", C.c_float * 3), ("near_plane", C.c_float), ("far_plane", C.c_float),
                ("focal_length", C.c_float), ("fovy", C.c_float), ("pixel_length", C.c_float * 2)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("max_depth", C.c_uint32), ("vertex_count", C.c_uint32),
        ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("uvs", C.POINTER(C.c_float)),
        ("indices", C.POINTER(C.c_uint32)), ("index_count", C.c_uint32), ("geometry_count", C.c_uint32),
        ("geometries", C.c_void_p), ("triangle_samplers", C.POINTER(AliasTable)),
        ("triangle_sampler_count", C.c_uint32), ("tlas_node_count", C.c_uint32),
        ("tlas_nodes", C.c_void_p), ("blases", C.c_void_p), ("blas_count", C.c_uint32), ("blas_node_count", C.c_uint32),
        ("blas_nodes", C.c_void_p), ("tri_indices", C.POINTER(C.c_uint32)), ("tri_index_count", C.c_uint32),
        ("material_count", C.c_uint32), ("materials", C.c_void_p), ("lights", C.c_void_p), ("light_count", C.c_uint32),
        ("total_light_power", C.c_float), ("light_sampler", C.POINTER(AliasTable)),
        ("textures1", C.POINTER(Texture1)), ("textures3", C.POINTER(Texture3)),
        ("texture1_count", C.c_uint32), ("texture3_count", C.c_uint32),
        ("images1", C.POINTER(Image)), ("images3", C.POINTER(Image)),
        ("image1_count", C.c_uint32), ("image3_count", C.c_uint32), ("camera", Camera)]


class Stats(C.Structure):
    _fields_ = [
        ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64),
        ("bounces_run", C.c_uint32), ("stack_spills", C.c_uint32), ("stack_drops", C.c_uint32), ("nan_pixels", C.c_uint32),
        ("queue_sizes", C.c_uint32 * PT_MAX_TRACKED_BOUNCES),
        ("ms_closest", C.c_double), ("ms_any", C.c_double), ("ms_generate", C.c_double), ("ms_shade", C.c_double),
        ("ms_film", C.c_double),
        ("n_closest", C.c_uint32), ("n_any", C.c_uint32), ("n_generate", C.c_uint32), ("n_shade", C.c_uint32),
        ("n_film", C.c_uint32), ("_reserved", C.c_uint32), ("shaded_paths", C.c_uint64)]


class IntegratorState(C.Structure):
    """pt_integrator_state (IntegratorState, kernel/device_types.h:26-34)."""
    _fields_ = [("visualize_bvh", C.c_int32), ("bvh_visualize_type", C.c_int32), ("bvh_visualize_depth", C.c_int32),
                ("debug_mode", C.c_int32)]


class OffscreenOptions(C.Structure):
    """pth_offscreen_options."""
    _fields_ = [("iterations", C.c_uint32), ("batch", C.c_uint32), ("save_every", C.c_uint32), ("resume", C.c_int32),
                ("output_prefix", C.c_char_p), ("checkpoint_path", C.c_char_p), ("window", C.c_int32 * 4)]


class OffscreenResult(C.Structure):
    """pth_offscreen_result."""
    _fields_ = [("first_iteration", C.c_uint32), ("iterations_done", C.c_uint32), ("images_written", C.c_uint32),
                ("seconds", C.c_double)]


# DebugVisualizeMode / BVH_VisualizeMode, kernel/device_types.h:10-24
(PT_DEBUG_NONE, PT_DEBUG_GEOMETRIC_NORMALS, PT_DEBUG_ALBEDO, PT_DEBUG_UV, PT_DEBUG_DEPTH, PT_DEBUG_METALLIC,
 PT_DEBUG_ROUGHNESS) = range(7)
PT_BVH_BOUNDS_HIT, PT_BVH_TLAS, PT_BVH_TRIANGLES_HIT = range(3)

assert C.sizeof(Material) == 16 and C.sizeof(Texture1) == 32 and C.sizeof(Texture3) == 44

# Every symbol the two public headers declare (tests check that the library exports all of them).
PT_SYMBOLS = [
    "pt_ctx_create", "pt_ctx_destroy", "pt_last_error", "pt_version", "pt_set_stream", "pt_synchronize",
    "pt_set_profiling", "pt_get_stats", "pt_reset_stats", "pt_scene_upload", "pt_camera_update", "pt_scene_free",
    "pt_integrator_init", "pt_integrator_destroy", "pt_integrator_resize", "pt_integrator_reset", "pt_render",
    "pt_render_async", "pt_render_batch", "pt_set_render_window", "pt_film_download", "pt_film_copy_to_device", "pt_film_copy_from_device", "pt_film_upload", "pt_display_download", "pt_tonemap",
    "pt_trace_closest", "pt_trace_any", "pt_resolve_hits", "pt_trace_closest_counted", "pt_trace_any_counted", "pt_trace_closest_static", "pt_trace_closest_host",
    "pt_trace_any_host", "pt_resolve_hits_host", "pt_generate_incoherent_rays", "pt_capture_bounce_rays", "pt_capture_shadow_rays",
    "pt_set_trace_mode", "pt_set_texture_filter", "pt_film_reduce", "pt_rccl_unique_id", "pt_rccl_comm_create", "pt_rccl_comm_destroy", "pt_debug_render", "pt_bvh_build", "pt_bvh_sizes", "pt_bvh_get", "pt_bvh_timing", "pt_bvh_free"]
PTH_SYMBOLS = [
    "pth_last_error", "pth_transform_matrix", "pth_matrix_derive", "pth_matrix_mul", "pth_gltf_local_matrix",
    "pth_alias_build", "pth_camera_make", "pth_bvh_build", "pth_bvh_sizes", "pth_bvh_get", "pth_bvh_free",
    "pth_scene_create", "pth_scene_load", "pth_scene_destroy", "pth_scene_set_integrator", "pth_scene_set_camera", "pth_scene_add_mesh",
    "pth_scene_add_material", "pth_scene_add_texture1", "pth_scene_add_texture3", "pth_scene_add_image",
    "pth_scene_new_blas_index", "pth_scene_add_instance", "pth_scene_finalize", "pth_scene_finalize_device", "pth_scene_get_desc",
    "pth_scene_update_device", "pth_integrator_create", "pth_integrator_destroy", "pth_integrator_init",
    "pth_integrator_resize", "pth_integrator_reset", "pth_integrator_render", "pth_debug_render", "pth_render_offscreen", "pth_checkpoint_read", "pth_save_png", "pth_save_hdr",
    "pth_decode_image", "pth_decode_image_error"]

# pth_image_decoder (include/ptamd_host.h)
IMAGE_DECODER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                            C.POINTER(C.c_int), C.POINTER(C.c_uint8), C.c_size_t)

_lib = None


def load_library() -> C.CDLL:
    """Loads libptamd.so (once). Raises ImportError with the build hint when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PTAMD_LIB", LIB_PATH)  # diagnostic builds (tools/) live beside the product library
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built. "
            "Run `python -c \"import __graft_entry__ as g; g.build()\"` (needs hipcc); there is no CPU fallback.")
    # One process must hold ONE HIP/HSA runtime: PyTorch-ROCm bundles its own libamdhip64.so, and
    # whichever runtime initialises second finds no device.  Importing torch first makes the
    # dynamic loader satisfy libptamd.so's libamdhip64 dependency with torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float
    fp = C.POINTER(C.c_float)
    up = C.POINTER(C.c_uint32)
    sz = C.c_size_t
    sig = {
        "pt_ctx_create": (i32, [i32, C.POINTER(vp)]),
        "pt_ctx_destroy": (None, [vp]),
        "pt_last_error": (C.c_char_p, [vp]),
        "pt_version": (C.c_char_p, []),
        "pt_set_stream": (i32, [vp, vp]),
        "pt_synchronize": (i32, [vp]),
        "pt_set_profiling": (i32, [vp, i32]),
        "pt_get_stats": (i32, [vp, C.POINTER(Stats)]),
        "pt_reset_stats": (i32, [vp]),
        "pt_scene_upload": (i32, [vp, C.POINTER(SceneDesc)]),
        "pt_camera_update": (i32, [vp, C.POINTER(Camera)]),
        "pt_scene_free": (i32, [vp]),
        "pt_integrator_init": (i32, [vp, i32, i32]),
        "pt_integrator_destroy": (i32, [vp]),
        "pt_integrator_resize": (i32, [vp, i32, i32]),
        "pt_integrator_reset": (i32, [vp, i32, i32]),
        "pt_render": (i32, [vp, i32, i32, i32, vp]),
        "pt_render_async": (i32, [vp, i32, i32, i32, vp]),
        "pt_render_batch": (i32, [vp, i32, i32, i32, i32, vp]),
        "pt_set_render_window": (i32, [vp, i32, i32, i32, i32]),
        "pt_film_download": (i32, [vp, vp]),
        "pt_film_copy_to_device": (i32, [vp, vp]),
        "pt_film_copy_from_device": (i32, [vp, vp]),
        "pt_tonemap": (i32, [vp, vp, i32, i32, i32, vp]),
        "pt_film_upload": (i32, [vp, vp]),
        "pt_display_download": (i32, [vp, i32, vp]),
        "pt_trace_closest": (i32, [vp, vp, sz, vp]),
        "pt_trace_any": (i32, [vp, vp, sz, vp]),
        "pt_resolve_hits": (i32, [vp, vp, vp, sz, vp]),
        "pt_trace_closest_counted": (i32, [vp, vp, sz, vp, vp]),
        "pt_trace_closest_static": (i32, [vp, vp, sz, vp]),
        "pt_trace_any_counted": (i32, [vp, vp, sz, vp, vp]),
        "pt_trace_closest_host": (i32, [vp, vp, sz, vp]),
        "pt_trace_any_host": (i32, [vp, vp, sz, vp]),
        "pt_resolve_hits_host": (i32, [vp, vp, vp, sz, vp]),
        "pt_generate_incoherent_rays": (i32, [vp, vp, sz, u64]),
        "pt_capture_bounce_rays": (i32, [vp, i32, i32, i32, i32, vp, sz, C.POINTER(sz)]),
        "pt_capture_shadow_rays": (i32, [vp, i32, i32, i32, i32, vp, sz, C.POINTER(sz)]),
        "pt_set_trace_mode": (i32, [vp, i32]),
        "pt_set_texture_filter": (i32, [vp, i32]),
        "pt_film_reduce": (i32, [vp, vp, i32]),
        "pt_rccl_unique_id": (i32, [vp, vp]),
        "pt_rccl_comm_create": (i32, [vp, i32, vp, i32, C.POINTER(vp)]),
        "pt_rccl_comm_destroy": (i32, [vp, vp]),
        "pt_bvh_build": (i32, [vp, fp, u32, up, u32, vp, u32, C.POINTER(vp)]),
        "pt_bvh_sizes": (i32, [vp, up, up, up, up]),
        "pt_bvh_get": (i32, [vp, vp, vp, vp, vp]),
        "pt_bvh_timing": (i32, [vp, fp, fp, up]),
        "pt_bvh_free": (None, [vp]),
        "pt_debug_render": (i32, [vp, i32, i32, C.POINTER(IntegratorState), vp, vp]),
        "pth_last_error": (C.c_char_p, []),
        "pth_transform_matrix": (i32, [fp, fp, fp, fp]),
        "pth_matrix_derive": (i32, [fp, fp, fp]),
        "pth_matrix_mul": (i32, [fp, fp, fp]),
        "pth_gltf_local_matrix": (i32, [fp, fp, fp, fp, fp]),
        "pth_alias_build": (i32, [fp, u32, fp, up]),
        "pth_camera_make": (i32, [fp, fp, fp, f32, i32, i32, C.POINTER(Camera)]),
        "pth_bvh_build": (i32, [fp, up, vp, u32, C.POINTER(vp)]),
        "pth_bvh_sizes": (i32, [vp, up, up, up, up]),
        "pth_bvh_get": (i32, [vp, vp, vp, vp, vp]),
        "pth_bvh_free": (None, [vp]),
        "pth_scene_create": (i32, [C.POINTER(vp)]),
        "pth_scene_load": (i32, [C.c_char_p, IMAGE_DECODER, vp, i32, i32, i32, C.POINTER(vp)]),
        "pth_decode_image": (i32, [vp, C.POINTER(C.c_uint8), C.c_size_t, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_uint8), C.c_size_t]),
        "pth_decode_image_error": (C.c_char_p, []),
        "pth_scene_destroy": (None, [vp]),
        "pth_scene_set_integrator": (i32, [vp, i32, i32]),
        "pth_scene_set_camera": (i32, [vp, fp, fp, fp, f32, i32, i32]),
        "pth_scene_add_mesh": (i32, [vp, fp, fp, fp, u32, up, u32, up]),
        "pth_scene_add_material": (i32, [vp, C.POINTER(Material), C.POINTER(i32)]),
        "pth_scene_add_texture1": (i32, [vp, C.POINTER(Texture1), C.POINTER(i32)]),
        "pth_scene_add_texture3": (i32, [vp, C.POINTER(Texture3), C.POINTER(i32)]),
        "pth_scene_add_image": (i32, [vp, fp, i32, i32, i32, i32, C.POINTER(i32)]),
        "pth_scene_new_blas_index": (i32, [vp, C.POINTER(i32)]),
        "pth_scene_add_instance": (i32, [vp, u32, u32, u32, i32, fp, i32, C.POINTER(i32)]),
        "pth_scene_finalize": (i32, [vp]),
        "pth_scene_finalize_device": (i32, [vp, vp]),
        "pth_scene_get_desc": (i32, [vp, C.POINTER(SceneDesc)]),
        "pth_scene_update_device": (i32, [vp, vp]),
        "pth_integrator_create": (i32, [vp, C.POINTER(vp)]),
        "pth_integrator_destroy": (None, [vp]),
        "pth_integrator_init": (i32, [vp, i32, i32]),
        "pth_integrator_resize": (i32, [vp, i32, i32]),
        "pth_integrator_reset": (i32, [vp, i32, i32]),
        "pth_integrator_render": (i32, [vp, i32, i32, vp, vp, i32]),
        "pth_debug_render": (i32, [vp, i32, i32, vp, C.POINTER(IntegratorState), vp]),
        "pth_render_offscreen": (i32, [vp, vp, C.POINTER(OffscreenOptions), C.POINTER(OffscreenResult)]),
        "pth_checkpoint_read": (i32, [C.c_char_p, C.POINTER(i32), C.POINTER(i32), up, vp]),
        "pth_save_png": (i32, [C.c_char_p, vp, i32, i32]),
        "pth_save_hdr": (i32, [C.c_char_p, vp, i32, i32, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def uptr(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def vptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)
