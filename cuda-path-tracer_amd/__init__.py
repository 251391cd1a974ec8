"""MI355X-native ray-scene hot path of the CUDA path tracer: ctypes bindings over libptamd.so
(HIP kernels + C ABI + host C++ facade) and the scene ingestion helpers."""
from . import abi, sharding  # noqa: F401
from .runtime import (Context, DebugIntegrator, HostScene, PathIntegrator, PtError, alias_build, bvh_build, camera_make,  # noqa: F401
                      checkpoint_read, gltf_local_matrix, matrix_derive, matrix_mul, render_offscreen, save_hdr, save_png,
                      transform_matrix)
from .scene_io import SceneSpec, load_scene_json, unit_cube, uv_sphere  # noqa: F401
