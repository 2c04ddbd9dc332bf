"""ctypes binding of the C ABI declared in include/rm.h (librm_hip.so).

There is no fallback: if the HIP library is missing, loading raises.
"""
import ctypes
import os
from ctypes import CFUNCTYPE, POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librm_hip.so")

RM_STAT_COUNT = 8
RM_STATIC_MAX_PATCHES = 1024
RM_GROUP_ID_BYTES = 128
STAT_NAMES = ("depth_cap_hits", "step_cap_hits", "lightmap_oob", "texture_oob")


class Material(ctypes.Structure):
    _fields_ = [(n, c_float) for n in ("ambient", "diffuse", "specular", "shininess", "reflectivity", "emissive")]


_F3 = POINTER(c_float)
# rm_allgather_fn: (user, d_send, d_recv, bytes_per_rank, hip_stream) -> status
ALLGATHER_FN = CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p)
# rm_gather_fn: (user, d_send, d_recv, bytes_per_rank, root, hip_stream) -> status
GATHER_FN = CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p)
RM_EXCHANGE_ALL_GATHER, RM_EXCHANGE_PRESENTER = 0, 1
RM_PRECISION_STRICT, RM_PRECISION_FAST = 0, 1

# name -> (restype, argtypes); mirrors include/rm.h one to one
SIGNATURES = {
    "rm_last_error": (c_char_p, []),
    "rm_version": (c_int, []),
    "rm_device_count": (c_int, [POINTER(c_int)]),
    "rm_host_alloc": (c_void_p, [c_size_t]),
    "rm_host_free": (None, [c_void_p]),
    "rm_scene_create": (c_void_p, []),
    "rm_scene_destroy": (None, [c_void_p]),
    "rm_texture_create_rgb8": (c_int, [c_void_p, POINTER(c_uint8), c_int, c_int, c_float, c_float]),
    "rm_scene_add_sphere": (c_int, [c_void_p, _F3, c_float, POINTER(Material), _F3]),
    "rm_scene_add_xplane": (c_int, [c_void_p, c_float, c_float, POINTER(Material), _F3]),
    "rm_scene_add_yplane": (c_int, [c_void_p, c_float, c_float, POINTER(Material), _F3, c_int]),
    "rm_scene_add_zplane": (c_int, [c_void_p, c_float, c_float, POINTER(Material), _F3]),
    "rm_scene_add_plane": (c_int, [c_void_p, _F3, c_float, POINTER(Material), _F3]),
    "rm_scene_add_box": (c_int, [c_void_p, _F3, _F3, POINTER(Material), _F3]),
    "rm_scene_set_smooth_union": (c_int, [c_void_p, c_float]),
    "rm_scene_set_light": (c_int, [c_void_p, _F3, c_float]),
    "rm_scene_set_camera": (c_int, [c_void_p, _F3]),
    "rm_engine_create": (c_void_p, [c_void_p, c_int, c_int, c_int, c_int]),
    "rm_engine_destroy": (None, [c_void_p]),
    "rm_engine_set_stream": (c_int, [c_void_p, c_void_p]),
    "rm_engine_compute_lightmaps": (c_int, [c_void_p, c_int]),
    "rm_engine_compute_lightmaps_static": (c_int, [c_void_p, c_int]),
    "rm_engine_radiosity_begin": (c_int, [c_void_p]),
    "rm_engine_radiosity_form_factors": (c_int, [c_void_p, c_int, c_int]),
    "rm_engine_radiosity_bounce": (c_int, [c_void_p, c_int, c_int]),
    "rm_engine_radiosity_commit": (c_int, [c_void_p]),
    "rm_engine_radiosity_buffers": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int)]),
    "rm_engine_get_lightmap": (c_int, [c_void_p, c_int, POINTER(c_float)]),
    "rm_engine_set_lightmap": (c_int, [c_void_p, c_int, POINTER(c_float)]),
    "rm_engine_save_lightmaps": (c_int, [c_void_p, c_char_p]),
    "rm_engine_load_lightmaps": (c_int, [c_void_p, c_char_p]),
    "rm_engine_input": (c_int, [c_void_p, c_int32, c_int32, _F3, c_int32]),
    "rm_engine_get_camera": (c_int, [c_void_p, _F3]),
    "rm_engine_set_camera": (c_int, [c_void_p, _F3]),
    "rm_engine_render": (c_int, [c_void_p, POINTER(c_uint8), c_size_t, c_int32, c_int32, _F3, c_int32]),
    "rm_engine_render_async": (c_int, [c_void_p, c_void_p, c_size_t, c_int32, c_int32, _F3, c_int32]),
    "rm_engine_render_wait": (c_int, [c_void_p]),
    "rm_engine_render_rows_device": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int]),
    "rm_engine_render_strips_device": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int]),
    "rm_engine_pack_bgr_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "rm_engine_unpack_bgr_strips_device": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "rm_engine_fxaa_device": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "rm_engine_trace_primary": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_float)]),
    "rm_engine_probe_shade": (c_int, [c_void_p, c_int, POINTER(c_int32), POINTER(c_float), POINTER(c_float), POINTER(c_float)]),
    "rm_engine_probe_pack": (c_int, [c_void_p, c_int, POINTER(c_float), c_float, POINTER(c_uint32), POINTER(c_uint32)]),
    "rm_engine_set_ssaa": (c_int, [c_void_p, c_int]),
    "rm_engine_set_tetrahedral_normals": (c_int, [c_void_p, c_int]),
    "rm_engine_set_refraction": (c_int, [c_void_p, c_int]),
    "rm_engine_set_march_cap": (c_int, [c_void_p, c_int]),
    "rm_engine_set_precision": (c_int, [c_void_p, c_int]),
    "rm_engine_set_exact_occlusion": (c_int, [c_void_p, c_int]),
    "rm_engine_set_tile_splitting": (c_int, [c_void_p, c_int]),
    "rm_engine_split_tiles": (c_int, [c_void_p, POINTER(c_uint32)]),
    "rm_engine_get_stats": (c_int, [c_void_p, POINTER(c_uint64)]),
    "rm_engine_reset_stats": (c_int, [c_void_p]),
    "rm_engine_shader_clock": (c_int, [c_void_p, c_int, POINTER(c_double)]),
    "rm_engine_synchronize": (c_int, [c_void_p]),
    "rm_group_unique_id": (c_int, [POINTER(c_uint8)]),
    "rm_group_create": (c_void_p, [c_void_p, POINTER(c_uint8), c_int, c_int, c_int]),
    "rm_group_create_custom": (c_void_p, [c_void_p, c_int, c_int, c_int, ALLGATHER_FN, c_void_p]),
    "rm_group_destroy": (None, [c_void_p]),
    "rm_group_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "rm_group_compute_lightmaps": (c_int, [c_void_p, c_int]),
    "rm_group_frame": (c_int, [c_void_p, c_void_p, c_int, c_int32, c_int32, _F3, c_int32]),
    "rm_group_wait": (c_int, [c_void_p]),
    "rm_group_set_timing": (c_int, [c_void_p, c_int]),
    "rm_group_render_time": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int)]),
    "rm_group_exchange_time": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int)]),
    "rm_group_set_exchange": (c_int, [c_void_p, c_int]),
    "rm_group_next_presenter": (c_int, [c_void_p, POINTER(c_int)]),
    "rm_group_set_custom_gather": (c_int, [c_void_p, GATHER_FN, c_void_p]),
    "rm_group_communicators": (c_int, [c_void_p, POINTER(c_int)]),
    "rm_multi_create": (c_void_p, [c_void_p, c_int, c_int, c_int, POINTER(c_int), c_int]),
    "rm_multi_destroy": (None, [c_void_p]),
    "rm_multi_device_count": (c_int, [c_void_p]),
    "rm_multi_engine": (c_void_p, [c_void_p, c_int]),
    "rm_multi_compute_lightmaps": (c_int, [c_void_p, c_int]),
    "rm_multi_render": (c_int, [c_void_p, POINTER(c_uint8), c_size_t, c_int32, c_int32, _F3, c_int32]),
}

_lib = None


def load():
    """Load librm_hip.so and declare every entry point.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C raymarching_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    # PyTorch wheels bundle a HIP runtime under the same soname as the system one, and a process can
    # only work with one: whichever is loaded first serves both.  Loading torch's first keeps
    # torch.cuda usable next to this library (the reverse order leaves torch without GPUs).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


class RMError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().rm_last_error()
        raise RMError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def f3(v):
    return (c_float * 3)(float(v[0]), float(v[1]), float(v[2]))
