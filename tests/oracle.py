"""ctypes wrapper of the CPU oracle (oracle/librm_oracle.so).  Test infrastructure:
imported only by tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_uint8, c_uint64, c_void_p

import numpy as np

from raymarching_amd.scene import Scene, build_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

COUNTER_NAMES = (
    "march_calls", "march_steps", "march_sphere_sdf", "march_plane_sdf", "march_gplane_sdf",
    "shadow_calls", "shadow_steps", "shadow_sphere_sdf", "shadow_plane_sdf", "shadow_gplane_sdf",
    "normals_sphere", "normals_plane", "normals_gplane", "shade_calls", "lightmap_samples",
    "lightmap_resamples", "lightmap_oob", "texture_samples", "powf_calls", "atan2_calls",
    "reflection_rays", "sky_hits", "max_depth", "max_march_steps", "pixels",
    "march_box_sdf", "shadow_box_sdf",
)

_F = POINTER(c_float)
_libs = {}


def build():
    """Compile the oracle if its shared objects are missing or stale."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-j2"], check=True, stdout=subprocess.DEVNULL)


def load(counting=False):
    key = bool(counting)
    if key in _libs:
        return _libs[key]
    name = "librm_oracle_count.so" if counting else "librm_oracle.so"
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        build()
    lib = ctypes.CDLL(path)
    lib.orc_engine_create.restype = c_void_p
    lib.orc_engine_create.argtypes = [c_int]
    lib.orc_engine_destroy.argtypes = [c_void_p]
    lib.orc_last_error.restype = c_char_p
    lib.orc_last_error.argtypes = [c_void_p]
    lib.orc_texture_create_rgb8.argtypes = [c_void_p, POINTER(c_uint8), c_int, c_int, c_float, c_float]
    lib.orc_add_sphere.argtypes = [c_void_p, _F, c_float, _F, _F]
    lib.orc_add_xplane.argtypes = [c_void_p, c_float, c_float, _F, _F]
    lib.orc_add_yplane.argtypes = [c_void_p, c_float, c_float, _F, _F, c_int]
    lib.orc_add_zplane.argtypes = [c_void_p, c_float, c_float, _F, _F]
    lib.orc_add_plane.argtypes = [c_void_p, _F, c_float, _F, _F]
    lib.orc_add_box.argtypes = [c_void_p, _F, _F, _F, _F]
    lib.orc_set_smooth_union.argtypes = [c_void_p, c_float]
    lib.orc_set_ssaa.argtypes = [c_void_p, c_int]
    lib.orc_set_tetrahedral_normals.argtypes = [c_void_p, c_int]
    lib.orc_set_refraction.argtypes = [c_void_p, c_int]
    lib.orc_refraction_depth_cuts.restype = ctypes.c_uint64
    lib.orc_refraction_depth_cuts.argtypes = [c_void_p]
    lib.orc_kat_internal_march.argtypes = [_F, c_float, _F, _F]
    lib.orc_kat_smooth_min.restype = c_float
    lib.orc_kat_smooth_min.argtypes = [c_float, c_float, c_float]
    lib.orc_set_light.argtypes = [c_void_p, _F, c_float]
    lib.orc_set_camera.argtypes = [c_void_p, _F]
    lib.orc_get_camera.argtypes = [c_void_p, _F]
    lib.orc_compute_lightmaps.argtypes = [c_void_p, c_int, _F, _F]
    lib.orc_get_lightmap.argtypes = [c_void_p, c_int, _F]
    lib.orc_set_lightmap.argtypes = [c_void_p, c_int, _F]
    lib.orc_render.argtypes = [c_void_p, POINTER(c_uint8), c_int, c_int, c_int, c_int, c_int, c_int, c_int, _F, c_int, c_int]
    lib.orc_pose.argtypes = [c_void_p, c_int, c_int, _F, c_int, _F]
    lib.orc_trace_primary.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_int32), POINTER(c_int32), _F]
    lib.orc_fxaa.argtypes = [POINTER(c_uint8), POINTER(c_uint8), c_int, c_int, c_int]
    lib.orc_counters_get.argtypes = [c_void_p, POINTER(c_uint64)]
    lib.orc_counters_reset.argtypes = [c_void_p]
    lib.orc_kat_rgb.argtypes = [c_float, c_float, c_float, _F]
    lib.orc_kat_aces.argtypes = [_F, _F]
    lib.orc_kat_to_pixel_range.argtypes = [c_float]
    lib.orc_kat_pack_pixels.argtypes = [c_int, _F, POINTER(ctypes.c_uint32)]
    lib.orc_kat_texture_sample.argtypes = [c_void_p, c_int, c_float, c_float, _F]
    lib.orc_kat_sdf.restype = c_float
    lib.orc_kat_sdf.argtypes = [c_void_p, c_int, _F]
    lib.orc_kat_normal.argtypes = [c_void_p, c_int, _F, _F]
    lib.orc_kat_march.argtypes = [c_void_p, _F, _F, c_int, c_int]
    lib.orc_kat_shadow.restype = c_float
    lib.orc_kat_shadow.argtypes = [c_void_p, _F, _F, c_int, c_float, c_float]
    lib.orc_kat_phong.argtypes = [_F, _F, c_float, c_float, _F, _F, _F]
    lib.orc_kat_bilinear.argtypes = [c_float, c_float, _F, _F]
    lib.orc_kat_sample_lightmap.argtypes = [c_void_p, c_int, _F, _F]
    lib.orc_kat_sample_pos.argtypes = [c_void_p, c_int, c_int, c_int, _F]
    lib.orc_kat_pixel_direction.argtypes = [c_int, c_int, c_int, c_int, _F, _F]
    lib.orc_kat_shade.argtypes = [c_void_p, c_int, _F, _F, _F]
    lib.orc_kat_refract.argtypes = [_F, _F, c_float, _F]
    _libs[key] = lib
    return lib


def _fa(values):
    return (c_float * len(values))(*[float(v) for v in values])


class OracleEngine:
    """The oracle's Engine, built from the same Scene description as the GPU engine."""

    def __init__(self, scene: Scene, map_size=8, counting=False):
        self.lib = load(counting)
        self.map_size = map_size
        self.handle = c_void_p(self.lib.orc_engine_create(map_size))
        self.n_objects = len(scene.objects)
        build_scene(self, scene)

    def close(self):
        if self.handle:
            self.lib.orc_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # build_scene protocol
    def texture(self, tex):
        px = np.ascontiguousarray(tex.rgb8, dtype=np.uint8)
        return self.lib.orc_texture_create_rgb8(self.handle, px.ctypes.data_as(POINTER(c_uint8)), tex.width, tex.height, tex.uscale, tex.vscale)

    def add_sphere(self, pos, radius, mat, colour):
        self.lib.orc_add_sphere(self.handle, _fa(pos), radius, _fa(mat), _fa(colour))

    def add_xplane(self, x, d, mat, colour):
        self.lib.orc_add_xplane(self.handle, x, d, _fa(mat), _fa(colour))

    def add_yplane(self, y, d, mat, colour, tex):
        assert self.lib.orc_add_yplane(self.handle, y, d, _fa(mat), _fa(colour), tex) >= 0

    def add_zplane(self, z, d, mat, colour):
        self.lib.orc_add_zplane(self.handle, z, d, _fa(mat), _fa(colour))

    def add_plane(self, normal, distance, mat, colour):
        self.lib.orc_add_plane(self.handle, _fa(normal), distance, _fa(mat), _fa(colour))

    def add_box(self, pos, half, mat, colour):
        self.lib.orc_add_box(self.handle, _fa(pos), _fa(half), _fa(mat), _fa(colour))

    def set_smooth_union(self, k):
        self.lib.orc_set_smooth_union(self.handle, k)

    def set_ssaa(self, samples):
        self.lib.orc_set_ssaa(self.handle, samples)

    def set_tetrahedral_normals(self, on):
        self.lib.orc_set_tetrahedral_normals(self.handle, 1 if on else 0)

    def set_refraction(self, on):
        self.lib.orc_set_refraction(self.handle, 1 if on else 0)

    def refraction_depth_cuts(self):
        return int(self.lib.orc_refraction_depth_cuts(self.handle))

    def set_light(self, pos, intensity):
        self.lib.orc_set_light(self.handle, _fa(pos), intensity)

    def set_camera(self, pos):
        self.lib.orc_set_camera(self.handle, _fa(pos))

    # reference operations
    @property
    def camera_position(self):
        out = (c_float * 3)()
        self.lib.orc_get_camera(self.handle, out)
        return tuple(out)

    def compute_lightmaps(self, bounces=4, stages=False):
        """Returns None, or (lightmaps, emissive) of shape [bounces+1][n_obj][M][M][3] if stages."""
        if not stages:
            rc = self.lib.orc_compute_lightmaps(self.handle, bounces, None, None)
            if rc:
                raise RuntimeError(self.lib.orc_last_error(self.handle).decode())
            return None
        shape = (bounces + 1, self.n_objects, self.map_size, self.map_size, 3)
        lm = np.zeros(shape, dtype=np.float32)
        em = np.zeros(shape, dtype=np.float32)
        rc = self.lib.orc_compute_lightmaps(self.handle, bounces, lm.ctypes.data_as(_F), em.ctypes.data_as(_F))
        if rc:
            raise RuntimeError(self.lib.orc_last_error(self.handle).decode())
        return lm, em

    def get_lightmap(self, obj):
        out = np.empty((self.map_size, self.map_size, 3), dtype=np.float32)
        assert self.lib.orc_get_lightmap(self.handle, obj, out.ctypes.data_as(_F)) == 0
        return out

    def set_lightmap(self, obj, values):
        v = np.ascontiguousarray(values, dtype=np.float32)
        assert self.lib.orc_set_lightmap(self.handle, obj, v.ctypes.data_as(_F)) == 0

    def render(self, width, height, mouse_x=0, mouse_y=0, rel_move=(0.0, 0.0, 0.0), scroll=0, rows=None, threads=0, row_step=1):
        row0, row1 = (0, height) if rows is None else rows
        n_rows = (row1 - row0 + row_step - 1) // row_step
        buf = np.empty((n_rows, width, 4), dtype=np.uint8)
        self.lib.orc_render(self.handle, buf.ctypes.data_as(POINTER(c_uint8)), width, height, row0, row1, row_step, mouse_x, mouse_y, _fa(rel_move), scroll, threads)
        return buf

    def pose(self, mouse_x=0, mouse_y=0, rel_move=(0.0, 0.0, 0.0), scroll=0):
        out = (c_float * 8)()
        self.lib.orc_pose(self.handle, mouse_x, mouse_y, _fa(rel_move), scroll, out)
        return np.array(out, dtype=np.float32)

    def trace_primary(self, width, height, mouse_x=0, mouse_y=0, scroll=0):
        n = width * height
        hit = np.empty(n, dtype=np.int32)
        steps = np.empty(n, dtype=np.int32)
        pos = np.empty((n, 3), dtype=np.float32)
        self.lib.orc_trace_primary(self.handle, width, height, mouse_x, mouse_y, scroll, hit.ctypes.data_as(POINTER(c_int32)), steps.ctypes.data_as(POINTER(c_int32)), pos.ctypes.data_as(_F))
        return hit.reshape(height, width), steps.reshape(height, width), pos.reshape(height, width, 3)

    def shade(self, obj, pos, direction):
        out = (c_float * 3)()
        self.lib.orc_kat_shade(self.handle, int(obj), _fa(pos), _fa(direction), out)
        return np.array(out, dtype=np.float32)

    def counters(self):
        out = (c_uint64 * len(COUNTER_NAMES))()
        self.lib.orc_counters_get(self.handle, out)
        return {n: int(out[i]) for i, n in enumerate(COUNTER_NAMES)}

    def reset_counters(self):
        self.lib.orc_counters_reset(self.handle)


def fxaa(frame, in_place=False):
    """frame: [H][W][4] uint8.  Returns the filtered copy."""
    lib = load()
    src = np.ascontiguousarray(frame, dtype=np.uint8)
    dst = np.empty_like(src)
    h, w = src.shape[:2]
    lib.orc_fxaa(src.ctypes.data_as(POINTER(c_uint8)), dst.ctypes.data_as(POINTER(c_uint8)), w, h, 1 if in_place else 0)
    return dst
