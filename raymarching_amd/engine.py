"""Host-side `Engine` over the C ABI: the reference's `Engine` interface
(engine.rs:37-41, 246, 329-333) with the pixel work done by the HIP kernels.
"""
import ctypes
import weakref
from ctypes import POINTER, c_float, c_int, c_int32, c_uint8, c_uint64, c_void_p

import numpy as np

from . import _lib
from .scene import Scene, build_scene

MAP_SIZE = 8  # radiosity.rs:12
WIDTH, HEIGHT = 800, 600  # engine.rs:9-10


class _SceneBuilder:
    """Adapter from scene.build_scene() to the rm_scene_* entry points."""

    def __init__(self, lib):
        self.lib = lib
        self.handle = c_void_p(lib.rm_scene_create())

    def _obj(self, rc, what):
        if rc < 0:
            _lib.check(-rc, what)

    def texture(self, tex):
        px = np.ascontiguousarray(tex.rgb8, dtype=np.uint8)
        rc = self.lib.rm_texture_create_rgb8(
            self.handle, px.ctypes.data_as(POINTER(c_uint8)), tex.width, tex.height, tex.uscale, tex.vscale
        )
        self._obj(rc, "rm_texture_create_rgb8")
        return rc

    def add_sphere(self, pos, radius, mat, colour):
        self._obj(self.lib.rm_scene_add_sphere(self.handle, _lib.f3(pos), radius, _lib.Material(*mat), _lib.f3(colour)), "rm_scene_add_sphere")

    def add_xplane(self, x, d, mat, colour):
        self._obj(self.lib.rm_scene_add_xplane(self.handle, x, d, _lib.Material(*mat), _lib.f3(colour)), "rm_scene_add_xplane")

    def add_yplane(self, y, d, mat, colour, tex):
        self._obj(self.lib.rm_scene_add_yplane(self.handle, y, d, _lib.Material(*mat), _lib.f3(colour), tex), "rm_scene_add_yplane")

    def add_zplane(self, z, d, mat, colour):
        self._obj(self.lib.rm_scene_add_zplane(self.handle, z, d, _lib.Material(*mat), _lib.f3(colour)), "rm_scene_add_zplane")

    def add_plane(self, normal, distance, mat, colour):
        self._obj(self.lib.rm_scene_add_plane(self.handle, _lib.f3(normal), distance, _lib.Material(*mat), _lib.f3(colour)), "rm_scene_add_plane")

    def add_box(self, pos, half, mat, colour):
        self._obj(self.lib.rm_scene_add_box(self.handle, _lib.f3(pos), _lib.f3(half), _lib.Material(*mat), _lib.f3(colour)), "rm_scene_add_box")

    def set_smooth_union(self, k):
        _lib.check(self.lib.rm_scene_set_smooth_union(self.handle, k), "rm_scene_set_smooth_union")

    def set_light(self, pos, intensity):
        _lib.check(self.lib.rm_scene_set_light(self.handle, _lib.f3(pos), intensity), "rm_scene_set_light")

    def set_camera(self, pos):
        _lib.check(self.lib.rm_scene_set_camera(self.handle, _lib.f3(pos)), "rm_scene_set_camera")

    def destroy(self):
        if self.handle:
            self.lib.rm_scene_destroy(self.handle)
            self.handle = None


class Engine:
    """`Engine { objects, camera_position, light }` bound to one MI355X.

    render() and compute_lightmaps() keep the reference's meaning; the *_device
    methods leave results in HBM for callers that hold device buffers.
    """

    def __init__(self, scene: Scene, width=WIDTH, height=HEIGHT, map_size=MAP_SIZE, device=0):
        self.lib = _lib.load()
        self.width, self.height, self.map_size, self.device = int(width), int(height), int(map_size), int(device)
        self.n_objects = len(scene.objects)
        builder = _SceneBuilder(self.lib)
        try:
            build_scene(builder, scene)
            handle = self.lib.rm_engine_create(builder.handle, self.width, self.height, self.map_size, self.device)
        finally:
            builder.destroy()
        if not handle:
            msg = self.lib.rm_last_error()
            raise _lib.RMError(f"rm_engine_create failed: {msg.decode() if msg else ''}")
        self.handle = c_void_p(handle)
        self._dependents = weakref.WeakSet()  # groups over this engine: they hold its streams and must go first

    def close(self):
        if getattr(self, "handle", None):
            for dep in list(getattr(self, "_dependents", ())):
                dep.close()
            self.lib.rm_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- streams ---------------------------------------------------------
    def set_stream(self, hip_stream: int):
        _lib.check(self.lib.rm_engine_set_stream(self.handle, c_void_p(hip_stream)), "rm_engine_set_stream")

    def synchronize(self):
        _lib.check(self.lib.rm_engine_synchronize(self.handle), "rm_engine_synchronize")

    # -- lightmaps (engine.rs:246) ---------------------------------------
    def compute_lightmaps(self, bounces=4):
        _lib.check(self.lib.rm_engine_compute_lightmaps(self.handle, bounces), "rm_engine_compute_lightmaps")

    def compute_lightmaps_static(self, bounces=4):
        """compute_lightmaps without any allocation: work buffers in the library's static device array
        (at most _lib.RM_STATIC_MAX_PATCHES patches; calls that overlap on one GPU race)."""
        _lib.check(self.lib.rm_engine_compute_lightmaps_static(self.handle, bounces), "rm_engine_compute_lightmaps_static")

    def radiosity_begin(self):
        _lib.check(self.lib.rm_engine_radiosity_begin(self.handle), "rm_engine_radiosity_begin")

    def radiosity_form_factors(self, p0, p1):
        _lib.check(self.lib.rm_engine_radiosity_form_factors(self.handle, p0, p1), "rm_engine_radiosity_form_factors")

    def radiosity_bounce(self, p0, p1):
        _lib.check(self.lib.rm_engine_radiosity_bounce(self.handle, p0, p1), "rm_engine_radiosity_bounce")

    def radiosity_commit(self):
        _lib.check(self.lib.rm_engine_radiosity_commit(self.handle), "rm_engine_radiosity_commit")

    def radiosity_buffers(self):
        """(device pointer of next-emissive [NP][3] f32, device pointer of lightmaps [NP][3] f32, NP)."""
        em, lm, n = c_void_p(), c_void_p(), c_int()
        _lib.check(self.lib.rm_engine_radiosity_buffers(self.handle, ctypes.byref(em), ctypes.byref(lm), ctypes.byref(n)), "rm_engine_radiosity_buffers")
        return em.value, lm.value, n.value

    def get_lightmap(self, obj):
        out = np.empty((self.map_size, self.map_size, 3), dtype=np.float32)
        _lib.check(self.lib.rm_engine_get_lightmap(self.handle, obj, out.ctypes.data_as(POINTER(c_float))), "rm_engine_get_lightmap")
        return out

    def set_lightmap(self, obj, values):
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(self.map_size, self.map_size, 3)
        _lib.check(self.lib.rm_engine_set_lightmap(self.handle, obj, v.ctypes.data_as(POINTER(c_float))), "rm_engine_set_lightmap")

    def save_lightmaps(self, path):
        """All lightmaps into one file (the reference keeps none: main.rs:73 recomputes them at every start)."""
        _lib.check(self.lib.rm_engine_save_lightmaps(self.handle, str(path).encode()), "rm_engine_save_lightmaps")

    def load_lightmaps(self, path):
        _lib.check(self.lib.rm_engine_load_lightmaps(self.handle, str(path).encode()), "rm_engine_load_lightmaps")

    # -- camera (engine.rs:63-75) ------------------------------------------
    def input(self, mouse_x=0, mouse_y=0, rel_move=(0.0, 0.0, 0.0), scroll=0):
        _lib.check(self.lib.rm_engine_input(self.handle, mouse_x, mouse_y, _lib.f3(rel_move), scroll), "rm_engine_input")

    @property
    def camera_position(self):
        out = (c_float * 3)()
        _lib.check(self.lib.rm_engine_get_camera(self.handle, out), "rm_engine_get_camera")
        return tuple(out)

    @camera_position.setter
    def camera_position(self, pos):
        _lib.check(self.lib.rm_engine_set_camera(self.handle, _lib.f3(pos)), "rm_engine_set_camera")

    # -- frames (engine.rs:38-127) -----------------------------------------
    def render(self, buffer=None, mouse_x=0, mouse_y=0, rel_move=(0.0, 0.0, 0.0), scroll=0):
        """Engine::render: returns/fills a [H][W][4] uint8 host array (B,G,R,0)."""
        if buffer is None:
            buffer = np.empty((self.height, self.width, 4), dtype=np.uint8)
        assert buffer.dtype == np.uint8 and buffer.flags["C_CONTIGUOUS"] and buffer.size == self.width * self.height * 4
        _lib.check(
            self.lib.rm_engine_render(self.handle, buffer.ctypes.data_as(POINTER(c_uint8)), self.width * 4, mouse_x, mouse_y, _lib.f3(rel_move), scroll),
            "rm_engine_render",
        )
        return buffer

    def render_async(self, host_ptr: int, mouse_x=0, mouse_y=0, rel_move=(0.0, 0.0, 0.0), scroll=0, pitch=None):
        """Enqueue a frame and its copy into page-locked memory at host_ptr (rm_host_alloc); returns at once."""
        pitch = self.width * 4 if pitch is None else pitch
        _lib.check(
            self.lib.rm_engine_render_async(self.handle, c_void_p(host_ptr), pitch, mouse_x, mouse_y, _lib.f3(rel_move), scroll),
            "rm_engine_render_async",
        )

    def render_wait(self):
        """Block until the oldest pending render_async frame has landed in its buffer."""
        _lib.check(self.lib.rm_engine_render_wait(self.handle), "rm_engine_render_wait")

    def render_rows_device(self, d_ptr: int, row_begin: int, row_end: int, pitch=None):
        """Rows [row_begin,row_end) at the current pose into device memory at d_ptr (async)."""
        pitch = self.width * 4 if pitch is None else pitch
        _lib.check(self.lib.rm_engine_render_rows_device(self.handle, c_void_p(d_ptr), pitch, row_begin, row_end), "rm_engine_render_rows_device")

    def render_strips_device(self, d_ptr: int, first_strip: int, strip_stride: int, n_strips: int, pitch=None):
        """Strips first, first+stride, ... (8 rows each) at the current pose, written compactly at d_ptr (async)."""
        pitch = self.width * 4 if pitch is None else pitch
        _lib.check(
            self.lib.rm_engine_render_strips_device(self.handle, c_void_p(d_ptr), pitch, first_strip, strip_stride, n_strips),
            "rm_engine_render_strips_device",
        )

    def pack_bgr_device(self, d_bgra: int, d_bgr: int, n_pixels: int):
        """B,G,R,0 pixels -> B,G,R triples (the form rows take between GPUs); n_pixels % 4 == 0 (async)."""
        _lib.check(self.lib.rm_engine_pack_bgr_device(self.handle, c_void_p(d_bgra), c_void_p(d_bgr), n_pixels), "rm_engine_pack_bgr_device")

    def unpack_bgr_strips_device(self, d_bgr: int, d_bgra_frame: int, world: int, strips_per_rank: int):
        """All-gathered packed strips of `world` ranks -> the whole B,G,R,0 frame in row order (async)."""
        _lib.check(
            self.lib.rm_engine_unpack_bgr_strips_device(self.handle, c_void_p(d_bgr), c_void_p(d_bgra_frame), world, strips_per_rank),
            "rm_engine_unpack_bgr_strips_device",
        )

    def fxaa_device(self, d_src: int, d_dst: int, row_begin: int, row_end: int):
        _lib.check(self.lib.rm_engine_fxaa_device(self.handle, c_void_p(d_src), c_void_p(d_dst), row_begin, row_end), "rm_engine_fxaa_device")

    # -- test and diagnostic entry points ----------------------------------
    def trace_primary(self):
        n = self.width * self.height
        hit = np.empty(n, dtype=np.int32)
        steps = np.empty(n, dtype=np.int32)
        pos = np.empty((n, 3), dtype=np.float32)
        _lib.check(
            self.lib.rm_engine_trace_primary(self.handle, hit.ctypes.data_as(POINTER(c_int32)), steps.ctypes.data_as(POINTER(c_int32)), pos.ctypes.data_as(POINTER(c_float))),
            "rm_engine_trace_primary",
        )
        shape = (self.height, self.width)
        return hit.reshape(shape), steps.reshape(shape), pos.reshape(shape + (3,))

    def probe_shade(self, obj, pos, direction):
        obj = np.ascontiguousarray(obj, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
        out = np.empty_like(pos)
        _lib.check(
            self.lib.rm_engine_probe_shade(self.handle, len(obj), obj.ctypes.data_as(POINTER(c_int32)), pos.ctypes.data_as(POINTER(c_float)), direction.ctypes.data_as(POINTER(c_float)), out.ctypes.data_as(POINTER(c_float))),
            "rm_engine_probe_shade",
        )
        return out

    def probe_pack(self, linear, exposure=1.0):
        """Tone map and pack [n][3] linear colours: (as the frames do it, by the exact operations alone)."""
        from ctypes import c_uint32

        linear = np.ascontiguousarray(linear, dtype=np.float32).reshape(-1, 3)
        fast = np.empty(len(linear), dtype=np.uint32)
        exact = np.empty(len(linear), dtype=np.uint32)
        _lib.check(
            self.lib.rm_engine_probe_pack(self.handle, len(linear), linear.ctypes.data_as(POINTER(c_float)), float(exposure),
                                          fast.ctypes.data_as(POINTER(c_uint32)), exact.ctypes.data_as(POINTER(c_uint32))),
            "rm_engine_probe_pack",
        )
        return fast, exact

    def set_ssaa(self, samples):
        """engine.rs:95-108: 4 = four rays per pixel, averaged; 1 = off."""
        _lib.check(self.lib.rm_engine_set_ssaa(self.handle, samples), "rm_engine_set_ssaa")

    def set_tetrahedral_normals(self, on):
        """objects.rs:142-149: the four-point normal instead of central differences."""
        _lib.check(self.lib.rm_engine_set_tetrahedral_normals(self.handle, 1 if on else 0), "rm_engine_set_tetrahedral_normals")

    def set_refraction(self, on):
        """engine.rs:196-228: reflective objects also pass a refracted ray (index ratio 2.5)."""
        _lib.check(self.lib.rm_engine_set_refraction(self.handle, 1 if on else 0), "rm_engine_set_refraction")

    def set_exact_occlusion(self, on):
        """radiosity.rs:130-141: march every patch pair and leave blocked pairs out of the sum (matters
        only for emission values that are not finite; see rm.h)."""
        _lib.check(self.lib.rm_engine_set_exact_occlusion(self.handle, 1 if on else 0), "rm_engine_set_exact_occlusion")

    def set_precision(self, precision):
        """'strict' (default: the reference's operations, bit for bit) or 'fast' (the sight-ray chain strict,
        the colour arithmetic relaxed; within 2/255 and 99.9 % identical pixels; see rm.h)."""
        code = {"strict": _lib.RM_PRECISION_STRICT, "fast": _lib.RM_PRECISION_FAST}[precision]
        _lib.check(self.lib.rm_engine_set_precision(self.handle, code), "rm_engine_set_precision")

    def set_march_cap(self, max_steps):
        _lib.check(self.lib.rm_engine_set_march_cap(self.handle, max_steps), "rm_engine_set_march_cap")

    def set_tile_splitting(self, on):
        """Slow tiles first and in quarters (small launches; see rm.h): on by default."""
        _lib.check(self.lib.rm_engine_set_tile_splitting(self.handle, 1 if on else 0), "rm_engine_set_tile_splitting")

    def split_tiles(self):
        """Tiles the most recent planned launch rendered in quarters (waits for the engine's work)."""
        n = ctypes.c_uint32(0)
        _lib.check(self.lib.rm_engine_split_tiles(self.handle, ctypes.byref(n)), "rm_engine_split_tiles")
        return int(n.value)

    def shader_clock_mhz(self, microseconds=200):
        """The shader clock right now (under whatever load the engine's streams carry), in MHz."""
        mhz = ctypes.c_double()
        _lib.check(self.lib.rm_engine_shader_clock(self.handle, int(microseconds), ctypes.byref(mhz)), "rm_engine_shader_clock")
        return float(mhz.value)

    def stats(self):
        out = (c_uint64 * _lib.RM_STAT_COUNT)()
        _lib.check(self.lib.rm_engine_get_stats(self.handle, out), "rm_engine_get_stats")
        return {name: int(out[i]) for i, name in enumerate(_lib.STAT_NAMES)}

    def reset_stats(self):
        _lib.check(self.lib.rm_engine_reset_stats(self.handle), "rm_engine_reset_stats")
