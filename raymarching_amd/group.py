"""Several GPUs through the C ABI: `Group` (one rank per process and GPU, frames left in HBM,
strips all-gathered by RCCL inside the library) and `MultiEngine` (one process, several GPUs, frames
into host memory).  The reference has one process and a thread pool (rayon rows, engine.rs:93);
these stand where its one `Engine::render` call per frame stands (main.rs:141-145).
"""
import ctypes
import weakref
from ctypes import POINTER, c_double, c_int, c_uint8, c_void_p

import numpy as np

from . import _lib
from .engine import Engine, _SceneBuilder
from .scene import Scene, build_scene


def unique_id() -> bytes:
    """The 128 bytes every rank of a Group is created with; one rank makes them, the host passes them on."""
    lib = _lib.load()
    buf = (c_uint8 * _lib.RM_GROUP_ID_BYTES)()
    _lib.check(lib.rm_group_unique_id(buf), "rm_group_unique_id")
    return bytes(buf)


class Group:
    """Rank `rank` of `world` engines, one per GPU.  `all_gather(d_send, d_recv, bytes_per_rank, hip_stream)`
    replaces RCCL where given (rehearsals on one GPU, other transports); it returns 0 on success.
    `gather(d_send, d_recv, bytes_per_rank, root, hip_stream)` is its gather-to-one-rank counterpart for
    the "presenter" exchange (d_recv is None on ranks other than root)."""

    def __init__(self, engine: Engine, rank: int, world: int, group_id: bytes = None, frames_in_flight: int = 3, all_gather=None,
                 gather=None):
        self.lib = _lib.load()
        self.engine = engine
        self.rank, self.world = rank, world
        self._callback_error = None
        self.handle = None
        if all_gather is not None:
            def _cb(user, d_send, d_recv, nbytes, stream):
                try:
                    return int(all_gather(d_send, d_recv, nbytes, stream) or 0)
                except Exception as exc:  # an exception must not cross the C frames
                    self._callback_error = exc
                    return 1

            self._cb = _lib.ALLGATHER_FN(_cb)  # kept alive as long as the group
            handle = self.lib.rm_group_create_custom(engine.handle, rank, world, frames_in_flight, self._cb, None)
        else:
            if group_id is None or len(group_id) != _lib.RM_GROUP_ID_BYTES:
                raise ValueError("a Group over RCCL needs the 128-byte id of unique_id()")
            buf = (c_uint8 * _lib.RM_GROUP_ID_BYTES).from_buffer_copy(group_id)
            handle = self.lib.rm_group_create(engine.handle, buf, rank, world, frames_in_flight)
        if not handle:
            msg = self.lib.rm_last_error()
            raise _lib.RMError(f"rm_group_create failed: {msg.decode() if msg else ''}")
        self.handle = c_void_p(handle)
        engine._dependents.add(self)  # the engine closes its groups before itself
        if gather is not None:
            def _gcb(user, d_send, d_recv, nbytes, root, stream):
                try:
                    return int(gather(d_send, d_recv, nbytes, root, stream) or 0)
                except Exception as exc:
                    self._callback_error = exc
                    return 1

            self._gcb = _lib.GATHER_FN(_gcb)
            self._check(self.lib.rm_group_set_custom_gather(self.handle, self._gcb, None), "rm_group_set_custom_gather")

    def _check(self, rc, what):
        """_lib.check, with an exception raised inside the host's exchange function brought back out."""
        if rc != 0 and self._callback_error is not None:
            exc, self._callback_error = self._callback_error, None
            msg = self.lib.rm_last_error()
            raise _lib.RMError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}") from exc
        _lib.check(rc, what)

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.engine, "handle", None):
                self.lib.rm_group_destroy(self.handle)
            # (an engine that is already gone took its device memory along; there is nothing left to release safely)
            self.handle = None
            try:
                self.engine._dependents.discard(self)
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        r, w, s, b = c_int(), c_int(), c_int(), c_int()
        _lib.check(self.lib.rm_group_info(self.handle, ctypes.byref(r), ctypes.byref(w), ctypes.byref(s), ctypes.byref(b)), "rm_group_info")
        return {"rank": r.value, "world": w.value, "strips_per_rank": s.value, "bytes_per_pixel_exchanged": b.value}

    def communicators(self):
        """RCCL communicators the frames travel through (one per frame in flight; 0 for a custom exchange)."""
        n = c_int()
        _lib.check(self.lib.rm_group_communicators(self.handle, ctypes.byref(n)), "rm_group_communicators")
        return n.value

    def set_exchange(self, mode):
        """'all_gather' (every rank receives every frame) or 'presenter' (frame f lands on rank f mod world)."""
        code = {"all_gather": _lib.RM_EXCHANGE_ALL_GATHER, "presenter": _lib.RM_EXCHANGE_PRESENTER}[mode]
        self._check(self.lib.rm_group_set_exchange(self.handle, code), "rm_group_set_exchange")

    def next_presenter(self):
        """The rank the next frame lands on, or -1 for every rank."""
        r = c_int()
        _lib.check(self.lib.rm_group_next_presenter(self.handle, ctypes.byref(r)), "rm_group_next_presenter")
        return r.value

    def compute_lightmaps(self, bounces=4):
        self._check(self.lib.rm_group_compute_lightmaps(self.handle, bounces), "rm_group_compute_lightmaps")

    def frame(self, d_frame: int, fxaa=False, mouse_x=0, mouse_y=0, rel_move=(0.0, 0.0, 0.0), scroll=0):
        """Enqueue one frame of the group; the whole frame lands at device pointer `d_frame` on this rank
        (on the frame's presenter alone with the 'presenter' exchange; d_frame may be 0 elsewhere)."""
        self._check(self.lib.rm_group_frame(self.handle, c_void_p(d_frame or None), 1 if fxaa else 0, mouse_x, mouse_y, _lib.f3(rel_move), scroll),
                    "rm_group_frame")

    def wait(self):
        self._check(self.lib.rm_group_wait(self.handle), "rm_group_wait")

    def set_timing(self, on=True):
        _lib.check(self.lib.rm_group_set_timing(self.handle, 1 if on else 0), "rm_group_set_timing")

    def render_time(self):
        """(mean render-kernel ms over the timed frames, number of frames)."""
        ms, n = c_double(), c_int()
        _lib.check(self.lib.rm_group_render_time(self.handle, ctypes.byref(ms), ctypes.byref(n)), "rm_group_render_time")
        return ms.value, n.value

    def exchange_time(self):
        """(mean ms of the collective alone over the timed frames, number of frames)."""
        ms, n = c_double(), c_int()
        _lib.check(self.lib.rm_group_exchange_time(self.handle, ctypes.byref(ms), ctypes.byref(n)), "rm_group_exchange_time")
        return ms.value, n.value


class _BorrowedEngine(Engine):
    """An engine owned by a MultiEngine: the Engine interface without ownership of the handle."""

    def __init__(self, lib, handle, width, height, map_size, device, n_objects):
        self.lib = lib
        self.handle = c_void_p(handle)
        self.width, self.height, self.map_size, self.device, self.n_objects = width, height, map_size, device, n_objects
        self._dependents = weakref.WeakSet()

    def close(self):
        self.handle = None


class MultiEngine:
    """`Engine { objects, camera_position, light }` over several GPUs of one process: compute_lightmaps()
    and render() keep the reference's meaning (engine.rs:246, 38-41); GPU i of n renders strips i, i+n, ..."""

    def __init__(self, scene: Scene, width, height, devices, map_size=8):
        self.lib = _lib.load()
        self.width, self.height, self.map_size = int(width), int(height), int(map_size)
        self.devices = [int(d) for d in devices]
        self.n_objects = len(scene.objects)
        builder = _SceneBuilder(self.lib)
        try:
            build_scene(builder, scene)
            arr = (c_int * len(self.devices))(*self.devices)
            handle = self.lib.rm_multi_create(builder.handle, self.width, self.height, self.map_size, arr, len(self.devices))
        finally:
            builder.destroy()
        if not handle:
            msg = self.lib.rm_last_error()
            raise _lib.RMError(f"rm_multi_create failed: {msg.decode() if msg else ''}")
        self.handle = c_void_p(handle)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.rm_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def engine(self, index) -> Engine:
        h = self.lib.rm_multi_engine(self.handle, index)
        if not h:
            raise _lib.RMError(self.lib.rm_last_error().decode())
        return _BorrowedEngine(self.lib, h, self.width, self.height, self.map_size, self.devices[index], self.n_objects)

    def compute_lightmaps(self, bounces=4):
        _lib.check(self.lib.rm_multi_compute_lightmaps(self.handle, bounces), "rm_multi_compute_lightmaps")

    def render(self, buffer=None, mouse_x=0, mouse_y=0, rel_move=(0.0, 0.0, 0.0), scroll=0, host_ptr=None, pitch=None):
        """Engine::render: fills a [H][W][4] uint8 host array (or page-locked memory at `host_ptr`)."""
        pitch = self.width * 4 if pitch is None else pitch
        if host_ptr is None:
            if buffer is None:
                buffer = np.empty((self.height, self.width, 4), dtype=np.uint8)
            assert buffer.dtype == np.uint8 and buffer.flags["C_CONTIGUOUS"] and buffer.size == self.width * self.height * 4
            ptr = buffer.ctypes.data_as(POINTER(c_uint8))
        else:
            ptr = ctypes.cast(c_void_p(host_ptr), POINTER(c_uint8))
        _lib.check(self.lib.rm_multi_render(self.handle, ptr, pitch, mouse_x, mouse_y, _lib.f3(rel_move), scroll), "rm_multi_render")
        return buffer
