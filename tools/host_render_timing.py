"""Times rm_engine_render (frame copied into pageable host memory) at 4K: the PCIe-inclusive rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from raymarching_amd import Engine, default_scene

W, H = int(os.environ.get("RM_W", 3840)), int(os.environ.get("RM_H", 2160))
eng = Engine(default_scene(), W, H)
eng.compute_lightmaps(4)
buf = np.empty((H, W, 4), dtype=np.uint8)
for _ in range(5):
    eng.render(buf)
t = []
for _ in range(30):
    t0 = time.perf_counter()
    eng.render(buf)
    t.append(time.perf_counter() - t0)
t.sort()
med = t[len(t) // 2]
print(f"rm_engine_render {W}x{H} into pageable host memory: median {med*1e3:.3f} ms -> {W*H/med/1e6:.0f} Mrays/s")

import ctypes

from raymarching_amd import _lib

lib = _lib.load()
ptr = lib.rm_host_alloc(W * H * 4)
pinned = np.ctypeslib.as_array((ctypes.c_uint8 * (W * H * 4)).from_address(ptr)).reshape(H, W, 4)
for _ in range(5):
    eng.render(pinned)
t = []
for _ in range(30):
    t0 = time.perf_counter()
    eng.render(pinned)
    t.append(time.perf_counter() - t0)
t.sort()
med = t[len(t) // 2]
print(f"rm_engine_render {W}x{H} into rm_host_alloc memory: median {med*1e3:.3f} ms -> {W*H/med/1e6:.0f} Mrays/s")
assert np.array_equal(pinned, buf)

# pipelined: the copy of frame f overlaps the render of frame f+1
ptr2 = lib.rm_host_alloc(W * H * 4)
ptrs = [ptr, ptr2]
N = 60
for f in range(4):
    eng.render_async(ptrs[f & 1])
for _ in range(2):
    eng.render_wait()
t0 = time.perf_counter()
for f in range(N):
    eng.render_async(ptrs[f & 1])
eng.render_wait()
eng.render_wait()
dt = (time.perf_counter() - t0) / N
print(f"rm_engine_render_async {W}x{H}, two frames in flight: {dt*1e3:.3f} ms per frame -> {W*H/dt/1e6:.0f} Mrays/s")
lib.rm_host_free(ptr)
lib.rm_host_free(ptr2)
eng.close()

# rm_multi_render with this GPU listed once, twice and four times: the per-frame cost of the fan-out
# over host threads (on one GPU the strips still share one PCIe link, so no bandwidth is gained here)
from raymarching_amd.group import MultiEngine

ptr = lib.rm_host_alloc(W * H * 4)
pinned = np.ctypeslib.as_array((ctypes.c_uint8 * (W * H * 4)).from_address(ptr)).reshape(H, W, 4)
for n in (1, 2, 4):
    multi = MultiEngine(default_scene(), W, H, [0] * n)
    multi.compute_lightmaps(4)
    for _ in range(5):
        multi.render(host_ptr=ptr)
    t = []
    for _ in range(30):
        t0 = time.perf_counter()
        multi.render(host_ptr=ptr)
        t.append(time.perf_counter() - t0)
    t.sort()
    med = t[len(t) // 2]
    print(f"rm_multi_render {W}x{H}, GPU 0 listed {n}x, into rm_host_alloc memory: median {med*1e3:.3f} ms -> {W*H/med/1e6:.0f} Mrays/s"
          f"  frame {'identical' if np.array_equal(pinned, buf) else 'DIFFERS'}")
    multi.close()
lib.rm_host_free(ptr)
