"""Latency of the slowest tiles: times the 8-row strips of a 1920x1080 frame that hold the longest
waves (mirror sphere inter-reflections), 240 waves per launch - every wave has a SIMD to itself.

usage: tile_latency.py [lib.so ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import _lib, default_scene

W, H = 1920, 1080
STRIPS = [int(x) for x in os.environ.get("RM_STRIPS", "97,98,99,109,110").split(",")]
libs = sys.argv[1:] or [_lib.LIB_PATH]
engines = []
for path in libs:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    from raymarching_amd.engine import Engine

    e = Engine(default_scene(), W, H)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.compute_lightmaps(4)
    e.input(0, 0, (0, 0, 0), 0)
    engines.append(e)
buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
for path, e in zip(libs, engines):
    res = []
    for strip in STRIPS:
        r0 = strip * 8
        ts = []
        for _ in range(7):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            e.render_rows_device(buf.data_ptr() + r0 * W * 4, r0, r0 + 8)
            ev1.record()
            torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1) * 1e3)
        ts.sort()
        res.append(ts[len(ts) // 2])
    print(f"{os.path.basename(path):28s} strip latency (us): " + "  ".join(f"{s}:{t:.0f}" for s, t in zip(STRIPS, res)))
