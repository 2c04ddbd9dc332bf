"""Times the 4K default-scene frame for one or more builds of librm_hip.so in ONE process
(interleaved rounds), and checks that every build produces the same image.

usage: quick_bench.py [lib.so ...]   (default: the in-tree library)
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import _lib, default_scene

W, H = int(os.environ.get("RM_W", 3840)), int(os.environ.get("RM_H", 2160))
ROUNDS, FRAMES = int(os.environ.get("RM_ROUNDS", 5)), int(os.environ.get("RM_FRAMES", 10))
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
bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in engines]
for e, b in zip(engines, bufs):
    e.render_rows_device(b.data_ptr(), 0, H)
torch.cuda.synchronize()
for i, b in enumerate(bufs[1:], 1):
    same = torch.equal(b, bufs[0])
    print(f"{libs[i]}: image {'identical to' if same else 'DIFFERS from'} {libs[0]}")
times = [[] for _ in engines]
for r in range(ROUNDS):
    # (the build that goes first after a pause measures up to 3 % slower: the rounds start with each in turn)
    for i in [(r + k) % len(engines) for k in range(len(engines))]:
        e, b = engines[i], bufs[i]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(FRAMES):
            e.render_rows_device(b.data_ptr(), 0, H)
        ev1.record()
        torch.cuda.synchronize()
        times[i].append(ev0.elapsed_time(ev1) / FRAMES)
torch.cuda.synchronize()
for i, b in enumerate(bufs[1:], 1):  # again after many frames: builds that plan a frame from the last one's timings
    if not torch.equal(b, bufs[0]):
        print(f"{libs[i]}: image DIFFERS from {libs[0]} after the timed frames")
for path, t in zip(libs, times):
    t.sort()
    print(f"{os.path.basename(path):40s} median {t[len(t)//2]:.4f} ms  min {t[0]:.4f} ms  -> {W*H/t[len(t)//2]/1e3:.0f} Mrays/s")
