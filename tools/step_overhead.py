"""Host-side cost of one frame of a several-GPU run: rm_group_frame (a rank's strips, pack, RCCL
all-gather on the frame's stream, unpack) issued through ctypes, measured with a one-rank RCCL group
on one GPU.  The GPU work per call is about an eighth of a 3840x2160 frame, so the issue rate shows
what the host can sustain.
Run on the GPU box: python tools/step_overhead.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import Engine, _lib, default_scene
from raymarching_amd.group import Group, unique_id

W, H = 3840, 2160
# an eighth of the frame's strips: an engine of one eighth of the height renders as many tiles as one
# of eight ranks would (the per-call host work does not depend on which strips they are)
eng = Engine(default_scene(), W, H // 8)
eng.compute_lightmaps(4)
out = [torch.zeros((H // 8, W, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
lib = _lib.load()
zero = _lib.f3((0.0, 0.0, 0.0))
for in_flight in (1, 3):
    grp = Group(eng, 0, 1, unique_id(), frames_in_flight=in_flight)
    for fxaa in (0, 1):
        for _ in range(30):
            grp.frame(out[0].data_ptr(), bool(fxaa))
        grp.wait()
        n = 400
        ptrs = [ctypes.c_void_p(o.data_ptr()) for o in out]
        t0 = time.perf_counter()
        for s in range(n):
            lib.rm_group_frame(grp.handle, ptrs[s & 3], fxaa, 0, 0, zero, 0)
        issue = time.perf_counter() - t0
        grp.wait()
        total = time.perf_counter() - t0
        print(f"frames in flight {in_flight}, fxaa {fxaa}: issue {issue / n * 1e6:6.1f} us/frame, with GPU drain {total / n * 1e6:6.1f} us/frame", flush=True)
    grp.close()
eng.close()
