"""Times a frame with a fresh launch plan for every launch (RM_PLAN_EVERY=1) against the default (a plan
kept for eight launches), two engines of the in-tree library in one process, rounds starting with each in
turn; the frames are compared byte for byte.  RM_W, RM_H: frame size (default 1920x1080)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import default_scene
from raymarching_amd.engine import Engine

W, H = int(os.environ.get("RM_W", 1920)), int(os.environ.get("RM_H", 1080))
engines, bufs, names = [], [], ["fresh plan per launch", "plan kept for 8 launches"]
for every in ("1", "8"):
    os.environ["RM_PLAN_EVERY"] = every  # read by an engine at its first planned launch
    e = Engine(default_scene(), W, H)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.compute_lightmaps(4)
    e.input(0, 0, (0, 0, 0), 0)
    b = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    e.render_rows_device(b.data_ptr(), 0, H)
    torch.cuda.synchronize()
    engines.append(e)
    bufs.append(b)
times = [[], []]
for r in range(10):
    for i in ((r + k) % 2 for k in range(2)):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(16):
            engines[i].render_rows_device(bufs[i].data_ptr(), 0, H)
        ev1.record()
        torch.cuda.synchronize()
        times[i].append(ev0.elapsed_time(ev1) / 16)
print("frames identical:", bool(torch.equal(bufs[0], bufs[1])))
for n, t in zip(names, times):
    t.sort()
    print(f"{n:28s} median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f} ms")
