"""Frames of a launch planned from the previous frame's tile timings (slow tiles in quarters) against
the first frame, which has no history: same bytes, at several sizes, for whole frames and strips."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import Engine, default_scene

for W, H in ((1920, 1080), (800, 600), (1000, 700), (2560, 1440)):
    e = Engine(default_scene(), W, H)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.compute_lightmaps(4)
    e.input(0, 0, (0, 0, 0), 0)
    frames = []
    for f in range(5):
        buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        e.render_rows_device(buf.data_ptr(), 0, H)
        torch.cuda.synchronize()
        frames.append(buf)
    same = [torch.equal(frames[0], f) for f in frames[1:]]
    # a moved camera: the plan comes from the old pose's timings
    e.input(40, -25, (0.3, 0.0, 0.4), 1)
    moved = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    e.render_rows_device(moved.data_ptr(), 0, H)
    torch.cuda.synchronize()
    e2 = Engine(default_scene(), W, H)
    e2.set_stream(torch.cuda.current_stream().cuda_stream)
    e2.compute_lightmaps(4)
    e2.input(40, -25, (0.3, 0.0, 0.4), 1)
    fresh = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    e2.render_rows_device(fresh.data_ptr(), 0, H)
    torch.cuda.synchronize()
    print(f"{W}x{H}: frames 2-5 equal frame 1: {same}; moved camera equals a fresh engine's frame: {torch.equal(moved, fresh)}", flush=True)
    assert all(same) and torch.equal(moved, fresh)
    e.close()
    e2.close()
print("ok")
