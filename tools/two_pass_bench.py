"""The frame in one kernel against the frame in two passes (primary hit + local shade, then the queued
reflection rays in full waves; RM_TWO_PASS=<waves of pass 2> when an engine is created): same bytes,
time per frame at 3840x2160 and 1920x1080, one frame at a time on one stream.
Run on the GPU box from the repo root: python tools/two_pass_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import Engine, default_scene

ROUNDS, FRAMES = 7, 20
for W, H in ((3840, 2160), (1920, 1080)):
    variants = [("one kernel", None)] + [(f"two passes, {n} waves", str(n)) for n in (2048, 4096, 8192, 16384)]
    engines = []
    for name, env in variants:
        if env is None:
            os.environ.pop("RM_TWO_PASS", None)
        else:
            os.environ["RM_TWO_PASS"] = env
        e = Engine(default_scene(), W, H)
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.compute_lightmaps(4)
        e.input(0, 0, (0.0, 0.0, 0.0), 0)
        engines.append(e)
    os.environ.pop("RM_TWO_PASS", None)
    bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in engines]
    for e, b in zip(engines, bufs):
        e.render_rows_device(b.data_ptr(), 0, H)
    torch.cuda.synchronize()
    same = [bool(torch.equal(b, bufs[0])) for b in bufs]
    times = [[] for _ in engines]
    for r in range(ROUNDS):
        for i, (e, b) in enumerate(zip(engines, bufs)):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(FRAMES):
                e.render_rows_device(b.data_ptr(), 0, H)
            ev1.record()
            torch.cuda.synchronize()
            times[i].append(ev0.elapsed_time(ev1) / FRAMES)
    for (name, _), t, ok, e in zip(variants, times, same, engines):
        t.sort()
        print(f"{W}x{H}  {name:28s} median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f} ms  frame {'identical' if ok else 'DIFFERS'}  caps {e.stats()}", flush=True)
    for e in engines:
        e.close()
