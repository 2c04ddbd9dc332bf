"""One rank's share of a 3840x2160 frame (every N-th 8-row strip) rendered with 1..4 frames in
flight on as many streams: how far pipelining hides the few slow tiles that bound a share's latency.
Run on the GPU box: python tools/share_pipeline.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import Engine, default_scene
from raymarching_amd.dist import strip_plan

W, H = 3840, 2160
FRAMES = 240
eng = Engine(default_scene(), W, H)
main = torch.cuda.current_stream()
eng.set_stream(main.cuda_stream)
eng.compute_lightmaps(4)
eng.input(0, 0, (0.0, 0.0, 0.0), 0)
streams = [torch.cuda.Stream() for _ in range(4)]
for world in (1, 2, 4, 8):
    first, stride, per = strip_plan(H, 0, world)
    bufs = [torch.zeros((per * 8, W, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
    line = []
    for depth in (1, 2, 3, 4):
        def run(n):
            for f in range(n):
                st = streams[f % depth]
                eng.set_stream(st.cuda_stream)
                eng.render_strips_device(bufs[f % depth].data_ptr(), first, stride, per)
        run(24)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(FRAMES)
        torch.cuda.synchronize()
        line.append((time.perf_counter() - t0) / FRAMES * 1e3)
    eng.set_stream(main.cuda_stream)
    ideal = line[0] if world == 1 else None
    print(f"1/{world} of the frame: " + "  ".join(f"{d} in flight {t:.4f} ms" for d, t in zip((1, 2, 3, 4), line)), flush=True)
eng.close()
