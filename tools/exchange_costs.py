"""What one rank of eight does per 3840x2160 frame besides waiting for the links, on one GPU: a group
of world 8 whose exchange function returns at once (nothing is copied, so the frames are not
meaningful - only the time is).  Per frame the rank renders its 34 strips and then, with the
all-gather exchange, packs them and unpacks (and filters) the whole frame every time; with the
presenter exchange it re-orders (and filters) the whole frame only for every eighth frame.  The
time per frame against an eighth of the single-GPU frame is the scaling efficiency the rank's own
GPU work allows, whatever the links do.
Run on the GPU box: python tools/exchange_costs.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import Engine, default_scene
from raymarching_amd.group import Group

W, H, WORLD = 3840, 2160, 8
FRAMES = int(os.environ.get("RM_FRAMES", 240))
scene = default_scene()
whole = Engine(scene, W, H)
whole.compute_lightmaps(4)
buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
for _ in range(20):
    whole.render_rows_device(buf.data_ptr(), 0, H)
whole.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    whole.render_rows_device(buf.data_ptr(), 0, H)
whole.synchronize()
n1_ms = (time.perf_counter() - t0) / 60 * 1e3
lightmaps = [whole.get_lightmap(i) for i in range(whole.n_objects)]
whole.close()
print(f"single GPU: {n1_ms:.4f} ms per frame; an eighth of it: {n1_ms / WORLD * 1e3:.1f} us")


def nothing(d_send, d_recv, nbytes, stream):
    return 0


def nothing_to(d_send, d_recv, nbytes, root, stream):
    return 0


outs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
for rank in (0, 7):
    for in_flight in [int(x) for x in os.environ.get("RM_IN_FLIGHT", "1,3").split(",")]:
        eng = Engine(scene, W, H)
        for i, lm in enumerate(lightmaps):
            eng.set_lightmap(i, lm)
        grp = Group(eng, rank, WORLD, frames_in_flight=in_flight, all_gather=nothing, gather=nothing_to)
        for exchange in ("all_gather", "presenter"):
            for fxaa in (False, True):
                grp.set_exchange(exchange)
                for f in range(48):
                    grp.frame(outs[f % 4].data_ptr(), fxaa)
                grp.wait()
                t0 = time.perf_counter()
                for f in range(FRAMES):
                    grp.frame(outs[f % 4].data_ptr(), fxaa)
                grp.wait()
                ms = (time.perf_counter() - t0) / FRAMES * 1e3
                print(f"rank {rank} of {WORLD}, {in_flight} frame(s) in flight, {exchange:10s}{' + FXAA' if fxaa else '       '}: "
                      f"{ms * 1e3:6.1f} us per frame -> at most {n1_ms / WORLD / ms:.1%} of linear scaling", flush=True)
        grp.close()
        eng.close()
