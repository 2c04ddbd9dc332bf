"""Host-side cost of one multi-GPU bench step (launch of a rank's strips, asynchronous all-gather,
wait, re-ordering copy), measured with a one-rank RCCL group on one GPU: the GPU work is an eighth
of a frame, so the loop shows how many steps per second the host can issue.
Run on the GPU box: python tools/step_overhead.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

from raymarching_amd import Engine, default_scene
from raymarching_amd.dist import deinterleave_strips, strip_plan

W, H, SHARE = 3840, 2160, 8
dist.init_process_group("nccl", rank=0, world_size=1)
eng = Engine(default_scene(), W, H)
main = torch.cuda.current_stream()
eng.set_stream(main.cuda_stream)
eng.compute_lightmaps(4)
eng.input(0, 0, (0.0, 0.0, 0.0), 0)
first, stride, per = strip_plan(H, 0, SHARE)
K = 3
streams = [torch.cuda.Stream() for _ in range(K)]
bands = [torch.zeros((per * 8, W, 4), dtype=torch.uint8, device="cuda") for _ in range(K)]
gathered = [torch.zeros((per * 8, W, 4), dtype=torch.uint8, device="cuda") for _ in range(K)]
# the re-ordering copy of a whole frame (what an 8-rank run does per frame)
frames = [torch.zeros((SHARE * per * 8, W, 4), dtype=torch.uint8, device="cuda") for _ in range(K)]
final = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(K)]


def run(n, parts):
    t = dict.fromkeys(("render", "gather", "wait", "reorder"), 0.0)
    pending = [None] * K
    for s in range(n):
        b = s % K
        with torch.cuda.stream(streams[b]):
            t0 = time.perf_counter()
            if pending[b] is not None:
                pending[b].wait()
            t1 = time.perf_counter()
            if pending[b] is not None and "reorder" in parts:
                final[b].copy_(deinterleave_strips(frames[b], SHARE, per, W, H))
            t2 = time.perf_counter()
            eng.set_stream(streams[b].cuda_stream)
            eng.render_strips_device(bands[b].data_ptr(), first, stride, per)
            t3 = time.perf_counter()
            if "gather" in parts:
                pending[b] = dist.all_gather_into_tensor(gathered[b].view(-1), bands[b].view(-1), async_op=True)
            t4 = time.perf_counter()
        t["wait"] += t1 - t0
        t["reorder"] += t2 - t1
        t["render"] += t3 - t2
        t["gather"] += t4 - t3
    return t


for parts in (("render",), ("render", "gather"), ("render", "gather", "reorder")):
    run(30, parts)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    t = run(n, parts)
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"{'+'.join(parts):24s} issue {issue / n * 1e6:6.1f} us/step, with GPU drain {total / n * 1e6:6.1f} us/step;  "
          + "  ".join(f"{k} {v / n * 1e6:.1f}" for k, v in t.items()), flush=True)
eng.close()
dist.destroy_process_group()
