"""Times the FXAA pass over a rendered default-scene frame for one or more builds of librm_hip.so in
one process and checks that every build writes the same bytes.

usage: fxaa_bench.py [lib.so ...]   (default: the in-tree library); RM_W / RM_H set the frame size
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import _lib, default_scene

W, H = int(os.environ.get("RM_W", 3840)), int(os.environ.get("RM_H", 2160))
libs = sys.argv[1:] or [_lib.LIB_PATH]
first = None
# the same bytes moved by a plain device copy: what the pass could cost if its arithmetic were free
_a = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
_b = torch.zeros_like(_a)
ts = []
for _ in range(30):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _b.copy_(_a)
    ev1.record()
    torch.cuda.synchronize()
    ts.append(ev0.elapsed_time(ev1) * 1e3)
ts.sort()
print(f"{'device copy of the frame':24s} median {ts[len(ts) // 2]:6.1f} us  min {ts[0]:6.1f} us  ({W * H * 8 / ts[len(ts) // 2] / 1e6:.2f} TB/s)", flush=True)
for path in libs:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    from raymarching_amd.engine import Engine

    e = Engine(default_scene(), W, H)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.compute_lightmaps(4)
    e.input(0, 0, (0, 0, 0), 0)
    src = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    dst = torch.zeros_like(src)
    e.render_rows_device(src.data_ptr(), 0, H)
    ts = []
    for _ in range(30):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        e.fxaa_device(src.data_ptr(), dst.data_ptr(), 0, H)
        ev1.record()
        torch.cuda.synchronize()
        ts.append(ev0.elapsed_time(ev1) * 1e3)
    ts.sort()
    if first is None:
        first = dst.clone()
    same = torch.equal(dst, first)
    changed = float((dst != src).any(dim=-1).float().mean())
    print(f"{os.path.basename(path):24s} median {ts[len(ts) // 2]:6.1f} us  min {ts[0]:6.1f} us  "
          f"({W * H * 8 / ts[len(ts) // 2] / 1e6:.2f} TB/s of 8 B per pixel; {changed:.1%} of the pixels change)"
          + ("" if same else "  OUTPUT DIFFERS from the first build"), flush=True)
    e.close()
