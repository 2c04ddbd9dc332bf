"""Time of one radiosity bounce (rad_bounce_kernel) for slab depths 64 and 128 at several lightmap sizes,
with the lightmaps of both compared bit for bit.  Run on the GPU box: python tools/bounce_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raymarching_amd import _lib, default_scene

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from raymarching_amd.engine import Engine

for M in (8, 16, 32, 64, 128):
    n = 8 * M * M
    results = {}
    for chunk in ("128", "192"):
        os.environ["RM_BOUNCE_CHUNK"] = chunk
        eng = Engine(default_scene(), 64, 64, map_size=M)
        eng.compute_lightmaps(2)
        lm = np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])
        ts = []
        for _ in range(5):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            eng.radiosity_bounce(0, n)
            ev1.record()
            torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1))
        eng.close()
        ts.sort()
        results[chunk] = (ts[len(ts) // 2], lm)
    os.environ.pop("RM_BOUNCE_CHUNK")
    same = all(np.array_equal(results["128"][1].view(np.uint32), results[c][1].view(np.uint32)) for c in results)
    print(f"MAP_SIZE {M:3d} ({n} patches): " + ", ".join(f"depth {c} {results[c][0]:8.4f} ms ({4 * n * n / results[c][0] / 1e9:5.2f} TB/s)" for c in results)
          + f", lightmaps {'identical' if same else 'DIFFER'}", flush=True)
