"""Time of the form-factor launch (rad_form_kernel / rad_form_multi_kernel) for several numbers of sources per
workgroup (RM_RAD_PER) at three lightmap sizes, with the lightmaps of each compared bit for bit.
Run on the GPU box: python tools/form_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raymarching_amd import _lib, default_scene
from raymarching_amd.engine import Engine
for M in (8, 32, 64):
    n = 8 * M * M
    ref = None
    for per in ("1", "2", "4", "8", "16"):
        os.environ["RM_RAD_PER"] = per
        eng = Engine(default_scene(), 64, 64, map_size=M)
        eng.compute_lightmaps(1)
        lm = np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])
        ts = []
        for _ in range(3):
            eng.radiosity_begin()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(); eng.radiosity_form_factors(0, n); ev1.record(); torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1))
        eng.close()
        if ref is None: ref = lm
        print(f"MAP_SIZE {M} per {per}: form {min(ts):.4f} ms same={np.array_equal(ref.view(np.uint32), lm.view(np.uint32))}", flush=True)
