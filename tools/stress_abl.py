"""Frame time of the smooth-union stress scene at 3840x2160 for every ablation build under tools/abl/
(librm_abl<bits>.so, built with -DRM_ABL=<bits>: 1 = no shadow rays, 2 = no reflections): what share of
the frame each part is.  Run on the GPU box from the repo root."""
import os, sys, time, glob
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raymarching_amd import _lib
from raymarching_amd.scene import stress_scene, Box
W, H = 3840, 2160
scene = stress_scene(250, boxes=True, smooth_union=0.08)
lm = np.full((8, 8, 3), 0.1, dtype=np.float32)
for path in sorted(glob.glob("tools/abl/librm_abl*.so")):
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    from raymarching_amd.engine import Engine
    eng = Engine(scene, W, H)
    for i, obj in enumerate(scene.objects):
        if not isinstance(obj, Box):
            eng.set_lightmap(i, lm)
    eng.set_march_cap(1024)
    eng.input(0, 0, (0, 0, 0), 0)
    buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    eng.render_rows_device(buf.data_ptr(), 0, H); eng.synchronize()
    t = time.time()
    for _ in range(3):
        eng.render_rows_device(buf.data_ptr(), 0, H)
    eng.synchronize()
    print(os.path.basename(path), f"{(time.time()-t)/3*1e3:.2f} ms/frame", eng.stats(), flush=True)
    hit, steps, pos = eng.trace_primary()
    print("   primary steps mean", float(steps.mean()), "max", int(steps.max()), "sky", float((hit < 0).mean()))
    eng.close()
