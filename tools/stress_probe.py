import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from raymarching_amd import Engine
from raymarching_amd.scene import stress_scene, Box
for su in (0.08, 0.0):
    scene = stress_scene(250, boxes=True, smooth_union=su)
    W, H = 1920, 1080
    eng = Engine(scene, W, H)
    u, v = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    lm = np.stack([0.10 + 0.02 * u, 0.12 + 0.01 * v, 0.08 + 0.01 * (u + v)], axis=-1).astype(np.float32)
    for i, obj in enumerate(scene.objects):
        if not isinstance(obj, Box):
            eng.set_lightmap(i, lm)
    eng.set_march_cap(1024)
    buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    eng.input(0, 0, (0, 0, 0), 0)
    eng.render_rows_device(buf.data_ptr(), 0, H); eng.synchronize()
    t = time.time()
    for _ in range(3):
        eng.render_rows_device(buf.data_ptr(), 0, H)
    eng.synchronize()
    print(f"smooth_union={su}: {W}x{H} {(time.time()-t)/3*1e3:.2f} ms/frame", eng.stats())
    hit, steps, pos = eng.trace_primary()
    print("  primary steps mean", steps.mean(), "max", steps.max())
