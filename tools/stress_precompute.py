"""Times the lightmap precompute on the stress scene and the stress frame rendered from its result
(beside the frame rendered from the fixed pattern)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raymarching_amd import Engine
from raymarching_amd.scene import Box, stress_lightmap_pattern, stress_scene

W, H = int(os.environ.get("RM_W", 7680)), int(os.environ.get("RM_H", 4320))
scene = stress_scene(250, boxes=True, smooth_union=0.08)
n_lm = sum(1 for o in scene.objects if not isinstance(o, Box))
print(f"{len(scene.objects)} objects, {n_lm} with a lightmap -> {n_lm * 64} patches", flush=True)


def frame_ms(eng):
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        eng.render_rows_device(out.data_ptr(), 0, H)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.render_rows_device(out.data_ptr(), 0, H)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3, out.cpu().numpy()


eng = Engine(scene, W, H)
eng.set_march_cap(1024)
eng.input(0, 0, (0.0, 0.0, 0.0), 0)
lm = stress_lightmap_pattern()
for i, obj in enumerate(scene.objects):
    if not isinstance(obj, Box):
        eng.set_lightmap(i, lm)
ms_pat, img_pat = frame_ms(eng)
print(f"pattern lightmaps: {ms_pat:.2f} ms per frame", flush=True)
t0 = time.perf_counter()
eng.compute_lightmaps(4)
torch.cuda.synchronize()
print(f"compute_lightmaps(4): {time.perf_counter() - t0:.3f} s", flush=True)
ms_real, img_real = frame_ms(eng)
print(f"computed lightmaps: {ms_real:.2f} ms per frame; stats {eng.stats()}", flush=True)
lms = [eng.get_lightmap(i) for i, o in enumerate(scene.objects) if not isinstance(o, Box)]
a = np.stack(lms)
print(f"lightmap range {a.min():.4f} .. {a.max():.4f}, mean {a.mean():.4f}, finite {np.isfinite(a).all()}")
print(f"pixels that differ between the two frames: {(img_pat != img_real).any(axis=2).mean() * 100:.1f} %")
