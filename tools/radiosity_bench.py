"""Times Engine::compute_lightmaps on the GPU for a range of lightmap sizes (default scene)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import Engine, default_scene

for m in [int(x) for x in (sys.argv[1:] or ["8", "16", "32"])]:
    eng = Engine(default_scene(), 64, 64, map_size=m)
    eng.compute_lightmaps(4)  # warm-up, allocates
    torch.cuda.synchronize()
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng.compute_lightmaps(4)
        t.append(time.perf_counter() - t0)
    n = 8 * m * m
    pairs = n * (n - m * m)
    print(f"map_size {m:3d}: {n:6d} patches, {pairs/1e6:9.1f} M occlusion marches, best {min(t)*1e3:9.3f} ms -> {pairs/min(t)/1e6:9.1f} Mmarches/s")
    eng.close()
