"""Lightmap precompute (4 bounces, default scene) at MAP_SIZE 8, 32 and 64 for several builds of the
library in one process: usage radiosity_ab.py lib.so [lib.so ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarching_amd import _lib, default_scene

for rnd in range(2):
    for path in sys.argv[1:]:
        _lib._lib = None
        _lib.LIB_PATH = os.path.abspath(path)
        from raymarching_amd.engine import Engine

        for m in (8, 32, 64):
            eng = Engine(default_scene(), 64, 64, map_size=m)
            eng.compute_lightmaps(4)
            eng.synchronize()
            ts = []
            for _ in range(5 if m < 64 else 3):
                t0 = time.perf_counter()
                eng.compute_lightmaps(4)
                eng.synchronize()
                ts.append(time.perf_counter() - t0)
            print(os.path.basename(path), m, f"{min(ts) * 1e3:.3f} ms", flush=True)
            eng.close()
