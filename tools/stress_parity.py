"""Stress scenes (250 primitives: smooth union with boxes, spheres only, hard minimum with boxes) at
512x288 from two poses against the oracle: primary march bit for bit, frames within the tolerance of
tests/conftest.py.  Larger than the cases in tests/test_scenes_gpu.py; run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_scenes_gpu as t
from conftest import frame_stats
from raymarching_amd.scene import stress_scene
for kw in (dict(boxes=True, smooth_union=0.08), dict(), dict(boxes=True)):
    for pose in ((0, 0, (0.0, 0.0, 0.0), 0), (240, 90, (0.4, 0.1, 0.8), 2)):
        t0 = time.time()
        got, ref, st, _ = t._compare(stress_scene(250, **kw), 512, 288, pose=pose, lightmaps=False)
        mx, exact = frame_stats(got, ref)
        print(kw, pose, f"max|d|={mx} exact={exact:.6%} caps={st} {time.time()-t0:.1f}s", flush=True)
