"""The fast tier of the frame kernel (rm_engine_set_precision) against the CPU oracle.

The tier keeps a pixel's chain of sight rays strict and relaxes the arithmetic of what a hit's colour is
made of (fused multiply-adds, the hardware's one-ulp root, reciprocal, exp2 and log2, soft-shadow rays
marched with relaxed arithmetic).  Its frames are held to the tolerance this renderer states against the
reference, written in conftest.py: at most 2/255 per channel anywhere, at least 99.9 % of the pixels
identical.  (The strict tier's own, tighter bar - at most 1/255 in at most one pixel of 100 000 - is not
asked of it.)  The lightmaps are the strict precompute's in both tiers."""
import hashlib
import os

import numpy as np
import pytest

from conftest import assert_frames_match, frame_stats, make_oracle

pytestmark = pytest.mark.gpu

THREADS = min(os.cpu_count() or 1, 16)
POSES = [(0, 0, (0.0, 0.0, 0.0), 0), (150, -60, (0.0, 0.0, 0.0), 0), (-420, 90, (0.05, 0.0, 0.15), 2), (900, 200, (-0.15, 0.0, -0.05), -3)]


def _fast_engine(scene, w, h, lightmaps):
    from raymarching_amd import Engine

    eng = Engine(scene, w, h)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lightmaps[i])
    eng.set_precision("fast")
    return eng


@pytest.mark.parametrize("pose", range(len(POSES)))
def test_fast_frames_at_800x600(scene, oracle_lightmaps, pose):
    lm, _ = oracle_lightmaps
    mx_, my_, move, scroll = POSES[pose]
    ora = make_oracle(scene, lm[-1])
    ora.set_camera(scene.camera_position)
    ref = ora.render(800, 600, mouse_x=mx_, mouse_y=my_, rel_move=move, scroll=scroll)
    ora.close()
    eng = _fast_engine(scene, 800, 600, lm[-1])
    got = eng.render(None, mx_, my_, move, scroll)
    again = eng.render(None, mx_, my_, (0.0, 0.0, 0.0), scroll)  # (the camera has moved already)
    strict_stats = eng.stats()
    eng.set_precision("strict")
    strict = eng.render(None, mx_, my_, (0.0, 0.0, 0.0), scroll)
    eng.close()
    mx, exact = assert_frames_match(got, ref, f"fast tier, pose {pose}", strict=False)
    print(f"fast tier 800x600 pose {pose}: max|d| = {mx}, {round((1 - exact) * 480000)} of 480000 pixels differ")
    assert np.array_equal(got, again), "the fast tier must render the same bytes from run to run"
    assert (got[:, :, 3] == 0).all()
    assert strict_stats["depth_cap_hits"] == 0 and strict_stats["lightmap_oob"] == 0
    # switching back gives the strict frame again (the oracle's, bar the powf/atan2f pixels)
    assert_frames_match(strict, ref, f"strict again, pose {pose}")


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160)])
def test_fast_frames_at_the_benchmarked_sizes(scene, oracle_lightmaps, w, h):
    lm, _ = oracle_lightmaps
    ora = make_oracle(scene, lm[-1])
    ref = ora.render(w, h, threads=THREADS)
    ora.close()
    eng = _fast_engine(scene, w, h, lm[-1])
    got = eng.render()
    digest = hashlib.sha256(got.tobytes()).hexdigest()
    for _ in range(2):
        assert hashlib.sha256(eng.render().tobytes()).hexdigest() == digest, "bit-stable from run to run"
    eng.close()
    mx, exact = assert_frames_match(got, ref, f"fast tier {w}x{h}", strict=False)
    print(f"fast tier {w}x{h}: max|d| = {mx}, {round((1 - exact) * w * h)} of {w * h} pixels differ")


def test_fast_tier_on_a_large_scene_and_in_strips():
    """The stress scene family (spatial index, boxes, smooth unions) at a size the oracle renders in
    seconds, whole and as interleaved strips with slow tiles in quarters: the tier's tolerance against
    the oracle, and strips equal to the whole frame byte for byte."""
    import torch

    from raymarching_amd import Engine
    from raymarching_amd.scene import Box, stress_lightmap_pattern, stress_scene

    w, h = 480, 272
    scene = stress_scene(60, boxes=True, smooth_union=0.08)
    lit = [i for i, obj in enumerate(scene.objects) if not isinstance(obj, Box)]
    lm = stress_lightmap_pattern()
    ora = make_oracle(scene)
    eng = Engine(scene, w, h)
    for i in lit:
        eng.set_lightmap(i, lm)
        ora.set_lightmap(i, lm)
    ref = ora.render(w, h, threads=THREADS)
    ora.close()
    eng.set_precision("fast")
    whole = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    eng.input(0, 0, (0.0, 0.0, 0.0), 0)
    for _ in range(3):  # (the third launch is planned from the first two's tile times)
        eng.render_rows_device(whole.data_ptr(), 0, h)
    eng.synchronize()
    got = whole.cpu().numpy()
    mx, exact = assert_frames_match(got, ref, "fast tier, stress scene", strict=False)
    print(f"fast tier stress scene {w}x{h}: max|d| = {mx}, {round((1 - exact) * w * h)} of {w * h} pixels differ")
    strips = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    n = (h + 7) // 8
    for first in range(3):
        k = (n - first + 2) // 3
        part = torch.zeros((k * 8, w, 4), dtype=torch.uint8, device="cuda")
        eng.render_strips_device(part.data_ptr(), first, 3, k)
        eng.synchronize()
        for j in range(k):
            y0 = (first + 3 * j) * 8
            rows = min(8, h - y0)
            if rows > 0:
                strips[y0:y0 + rows] = part[8 * j:8 * j + rows]
    assert torch.equal(strips, whole)
    eng.close()
