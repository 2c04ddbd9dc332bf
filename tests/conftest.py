import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# Tolerance of the frame comparison, as stated for this renderer: at most 2/255
# per 8-bit channel anywhere, and at least 99.9 % of pixels identical.
MAX_ABS_DIFF = 2
MIN_EXACT_FRACTION = 0.999
# Lightmaps: relative error at most 1e-4.
LIGHTMAP_RTOL = 1e-4


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def scene():
    from raymarching_amd.scene import default_scene

    return default_scene()


@pytest.fixture(scope="session")
def oracle_lightmaps(scene):
    """Oracle lightmaps after direct lighting and each of 4 bounces: (lm, em) [5][8][8][8][3]."""
    import oracle

    e = oracle.OracleEngine(scene)
    lm, em = e.compute_lightmaps(4, stages=True)
    e.close()
    return lm, em


def make_oracle(scene, lightmaps=None, map_size=8, counting=False):
    import oracle

    e = oracle.OracleEngine(scene, map_size=map_size, counting=counting)
    if lightmaps is not None:
        for i in range(e.n_objects):
            e.set_lightmap(i, lightmaps[i])
    return e


def frame_stats(a, b):
    """(max abs channel difference, fraction of pixels identical) of two [H][W][4] uint8 frames."""
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    exact = (d.reshape(-1, 4).max(axis=1) == 0).mean()
    return int(d.max()), float(exact)


# What the kernels actually deliver, asserted next to the contract above so that a regression inside
# the contract still fails: no channel off by more than 1, and no more than one pixel in 100 000 (or
# four pixels of a small frame) off at all.  Only powf and atan2f - the two calls whose last bit is the
# platform's - can move a pixel.
STRICT_MAX_ABS_DIFF = 1
STRICT_MAX_INEXACT_FRACTION = 1e-5
STRICT_MAX_INEXACT_PIXELS = 4


def assert_frames_match(gpu, ref, what="", strict=True):
    mx, exact = frame_stats(gpu, ref)
    assert mx <= MAX_ABS_DIFF, f"{what}: max |delta| {mx} > {MAX_ABS_DIFF}"
    assert exact >= MIN_EXACT_FRACTION, f"{what}: only {exact:.5%} of pixels exact"
    if strict:
        pixels = gpu.shape[0] * gpu.shape[1]
        inexact = round((1.0 - exact) * pixels)
        assert mx <= STRICT_MAX_ABS_DIFF, f"{what}: max |delta| {mx} > {STRICT_MAX_ABS_DIFF} (inside the contract, beyond what the kernels deliver)"
        assert inexact <= max(STRICT_MAX_INEXACT_PIXELS, STRICT_MAX_INEXACT_FRACTION * pixels), (
            f"{what}: {inexact} of {pixels} pixels differ (inside the contract, beyond what the kernels deliver)")
    return mx, exact
