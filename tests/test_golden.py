"""The CPU oracle against the committed fixtures in tests/golden (made by make_golden.py).

The fixtures are outputs of the oracle itself (the reference has none): they pin it
against accidental change and against a different libm on another machine."""
import hashlib
import json
import os

import numpy as np
import pytest
from PIL import Image

import oracle
from conftest import make_oracle
from raymarching_amd.flops import frame_flops

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

POSES = {
    "pose0_800x600": (800, 600, 0, 0, (0.0, 0.0, 0.0), 0),
    "pose1_400x300": (400, 300, 150, -60, (0.0, 0.0, 0.0), 0),
    "pose2_400x300": (400, 300, -420, 90, (0.05, 0.0, 0.15), 2),
}


def load_bgra(name):
    rgb = np.asarray(Image.open(os.path.join(GOLDEN, name)).convert("RGB"))
    out = np.zeros(rgb.shape[:2] + (4,), dtype=np.uint8)
    out[:, :, :3] = rgb[:, :, ::-1]
    return out


def test_lightmap_stages(oracle_lightmaps):
    lm, em = oracle_lightmaps
    ref = np.load(os.path.join(GOLDEN, "lightmaps_default.npz"))
    assert np.array_equal(lm.view(np.uint32), ref["lightmaps"].view(np.uint32))
    assert np.array_equal(em.view(np.uint32), ref["emissive"].view(np.uint32))


@pytest.mark.parametrize("name", sorted(POSES))
def test_frames(scene, oracle_lightmaps, name):
    w, h, mx, my, mv, sc = POSES[name]
    ora = make_oracle(scene, oracle_lightmaps[0][-1])
    frame = ora.render(w, h, mx, my, mv, sc)
    ref = load_bgra(f"frame_{name}.png")
    diff = np.abs(frame.astype(np.int16) - ref.astype(np.int16))
    # identical on the machine that made the fixture; elsewhere libm's powf/atan2f may move a last bit
    assert diff.max() <= 1 and (diff.reshape(-1, 4).max(axis=1) == 0).mean() >= 0.9999
    if name == "pose1_400x300":
        assert np.array_equal(oracle.fxaa(ref), load_bgra("fxaa_pose1_400x300.png"))


def test_primary_trace(scene):
    ora = make_oracle(scene)
    hit, steps, pos = ora.trace_primary(800, 600)
    ref = np.load(os.path.join(GOLDEN, "trace_pose0_800x600.npz"))
    assert np.array_equal(hit, ref["hit"].astype(np.int32))
    assert np.array_equal(steps, ref["steps"].astype(np.int32))
    assert hashlib.sha256(np.ascontiguousarray(pos).tobytes()).hexdigest() == str(ref["pos_sha256"])
    assert (hit >= 0).all()  # closed room: no primary ray reaches the sky


def test_counters_and_flop_model(scene, oracle_lightmaps):
    with open(os.path.join(GOLDEN, "counters.json")) as f:
        ref = json.load(f)
    ora = make_oracle(scene, oracle_lightmaps[0][-1], counting=True)
    ora.reset_counters()
    ora.render(800, 600)
    got = ora.counters()
    want = ref["frame_pose0_800x600"]
    for key in ("pixels", "march_calls", "march_steps", "shadow_steps", "reflection_rays", "shade_calls",
                "normals_sphere", "normals_plane", "texture_samples", "max_depth", "max_march_steps"):
        assert got[key] == want[key], key
    assert got["lightmap_oob"] == 0 and got["sky_hits"] == 0
    # one march per primary ray and per reflection ray; one shadow march per shaded hit
    assert got["march_calls"] == got["pixels"] + got["reflection_rays"]
    assert got["shadow_calls"] == got["shade_calls"]
    # default scene: 2 spheres + 6 axis planes per march step (49 flops per step as counted)
    assert got["march_sphere_sdf"] == 2 * got["march_steps"] and got["march_plane_sdf"] == 6 * got["march_steps"]
    flops = frame_flops(want)
    assert 1000 < flops / want["pixels"] < 3000
    for key in ("frame_pose0_1920x1080", "frame_pose0_3840x2160"):
        assert ref[key]["lightmap_oob"] == 0 and ref[key]["max_depth"] < 16
