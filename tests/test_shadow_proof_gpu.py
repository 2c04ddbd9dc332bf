"""Shadow rays that the frame kernel proves open are not marched (shadow_stays_open in
raymarching_amd/csrc/rm_march.inc; the loop it stands in for is ray.rs:87-118).  The proof must never
change a byte: random rooms - lights close to walls, close to spheres, outside the room; spheres from
pebbles to boulders; rooms scaled down and up; planes with directions and normals that are not unit
vectors (the proof then stands aside) - are rendered by an engine with the proof and by one without
(RM_SHADOW_PROOF=0 when the engine is created), and a sample of them by the oracle."""
import os

import numpy as np
import pytest

from conftest import assert_frames_match, make_oracle
from raymarching_amd.scene import (SOFT_GRAY, SOFT_GREEN, SOFT_RED, SOFT_YELLOW, WHITE, Material, Plane, PointLight, Scene,
                                   Sphere, XPlane, YPlane, ZPlane, default_scene)

pytestmark = pytest.mark.gpu

GLOSSY = Material(0.1, 1.0, 0.9, 32.0, 0.25, 0.0)
MATT = Material.basic()
SHINY = Material(0.08, 0.6, 0.8, 24.0, 0.0, 0.0)


def _random_room(seed):
    rng = np.random.default_rng(1000 + seed)
    base = default_scene()
    floor_tex = base.objects[2].texture
    scale = [1.0, 1.0, 1.0, 0.1, 5.0][seed % 5]
    lo, hi = np.array([-3.0, -2.0, -4.0]), np.array([3.0, 4.0, 2.0])
    planes = [YPlane(-2.0, 1.0, MATT, SOFT_GRAY, floor_tex), YPlane(4.0, -1.0, MATT, SOFT_GRAY, floor_tex), XPlane(-3.0, 1.0, MATT, SOFT_RED),
              XPlane(3.0, -1.0, MATT, SOFT_GREEN), ZPlane(2.0, -1.0, MATT, SOFT_GRAY), ZPlane(-4.0, 1.0, MATT, SOFT_GRAY)]
    if seed % 7 == 3:
        planes[2] = XPlane(-3.0, 0.5, MATT, SOFT_RED)  # a direction that is not +-1
    if seed % 7 == 5:
        planes.pop(1)  # no ceiling
    extra = []
    if seed % 4 == 1:
        n = np.array([0.0, 0.70710678, -0.70710678]) * (1.0 if seed % 8 == 1 else 1.7)  # unit normal, or one that is not
        extra.append(Plane(tuple(float(v) for v in n), -2.0 * float(np.linalg.norm(n)), MATT, SOFT_YELLOW))
    n_spheres = int(rng.integers(0, 4 if extra else 3)) if seed % 3 else 2
    spheres = []
    for i in range(n_spheres):
        r = float(rng.choice([0.05, 0.2, 0.6, 1.0, 1.5]))
        c = rng.uniform(lo + 0.2, hi - 0.2)
        spheres.append(Sphere(tuple(float(v) for v in c), r, [GLOSSY, MATT, SHINY][i % 3], [WHITE, SOFT_YELLOW, SOFT_GREEN][i % 3]))
    kind = seed % 6
    if kind == 0:
        light = rng.uniform(lo + 0.5, hi - 0.5)
    elif kind == 1:  # a hand's breadth from a wall
        light = rng.uniform(lo + 0.5, hi - 0.5)
        axis = int(rng.integers(0, 3))
        light[axis] = (lo if rng.random() < 0.5 else hi)[axis] * 0.995
    elif kind == 2 and spheres:  # just above a sphere's surface
        s = spheres[0]
        d = rng.normal(size=3)
        light = np.array(s.position) + d / np.linalg.norm(d) * (s.radius + 0.02)
    elif kind == 3:  # outside the room
        light = np.array([5.0, 1.0, -1.0])
    elif kind == 4:  # in a corner
        light = np.array([2.98, 3.97, 1.98])
    else:
        light = rng.uniform(lo + 0.05, hi - 0.05)

    def scaled(o):
        if isinstance(o, Sphere):
            return Sphere(tuple(scale * v for v in o.position), scale * o.radius, o.material, o.colour)
        if isinstance(o, XPlane):
            return XPlane(scale * o.x, o.dir, o.material, o.colour)
        if isinstance(o, YPlane):
            return YPlane(scale * o.y, o.dir, o.material, o.colour, o.texture)
        if isinstance(o, ZPlane):
            return ZPlane(scale * o.z, o.dir, o.material, o.colour)
        return Plane(o.normal, scale * o.distance, o.material, o.colour)

    cam = rng.uniform(lo + 0.6, hi - 0.6)
    return Scene(objects=[scaled(o) for o in spheres + planes + extra], camera_position=tuple(float(scale * v) for v in cam),
                 light=PointLight(tuple(float(scale * v) for v in light), float(rng.uniform(1.0, 6.0)) * scale * scale))


def _engine(scene, w, h, proof):
    from raymarching_amd import Engine

    before = os.environ.get("RM_SHADOW_PROOF")
    os.environ["RM_SHADOW_PROOF"] = "1" if proof else "0"
    try:
        e = Engine(scene, w, h)
    finally:
        if before is None:
            del os.environ["RM_SHADOW_PROOF"]
        else:
            os.environ["RM_SHADOW_PROOF"] = before
    return e


POSES = [(0, 0), (157, -40), (-310, 95), (628, 20)]  # mouse positions: the view turns about both axes


@pytest.mark.parametrize("block", range(6))
def test_a_proven_ray_changes_no_byte(block):
    import torch

    w, h = 640, 360
    for seed in range(block * 8, block * 8 + 8):
        scene = _random_room(seed)
        engines = [_engine(scene, w, h, proof) for proof in (True, False)]
        frames = []
        for e in engines:
            e.compute_lightmaps(1)
            out = []
            for mx, my in POSES:
                e.input(mx, my, (0, 0, 0), 0)
                buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
                e.render_rows_device(buf.data_ptr(), 0, h)
                e.synchronize()
                out.append(buf)
            frames.append(out)
            e.close()
        for k, (a, b) in enumerate(zip(*frames)):
            assert torch.equal(a, b), (seed, POSES[k], int((a != b).any(dim=2).sum()))


@pytest.mark.parametrize("seed", [0, 1, 2, 4, 9, 13, 16, 22, 27, 33])
def test_random_rooms_against_the_oracle(seed):
    scene = _random_room(seed)
    w, h = 200, 120
    e = _engine(scene, w, h, True)
    e.compute_lightmaps(1)
    ref = make_oracle(scene, [e.get_lightmap(i) for i in range(e.n_objects)])
    # (both sides keep a camera that input moves: the same inputs in the same order)
    for mx, my in POSES[:2]:
        assert_frames_match(e.render(None, mx, my, (0.0, 0.0, 0.0), 0), ref.render(w, h, mx, my, (0.0, 0.0, 0.0), 0), f"seed {seed} pose {mx},{my}")
    e.close()
