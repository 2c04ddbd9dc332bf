"""GPU parity on scenes other than the default one: every scene tier of the march loops
(exact <2,2,0>, <2,2,0> with free slots, <4,4,2>, and the large path in global memory),
generic planes, and a 250-sphere stress scene."""
import numpy as np
import pytest

from conftest import assert_frames_match, make_oracle
from raymarching_amd.scene import (SOFT_GRAY, SOFT_GREEN, SOFT_RED, SOFT_YELLOW, WHITE, Box, Material, Plane,
                                   PointLight, Scene, Sphere, XPlane, YPlane, ZPlane, default_scene, stress_scene)

pytestmark = pytest.mark.gpu

W, H = 320, 240


def _room(extra, drop=(), smooth=0.0):
    base = default_scene()
    objs = [o for i, o in enumerate(base.objects) if i not in drop]
    return Scene(objects=list(extra) + objs, camera_position=base.camera_position, light=base.light, smooth_union=smooth)


def _scaled_room():
    base = default_scene()
    objs = list(base.objects)
    objs[4] = XPlane(-3.0, 0.5, MATT, SOFT_RED)
    objs[6] = ZPlane(2.0, -0.75, MATT, SOFT_GRAY)
    return Scene(objects=objs, camera_position=base.camera_position, light=base.light)


def _scatter(n, boxes, matt=False):
    """n small primitives in the room, clear of each other and of the two default spheres (surfaces
    inside one another make mirror chains hundreds of bounces long)."""
    rng = np.random.default_rng(11)
    placed = [((-1.2, -1.0, 0.1), 1.0), ((1.0, -1.0, -0.7), 1.0)]
    out = []
    while len(out) < n:
        c = tuple(float(v) for v in rng.uniform([-2.4, -1.4, -0.5], [2.4, 2.8, 1.6]))
        r = float(rng.uniform(0.12, 0.3))
        if any(np.linalg.norm(np.subtract(c, pc)) < r + pr + 0.05 for pc, pr in placed):
            continue
        placed.append((c, r))
        i = len(out)
        mat = MATT if matt else [MIRROR, GLOSSY, MATT][i % 3]
        if boxes and i % 2:
            h = r / np.sqrt(3.0)
            out.append(Box(c, (float(1.2 * h), float(0.8 * h), float(h)), mat, SOFT_RED))
        else:
            out.append(Sphere(c, r, mat, [WHITE, SOFT_GREEN, SOFT_YELLOW][i % 3]))
    return out


GLOSSY = Material(0.1, 1.0, 0.9, 32.0, 0.25, 0.0)
MIRROR = Material(0.05, 0.03, 0.2, 16.0, 1.0, 0.0)
MATT = Material.basic()

SCENES = {
    # one sphere missing: <2,2,0> with a free slot
    "tier_a_partial": lambda: _room([], drop=(1,)),
    # three more spheres and a slanted generic plane: <4,4,2>
    "tier_b": lambda: _room([Sphere((0.0, 1.5, 0.5), 0.6, GLOSSY, SOFT_RED), Sphere((2.0, 0.0, 1.0), 0.5, MATT, SOFT_GREEN),
                             Plane((0.0, 0.70710678, -0.70710678), -2.0, MATT, SOFT_YELLOW)]),
    # five spheres: beyond every by-value tier -> geometry in global memory
    "large_small": lambda: _room([Sphere((-2.0 + i, 2.0, 1.0), 0.4, [MIRROR, GLOSSY, MATT][i % 3], WHITE) for i in range(3)]),
    # boxes (hard minimum) and a smooth union of the default scene: the ordered path
    "boxes": lambda: _room([Box((0.0, 1.0, 0.5), (0.5, 0.3, 0.4), GLOSSY, SOFT_RED), Box((-2.0, -1.5, 1.0), (0.4, 0.5, 0.4), MATT, SOFT_GREEN)]),
    # (matt spheres: a reflective surface inside a blend zone re-hits itself and recurses hundreds deep)
    "smooth_union": lambda: _room([Sphere((-1.2, -1.0, 0.1), 1.0, MATT, WHITE), Sphere((1.0, -1.0, -0.7), 1.0, Material(0.1, 1.0, 0.9, 32.0, 0.0, 0.0), SOFT_YELLOW),
                                   Box((0.2, -1.6, 0.6), (0.5, 0.4, 0.4), MATT, SOFT_RED)], drop=(0, 1), smooth=0.3),
    # same counts as the default scene but plane directions other than +-1: must not take the
    # exact tier's one-instruction plane distance
    "scaled_dirs": lambda: _scaled_room(),
    # enough bounded primitives for the spatial index (16 or more), in an open scene: rays leave the
    # grid, reflections and shadow rays start on listed primitives
    "grid_sparse_open": lambda: _room(_scatter(16, boxes=False), drop=(3, 6)),
    # the same with boxes and a wide blend: the ordered smooth fold over cell lists
    "grid_smooth": lambda: _room(_scatter(20, boxes=True, matt=True), drop=(0, 1), smooth=0.15),
    # open scene: two planes removed, primary rays reach the sky (engine.rs:141)
    "open_sky": lambda: _room([], drop=(3, 6)),
}


def _compare(scene, w=W, h=H, pose=(0, 0, (0.0, 0.0, 0.0), 0), lightmaps=True, refraction=False):
    from raymarching_amd import Engine

    eng = Engine(scene, w, h)
    ora = make_oracle(scene)
    eng.set_refraction(refraction)
    ora.set_refraction(refraction)
    if not lightmaps:
        # no precompute: fill every lightmap with the same smooth, nowhere-zero pattern on both sides
        # (with empty maps the reference's resampling walks off the map and aborts, objects.rs:78-113)
        u, v = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
        lm = np.stack([0.10 + 0.02 * u, 0.12 + 0.01 * v, 0.08 + 0.01 * (u + v)], axis=-1).astype(np.float32)
        for i in range(eng.n_objects):
            if not isinstance(scene.objects[i], Box):
                eng.set_lightmap(i, lm)
                ora.set_lightmap(i, lm)
    if lightmaps:
        eng.compute_lightmaps(2)
        ora.compute_lightmaps(2)
        for i in range(eng.n_objects):
            if not isinstance(scene.objects[i], Box):  # (a box has no lightmap)
                assert np.array_equal(eng.get_lightmap(i).view(np.uint32), ora.get_lightmap(i).view(np.uint32)), i
    hit_g, steps_g, pos_g = eng.trace_primary()
    hit_o, steps_o, pos_o = ora.trace_primary(w, h)
    assert np.array_equal(hit_g, hit_o) and np.array_equal(steps_g, steps_o)
    assert np.array_equal(pos_g.view(np.uint32), pos_o.view(np.uint32))
    got = eng.render(None, *pose)
    ref = ora.render(w, h, *pose)
    st = eng.stats()
    st["oracle_depth_cuts"] = ora.refraction_depth_cuts()
    eng.close()
    return got, ref, st, hit_o


@pytest.mark.parametrize("name", sorted(SCENES))
def test_scene_tiers(name):
    scene = SCENES[name]()
    # scenes with many boxes are rendered with a fixed lightmap pattern instead of the precompute; every other
    # scene, the one with a generic Plane included, runs the precompute on both sides first
    # (the two small box scenes run it as well - boxes occlude and receive nothing; for the scenes with
    # many boxes the oracle's precompute would take minutes)
    has_box = any(isinstance(o, Box) for o in scene.objects)
    got, ref, st, hit = _compare(scene, lightmaps=not has_box or name in ("boxes", "smooth_union"))
    mx, exact = assert_frames_match(got, ref, name)
    print(f"{name}: max|d|={mx} exact={exact:.5%} caps={st}")
    assert st["depth_cap_hits"] == 0
    if name in ("open_sky", "grid_sparse_open"):
        # with walls missing many luxels stay empty and the reference's resampling leaves the map
        # (it would abort); both sides clamp such fetches the same way and count them
        assert (hit < 0).any(), "no primary ray reached the sky"
    else:
        assert st["lightmap_oob"] == 0


def _mirror_floor():
    base = default_scene()
    objs = list(base.objects)
    objs[2] = YPlane(-2.0, 1.0, GLOSSY, SOFT_GRAY, objs[2].texture)
    return Scene(objects=objs, camera_position=base.camera_position, light=base.light)


@pytest.mark.parametrize("name", ["tier_b", "large_small", "boxes", "grid_sparse_open", "mirror_floor"])
def test_refraction_in_scene_tiers(name):
    """The refraction block (engine.rs:196-228) over the other scene tiers: refracting boxes (their
    own distance walks the ray out, ray.rs:44-56), many small refracting spheres through the spatial
    index, and a refracting plane, which has no far side - the walk runs its 20 doubling steps into
    the half space and the ray goes on from there."""
    scene = _mirror_floor() if name == "mirror_floor" else SCENES[name]()
    has_plane = any(isinstance(o, (Plane, Box)) for o in scene.objects)
    got, ref, st, _ = _compare(scene, 240, 180, pose=(60, 30, (0.1, 0.0, 0.3), 0), lightmaps=not has_plane, refraction=True)
    mx, exact = assert_frames_match(got, ref, name)
    print(f"refraction, {name}: max|d|={mx} exact={exact:.5%} caps={st}")
    # both sides cut the tree 16 hits deep and count the cuts (a ray caught between a box and the
    # mirror next to it gets there in the "boxes" scene)
    assert st["depth_cap_hits"] == st["oracle_depth_cuts"]


def test_stress_scene_250_spheres():
    """256 primitives (250 spheres + the room): the large-scene path under heavy divergence."""
    scene = stress_scene(250)
    got, ref, st, _ = _compare(scene, 256, 192, lightmaps=False)
    mx, exact = assert_frames_match(got, ref, "stress")
    print(f"stress: max|d|={mx} exact={exact:.5%} caps={st}")
    # reflection chains between the many spheres run deeper than the 16-frame stack; what is cut
    # has been attenuated below one float32 ulp of the pixel by then, and the cuts are counted
    assert st["lightmap_oob"] == 0


def test_mixed_stress_scene_with_smooth_unions():
    """256 mixed primitives (spheres, boxes, the room) blended with a smooth union: the shape of the
    benchmark's stress configuration at a size the CPU oracle renders in seconds."""
    scene = stress_scene(250, boxes=True, smooth_union=0.08)
    got, ref, st, _ = _compare(scene, 192, 144, lightmaps=False)
    mx, exact = assert_frames_match(got, ref, "mixed stress")
    print(f"mixed stress: max|d|={mx} exact={exact:.5%} caps={st}")
    assert st["depth_cap_hits"] == 0


def test_march_cap_counts_and_binds(scene):
    """The optional iteration cap cuts marches and says so."""
    from raymarching_amd import Engine

    eng = Engine(scene, 160, 120)
    eng.set_march_cap(6)
    hit, steps, _ = eng.trace_primary()
    assert steps.max() <= 6 and (hit < 0).any()
    eng.render()
    assert eng.stats()["step_cap_hits"] > 0
    eng.set_march_cap(0)
    eng.reset_stats()
    hit2, steps2, _ = eng.trace_primary()
    assert (hit2 >= 0).all() and steps2.max() > 6
    eng.close()


@pytest.mark.parametrize("cap", [40, 100000])
def test_capped_march_matches_oracle(scene, cap):
    """The march with an iteration cap (the loop whose lanes leave one by one) against the oracle's
    uncapped march: wherever the oracle needs at most `cap` iterations the hit object, the iteration
    count and the end position are the oracle's bit for bit; everywhere else the march reports no hit
    after exactly `cap` iterations.  A frame rendered under a cap that never binds is the frame
    without a cap."""
    from raymarching_amd import Engine

    w, h = 400, 300
    ora = make_oracle(scene)
    hit_o, steps_o, pos_o = ora.trace_primary(w, h)
    ora.close()
    eng = Engine(scene, w, h)
    eng.compute_lightmaps(4)
    plain = eng.render()
    eng.set_march_cap(cap)
    hit_g, steps_g, pos_g = eng.trace_primary()
    within = steps_o <= cap
    assert within.any()
    assert np.array_equal(hit_g[within], hit_o[within])
    assert np.array_equal(steps_g[within], steps_o[within])
    assert np.array_equal(pos_g[within].view(np.uint32), pos_o[within].view(np.uint32))
    assert (hit_g[~within] == -1).all() and (steps_g[~within] == cap).all()
    if within.all():
        capped_frame = eng.render()
        assert np.array_equal(capped_frame, plain)
        assert eng.stats()["step_cap_hits"] == 0
    eng.close()


def test_generic_plane_takes_part_in_the_lightmap_precompute():
    """A generic Plane has a lightmap in the reference (objects.rs:216) but no way to address it
    (objects.rs:39-42); with the parameterisation defined in include/rm.h it joins the clouds.  A Plane
    with an axis normal in place of the red wall must reproduce the default scene's lightmaps bit for bit
    (same sample points, same distances), and a tilted Plane must agree with the oracle."""
    from raymarching_amd import Engine

    base = default_scene()
    ref_eng = Engine(base, 64, 64)
    ref_eng.compute_lightmaps(4)
    ref = np.stack([ref_eng.get_lightmap(i) for i in range(ref_eng.n_objects)])
    ref_eng.close()
    objs = list(base.objects)
    objs[4] = Plane((1.0, 0.0, 0.0), -3.0, MATT, SOFT_RED)  # XPlane(-3, +1): x + 3
    sub = Scene(objects=objs, camera_position=base.camera_position, light=base.light)
    eng = Engine(sub, W, H)
    eng.compute_lightmaps(4)
    got = np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])
    assert np.abs(got[4]).max() > 0
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    ora = make_oracle(sub)
    ora.compute_lightmaps(4)
    assert_frames_match(eng.render(), ora.render(W, H), "axis-normal Plane")
    eng.close()
    ora.close()
    # a tilted Plane cutting off a corner of the room, seen by the camera and lit by the lamp
    tilted = _room([Plane((-0.6, 0.0, -0.8), -2.6, MATT, SOFT_YELLOW)])
    eng = Engine(tilted, W, H)
    ora = make_oracle(tilted)
    eng.compute_lightmaps(4)
    ora.compute_lightmaps(4)
    lit = 0
    for i in range(eng.n_objects):
        a, b = eng.get_lightmap(i), ora.get_lightmap(i)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
        lit += int(i == 0 and np.abs(a).max() > 0)
    assert lit == 1, "the Plane's own lightmap stayed empty"
    hit = eng.trace_primary()[0]
    assert (hit == 0).any(), "the Plane is not in view"
    assert_frames_match(eng.render(), ora.render(W, H), "tilted Plane")
    eng.close()
    ora.close()
