"""Known answers for the CPU oracle, derived by hand from the reference's source.

The reference ships no tests or golden vectors, so these are the only pins the
oracle has besides review: small cases whose answer follows from the cited lines
with exact (or separately computed float32) arithmetic.
"""
import ctypes
import math

import numpy as np
import pytest

import oracle
from raymarching_amd.scene import (Material, PointLight, Scene, Sphere, Texture, XPlane, YPlane, ZPlane, default_scene,
                                   rgb)

f32 = np.float32
F3 = ctypes.c_float * 3


def _out3():
    return F3()


@pytest.fixture(scope="module")
def lib():
    return oracle.load()


@pytest.fixture(scope="module")
def eng():
    e = oracle.OracleEngine(default_scene())
    yield e
    e.close()


def test_rgb_macro(lib):
    # colour.rs:7-15: (c/255)^2 in f32
    out = _out3()
    lib.orc_kat_rgb(255, 255, 255, out)
    assert tuple(out) == (1.0, 1.0, 1.0)
    lib.orc_kat_rgb(214, 81, 0, out)
    exp = [(f32(c) / f32(255)) * (f32(c) / f32(255)) for c in (214, 81, 0)]
    assert [f32(v) for v in out] == exp
    assert rgb(214, 81, 0) == tuple(float(v) for v in exp)


def test_aces_film(lib):
    # colour.rs:33-44
    out = _out3()
    lib.orc_kat_aces(F3(0.0, 1.0, 0.18), out)
    assert out[0] == 0.0
    one = (f32(1) * (f32(2.51) * f32(1) + f32(0.03))) / (f32(1) * (f32(2.43) * f32(1) + f32(0.59)) + f32(0.14))
    assert f32(out[1]) == one
    x = f32(0.18)
    mid = (x * (f32(2.51) * x + f32(0.03))) / (x * (f32(2.43) * x + f32(0.59)) + f32(0.14))
    assert f32(out[2]) == mid
    assert abs(float(mid) - 0.2667) < 1e-3


def test_to_pixel_range(lib):
    # engine.rs:21-22: round half away from zero, clamp, saturating cast, NaN -> 0
    g = lib.orc_kat_to_pixel_range
    assert g(0.0) == 0 and g(1.0) == 255 and g(2.0) == 255 and g(-1.0) == 0
    assert g(float("nan")) == 0 and g(float("inf")) == 255 and g(float("-inf")) == 0
    assert g(0.4 / 255) == 0 and g(0.6 / 255) == 1
    assert g(127.5 / 255) == 128  # 255*x = 127.5 exactly in f32 -> rounds away from zero
    assert g(254.4 / 255) == 254


def test_texture_sample(lib, eng):
    # texture.rs:31-37 on Floor128.bmp with uscale=vscale=32: nearest, abs, wrap every 128 texels
    out = _out3()
    lib.orc_kat_texture_sample(eng.handle, 0, 0.0, 0.0, out)
    assert [f32(v) for v in out] == [f32(v) for v in rgb(233, 197, 143)]  # top-left texel
    tex = default_scene().objects[2].texture.rgb8

    def texel(u, v):
        o = _out3()
        lib.orc_kat_texture_sample(eng.handle, 0, u, v, o)
        return [f32(c) for c in o]

    def expect(x, y):
        return [f32(c) for c in rgb(*tex[y, x])]

    assert texel(1.0 / 32, 0.0) == expect(1, 0)
    assert texel(0.0, 5.0 / 32) == expect(0, 5)
    assert texel(-3.0 / 32, 0.0) == expect(3, 0)  # abs(): mirrored about 0
    assert texel(4.0 + 2.0 / 32, 0.0) == expect(2, 0)  # 128 texels = 4 world units
    assert texel(127.9 / 32, 127.2 / 32) == expect(127, 127)


def test_centre_pixel_ray(lib, eng):
    # engine.rs:79-84 at 800x600, x=400, y_inv=300: u=v=0, direction (0,0,1)
    pose = eng.pose()
    assert tuple(pose[:5]) == (0.0, 1.0, 0.0, 1.0, pose[4])
    zdepth = f32(1) / f32(math.tan(float((f32(90) * f32(0.5)) * (f32(math.pi) / f32(180)))))
    assert abs(float(pose[4]) - float(zdepth)) <= 1e-7
    out = _out3()
    lib.orc_kat_pixel_direction(800, 600, 400, 300, (ctypes.c_float * 5)(*pose[:5]), out)
    assert tuple(out) == (0.0, 0.0, 1.0)
    # from (0,0.5,-3.5) it passes both spheres and stops within 1e-3 of the back wall z=2 (object 6)
    pos = F3(0.0, 0.5, -3.5)
    hit = lib.orc_kat_march(eng.handle, pos, F3(0.0, 0.0, 1.0), -1, 0)
    assert hit == 6
    assert pos[0] == 0.0 and pos[1] == 0.5 and 2.0 - 1e-3 < pos[2] <= 2.0
    # first pixel of the first row looks up-left: u = 2*(0-400)/600, v = 2*(600-300)/600 = 1
    lib.orc_kat_pixel_direction(800, 600, 0, 0, (ctypes.c_float * 5)(*pose[:5]), out)
    u = f32(2) * (f32(0) - f32(0.5) * f32(800)) / f32(600)
    n = math.sqrt(float(u) ** 2 + 1 + float(pose[4]) ** 2)
    assert abs(out[0] - float(u) / n) < 1e-6 and abs(out[1] - 1 / n) < 1e-6


def _simple_scene():
    tex = Texture(np.full((2, 2, 3), 255, dtype=np.uint8), 1.0, 1.0)
    m = Material(0.25, 1.0, 0.5, 2.0, 0.0, 0.0)
    return Scene(
        objects=[
            Sphere((0.0, 0.0, 0.0), 1.0, m, (1.0, 1.0, 1.0)),
            XPlane(-4.0, 1.0, m, (1.0, 0.0, 0.0)),
            YPlane(-2.0, 1.0, m, (1.0, 1.0, 1.0), tex),
            ZPlane(8.0, -1.0, m, (0.0, 0.0, 1.0)),
        ],
        camera_position=(0.0, 0.0, -5.0),
        light=PointLight((0.0, 5.0, 0.0), 4.0),
    )


def test_sdf_and_normals(lib):
    e = oracle.OracleEngine(_simple_scene())
    sdf = lambda o, p: lib.orc_kat_sdf(e.handle, o, F3(*p))
    assert sdf(0, (3.0, 4.0, 0.0)) == 4.0  # objects.rs:334: |p-c| - r = 5 - 1
    assert sdf(0, (0.0, 0.0, 0.0)) == -1.0
    assert sdf(1, (1.0, 7.0, 7.0)) == 5.0  # objects.rs:295: dir*(x - X)
    assert sdf(2, (9.0, 0.5, 9.0)) == 2.5  # objects.rs:275
    assert sdf(3, (0.0, 0.0, 2.0)) == 6.0 and sdf(3, (0.0, 0.0, 9.0)) == -1.0  # objects.rs:315, dir=-1
    out = _out3()
    lib.orc_kat_normal(e.handle, 0, F3(2.0, 0.0, 0.0), out)  # objects.rs:135-141
    assert abs(out[0] - 1.0) < 1e-3 and abs(out[1]) < 1e-3 and abs(out[2]) < 1e-3
    lib.orc_kat_normal(e.handle, 3, F3(1.0, 1.0, 2.0), out)
    assert out[0] == 0.0 and out[1] == 0.0 and out[2] == -1.0
    # the commented-out four-point form (objects.rs:18-24, 142-149).  For the plane z = 2, dir = -1:
    # with s the distance at p and T = 0.0015, the corner distances are s+T (A), s-T (B), s+T (C),
    # s-T (D); the z component is -T(s+T) + T(s-T) - T(s+T) + T(s-T) = -4T^2, x and y cancel up to
    # the rounding of the four products: the normal is (0, 0, -1) to within 1e-6.
    e.set_tetrahedral_normals(True)
    lib.orc_kat_normal(e.handle, 3, F3(1.0, 1.0, 1.5), out)
    assert abs(out[0]) < 1e-6 and abs(out[1]) < 1e-6 and abs(out[2] + 1.0) < 1e-6
    # unit sphere at the origin, p = (2,0,0): corner distances differ from |p| - 1 by +-T/1 in first
    # order along x only, so the sum points along +x; the second-order terms are O(T^2 / |p|)
    lib.orc_kat_normal(e.handle, 0, F3(2.0, 0.0, 0.0), out)
    assert abs(out[0] - 1.0) < 1e-3 and abs(out[1]) < 2e-3 and abs(out[2]) < 2e-3
    e.close()


def test_march_hits_and_misses(lib):
    e = oracle.OracleEngine(_simple_scene())
    # straight at the sphere: first step is |p|-1 = 4, landing exactly on the surface
    pos = F3(0.0, 0.0, -5.0)
    assert lib.orc_kat_march(e.handle, pos, F3(0.0, 0.0, 1.0), -1, 0) == 0
    assert tuple(pos) == (0.0, 0.0, -1.0)
    # ignoring the sphere the same ray reaches the z=8 wall
    pos = F3(0.0, 0.0, -5.0)
    assert lib.orc_kat_march(e.handle, pos, F3(0.0, 0.0, 1.0), 0, 0) == 3
    assert 8.0 - 1e-3 < pos[2] <= 8.0
    # a ray that leaves through the open side travels MAX_MARCH_DISTANCE and misses (ray.rs:19,41)
    pos = F3(0.0, 3.0, 0.0)
    assert lib.orc_kat_march(e.handle, pos, F3(1.0, 0.0, 0.0), -1, 0) == -1
    assert pos[0] >= 50.0
    # a start point inside an object returns at once (negative distance < SMALL_DISTANCE)
    pos = F3(0.2, 0.0, 0.0)
    assert lib.orc_kat_march(e.handle, pos, F3(1.0, 0.0, 0.0), -1, 0) == 0
    assert abs(pos[0] - 0.2) < 1e-7
    # equal distances: strict `<` keeps the lower index (ray.rs:29)
    pos = F3(-3.9995, -1.9995, 0.0)
    assert lib.orc_kat_march(e.handle, pos, F3(0.0, 0.0, 1.0), 0, 0) == 1
    e.close()


def test_smooth_shadow_first_iteration(lib):
    e = oracle.OracleEngine(_simple_scene())
    sh = lambda p, d, ign, dist: lib.orc_kat_shadow(e.handle, F3(*p), F3(*d), ign, dist, 16.0)
    # unobstructed towards the light: k*d/0 = +inf -> clamp 1 -> smoothstep 1 -> shade stays 1 (ray.rs:109)
    assert sh((3.0, 0.0, 0.0), (0.0, 1.0, 0.0), 2, 5.0) == 1.0
    # start inside the sphere: d<0, k*d/0 = -inf -> clamp 0 -> shade 0, and the loop breaks (ray.rs:113)
    assert sh((0.2, 0.0, 0.0), (0.0, 1.0, 0.0), 2, 5.0) == 0.0
    # start exactly on the sphere surface (d = 0): 0/0 = NaN survives clamp, min() drops it -> shade 1
    assert sh((0.0, 0.0, -1.0), (0.0, 0.0, -1.0), 2, 3.0) == 1.0
    # a ray from the floor below the sphere straight up is fully blocked
    assert sh((0.0, -2.0, 0.0), (0.0, 1.0, 0.0), 2, 7.0) == 0.0
    e.close()


def test_phong(lib):
    # colour.rs:74-98 with n = l = +y, view = -y: R = l - 2n = -y, R.view = 1
    out = (ctypes.c_float * 2)()
    mat = (ctypes.c_float * 6)(0.25, 1.0, 0.5, 2.0, 0.0, 0.0)
    lib.orc_kat_phong(F3(0, 1, 0), F3(0, 1, 0), 2.0, 4.0, mat, F3(0, -1, 0), out)
    assert tuple(out) == (1.0, 0.5)  # intensity/d^2 = 1
    lib.orc_kat_phong(F3(0, 1, 0), F3(0, -1, 0), 2.0, 4.0, mat, F3(0, -1, 0), out)
    assert tuple(out) == (0.0, 0.0)  # light behind the surface: no diffuse, so no specular
    lib.orc_kat_phong(F3(0, 1, 0), F3(0, 1, 0), 1.0, 3.0, mat, F3(0.6, -0.8, 0.0), out)
    assert out[0] == 3.0 and abs(out[1] - 0.5 * 3.0 * 0.8**2) < 1e-6


def test_bilinear(lib):
    # colour.rs:47-71: corners reproduce, centre averages, outside extrapolates (no clamp)
    pts = [(2, 3, 1, 0, 0), (2, 4, 0, 1, 0), (3, 3, 0, 0, 1), (3, 4, 1, 1, 1)]
    flat = (ctypes.c_float * 20)(*[float(v) for p in pts for v in p])
    out = _out3()
    for x, y, *c in pts:
        lib.orc_kat_bilinear(float(x), float(y), flat, out)
        assert tuple(out) == tuple(float(v) for v in c)
    lib.orc_kat_bilinear(2.5, 3.5, flat, out)
    assert tuple(out) == (0.5, 0.5, 0.5)
    lib.orc_kat_bilinear(1.0, 3.0, flat, out)  # one cell to the left of x1
    assert tuple(out) == (2.0, 0.0, -1.0)
    # order of the four points does not matter (the function sorts them)
    shuffled = [pts[3], pts[0], pts[2], pts[1]]
    flat2 = (ctypes.c_float * 20)(*[float(v) for p in shuffled for v in p])
    lib.orc_kat_bilinear(2.25, 3.75, flat, out)
    a = tuple(out)
    lib.orc_kat_bilinear(2.25, 3.75, flat2, out)
    assert tuple(out) == a


def test_sample_positions(lib, eng):
    # objects.rs:278-284 etc.: plane luxels sit at u - 4 + 0.5; sphere luxels on (sin t, sin p, -cos t)
    out = _out3()
    lib.orc_kat_sample_pos(eng.handle, 2, 0, 7, out)
    assert tuple(out) == (-3.5, -2.0, 3.5)
    lib.orc_kat_sample_pos(eng.handle, 5, 3, 4, out)
    assert tuple(out) == (3.0, -0.5, 0.5)
    lib.orc_kat_sample_pos(eng.handle, 6, 1, 6, out)
    assert tuple(out) == (-2.5, 2.5, 2.0)
    lib.orc_kat_sample_pos(eng.handle, 0, 4, 4, out)  # theta = phi = 0: n = (0, 0, -1)
    c = default_scene().objects[0].position
    assert out[0] == f32(c[0]) and out[1] == f32(c[1]) and out[2] == f32(f32(c[2]) + f32(-1.0))


def test_sample_lightmap_branches(lib):
    """objects.rs:45-133 on a hand-filled 8x8 map of the z=8 wall (u = x+3.5, v = y+3.5)."""
    e = oracle.OracleEngine(_simple_scene())
    lm = np.zeros((8, 8, 3), dtype=np.float32)
    for u in range(8):
        for v in range(8):
            lm[u, v] = (u + 1, v + 1, 1.0)
    e.set_lightmap(3, lm)
    out = _out3()

    def sample(x, y):
        lib.orc_kat_sample_lightmap(e.handle, 3, F3(x, y, 8.0), out)
        return tuple(out)

    # interior: plain bilinear between (u0,v0)..(u0+1,v0+1); the map is linear, so exact
    assert sample(-1.0, 0.25) == (3.5, 4.75, 1.0)  # u=2.5, v=3.75
    # at the last row u1 is folded onto u0, then the coordinate is pushed out again: flat in u
    assert sample(3.75, -3.5) == (8.0, 1.0, 1.0)  # u=7.25 -> both u samples are row 7
    # below zero the floor is clamped to 0 but (u,v) keep their value: extrapolation
    assert sample(-4.0, -3.5) == (0.5, 1.0, 1.0)  # u=-0.5
    # an empty luxel to the right (s10 == 0) makes the right column reuse the left one
    lm2 = lm.copy()
    lm2[3, :] = 0.0
    e.set_lightmap(3, lm2)
    assert sample(-1.0, 0.25) == (3.0, 4.75, 1.0)
    # an empty luxel above (s01 == 0) does the same for v
    lm3 = lm.copy()
    lm3[:, 4] = 0.0
    e.set_lightmap(3, lm3)
    assert sample(-1.0, 0.25) == (3.5, 4.0, 1.0)
    # empty own luxel with filled right/up neighbours: s11 != 0, so nothing shifts; s00 contributes 0
    lm4 = lm.copy()
    lm4[2, 3] = 0.0
    e.set_lightmap(3, lm4)
    got = sample(-1.0, 0.25)
    w00 = 0.5 * 0.25
    assert got == pytest.approx((3.5 - 3 * w00, 4.75 - 4 * w00, 1.0 - w00), abs=1e-6)
    e.close()


def test_box_and_smooth_min_extensions(lib):
    """Extensions without live reference code: the box of the commented block objects.rs:370-380
    (centre and half extents as parameters) and the polynomial smooth minimum."""
    from raymarching_amd.scene import Box

    sc = _simple_scene()
    sc.objects.append(Box((0.0, 0.0, 0.0), (1.0, 2.0, 3.0), Material.basic(), (1.0, 1.0, 1.0)))
    e = oracle.OracleEngine(sc)
    sdf = lambda p: lib.orc_kat_sdf(e.handle, 4, F3(*p))
    assert sdf((3.0, 0.0, 0.0)) == 2.0  # outside one face: |max(q,0)| = 2, inner term 0
    assert sdf((0.0, 0.0, 0.0)) == -1.0  # inside: max(q) = -1 (nearest face)
    assert sdf((2.0, 3.0, 4.0)) == f32(np.sqrt(f32(3.0)))  # outside a corner
    assert sdf((-1.0, 2.0, 0.0)) == 0.0  # on an edge
    e.close()
    sm = lib.orc_kat_smooth_min
    assert sm(1.0, 1.0, 0.5) == 0.875  # h = 1: min - k/4
    assert sm(0.0, 2.0, 0.5) == 0.0  # farther apart than k: plain minimum
    assert sm(float("inf"), 3.0, 0.25) == 3.0  # the fold starts from +inf
    assert sm(1.0, 1.25, 0.5) == f32(1.0) - f32(0.5) * f32(0.5) * f32(0.5) * f32(0.25)
    # blending pulls the marched surface outwards between neighbours: a ray half way between the unit
    # sphere and the x=-4 wall (1.5 from each) passes without blending; with k = 8 the blend of the
    # two alone is 1.5 - k/4 < 0 there, so the march stops before it gets that far
    sc2 = _simple_scene()
    sc2.smooth_union = 0.0
    hard = oracle.OracleEngine(sc2)
    sc2.smooth_union = 8.0
    soft = oracle.OracleEngine(sc2)
    p_hard, p_soft = F3(-2.5, 0.0, -5.0), F3(-2.5, 0.0, -5.0)
    hit_h = lib.orc_kat_march(hard.handle, p_hard, F3(0.0, 0.0, 1.0), -1, 0)
    hit_s = lib.orc_kat_march(soft.handle, p_soft, F3(0.0, 0.0, 1.0), -1, 0)
    assert hit_h == 3 and p_hard[2] > 7.9  # reaches the z=8 wall
    assert hit_s in (0, 1, 2) and p_soft[2] < 0.5  # stopped by the blended surface
    hard.close()
    soft.close()


def test_refract_dead_code(lib):
    # vector.rs:40-48: total internal reflection returns the zero vector
    out = _out3()
    lib.orc_kat_refract(F3(0.8, -0.6, 0.0), F3(0.0, 1.0, 0.0), 2.5, out)
    assert tuple(out) == (0.0, 0.0, 0.0)
    lib.orc_kat_refract(F3(0.0, -1.0, 0.0), F3(0.0, 1.0, 0.0), 0.5, out)  # head-on: passes straight through
    assert tuple(out) == (0.0, -1.0, 0.0) or (abs(out[0]) < 1e-7 and abs(out[1] + 1.0) < 1e-6)


def test_internal_march_leaves_the_sphere(lib):
    # ray.rs:44-56: from just inside the near side, straight through the centre, the steps double up
    # to the centre and then halve towards the far side, where the march stalls on the surface
    pos = F3(0.0, 0.0, -0.997)
    lib.orc_kat_internal_march(F3(0.0, 0.0, 0.0), 1.0, pos, F3(0.0, 0.0, 1.0))
    assert pos[0] == 0.0 and pos[1] == 0.0 and abs(pos[2] - 1.0) < 1e-3
    # a ray that starts outside by more than SMALL_DISTANCE does not move
    pos = F3(0.0, 0.0, -1.5)
    lib.orc_kat_internal_march(F3(0.0, 0.0, 0.0), 1.0, pos, F3(0.0, 0.0, 1.0))
    assert tuple(pos) == (0.0, 0.0, -1.5)


def test_refraction_mode_of_the_oracle(lib):
    """engine.rs:196-228 switched on: only pixels that show a reflective object (the two spheres,
    directly or in a mirror image) change, and the added light never darkens a pixel's linear colour
    (checked on the encoded bytes, which are monotonic in it)."""
    sc = default_scene()
    ora = oracle.OracleEngine(sc)
    ora.compute_lightmaps(1)
    w, h = 160, 120
    base = ora.render(w, h, 0, 0, (0.0, 0.0, 0.0), 0)
    hit, _, _ = ora.trace_primary(w, h)
    ora.set_refraction(True)
    glass = ora.render(w, h, 0, 0, (0.0, 0.0, 0.0), 0)
    changed = (glass != base).any(axis=-1)
    assert changed.any() and not changed[hit > 1].any()
    assert (glass.astype(int) >= base.astype(int)).all()
    assert ora.refraction_depth_cuts() == 0
    ora.close()


def _fxaa_reference_pixel(img, x, y):
    """fxaa.txt:15-133 for one pixel in numpy float32, written independently of the oracle."""
    W = img.shape[1]

    def px(yy, xx):
        return [f32(img[yy, xx, c]) / f32(255) for c in range(3)]

    def luma(c):
        return f32(0.299) * c[0] + f32(0.587) * c[1] + f32(0.114) * c[2]

    nw, ne, sw, se, m = px(y - 1, x - 1), px(y - 1, x + 1), px(y + 1, x - 1), px(y + 1, x + 1), px(y, x)
    lnw, lne, lsw, lse, lm = luma(nw), luma(ne), luma(sw), luma(se), luma(m)
    lmin = min(lm, min(min(lnw, lne), min(lsw, lse)))
    lmax = max(lm, max(max(lnw, lne), max(lsw, lse)))
    d0 = -((lnw + lne) - (lsw + lse))
    d1 = (lnw + lsw) - (lne + lse)
    red = max((lnw + lne + lsw + lse) * (f32(0.25) * (f32(1) / f32(8))), f32(1) / f32(128))
    rcp = f32(1) / (min(abs(d0), abs(d1)) + red)
    d0 = min(f32(8), max(f32(-8), d0 * rcp))
    d1 = min(f32(8), max(f32(-8), d1 * rcp))
    ks = [f32(1) / f32(3) - f32(0.5), f32(2) / f32(2) - f32(0.5), f32(0) / f32(3) - f32(0.5), f32(3) / f32(3) - f32(0.5)]

    def off(v):  # f32::round then `as usize`
        r = math.floor(abs(float(v)) + 0.5) * (1 if v >= 0 else -1)
        return max(int(r), 0)

    taps = [px(y + off(d1 * k), x + off(d0 * k)) for k in ks]
    A = [f32(0.5) * (taps[0][c] + taps[1][c]) for c in range(3)]
    B = [A[c] * f32(0.5) + f32(0.25) * (taps[2][c] + taps[3][c]) for c in range(3)]
    lb = luma(B)
    cv = A if (lb < lmin or lb > lmax) else B
    return [int(max(0.0, min(255.0, math.floor(float(f32(255) * c))))) for c in cv]


def test_fxaa_against_independent_restatement():
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (20, 24, 4), dtype=np.uint8)
    img[:, :, 3] = 0
    img[6:14, 5:12, :3] = (250, 240, 10)  # a flat block with hard edges
    out = oracle.fxaa(img)
    # border of 4 pixels untouched (fxaa.txt:7-8), alpha untouched
    assert np.array_equal(out[:4], img[:4]) and np.array_equal(out[-4:], img[-4:])
    assert np.array_equal(out[:, :4], img[:, :4]) and np.array_equal(out[:, -4:], img[:, -4:])
    assert (out[:, :, 3] == 0).all()
    for y in range(4, 16):
        for x in range(4, 20):
            assert list(out[y, x, :3]) == _fxaa_reference_pixel(img, x, y), (x, y)
    # a constant image is a fixed point
    flat = np.full((16, 16, 4), 77, dtype=np.uint8)
    assert np.array_equal(oracle.fxaa(flat), flat)
    # the verbatim in-place reading differs from the out-of-place one as soon as neighbours change
    assert not np.array_equal(oracle.fxaa(img, in_place=True), out)


def test_lightmap_structure(oracle_lightmaps):
    """engine.rs:246-326 on the default scene: structural facts that follow from the geometry."""
    lm, em = oracle_lightmaps
    assert lm.shape == (5, 8, 8, 8, 3)
    assert not lm[0].any()  # no emissive object: direct lighting leaves every lightmap at zero (radiosity.rs:94)
    assert em[0].any()
    # floor luxels outside the room (|x| > 3) never receive light: u = 0 and u = 7 rows stay empty
    floor = lm[4][2]
    assert not floor[0].any() and not floor[7].any()
    assert floor[3, 3].all() and floor[4, 4].all()
    # lightmaps only grow from bounce to bounce (radiosity.rs:190 adds a non-negative term)
    for b in range(1, 5):
        assert (lm[b] >= lm[b - 1]).all()
    # the red wall bleeds red onto the floor next to it: r > g near x=-3
    assert floor[1, 4, 0] > floor[1, 4, 1]
    assert np.isfinite(lm).all()


def test_generic_plane_lightmap_parameterisation(lib):
    """The Plane's lightmap layout (an extension: the reference's is unimplemented!(), objects.rs:39-42).
    With an axis normal it is the layout of the matching axis plane - sample points, uv coordinates and
    with them every lightmap of the default scene - and for a tilted normal the tangents are unit
    vectors in the plane and sample_uv_from_pos inverts get_sample_pos."""
    from raymarching_amd.scene import Plane, Scene

    base = default_scene()
    objs = list(base.objects)
    matt, red = base.objects[4].material, base.objects[4].colour
    objs[4] = Plane((1.0, 0.0, 0.0), -3.0, matt, red)                            # XPlane(-3, +1)
    objs[3] = Plane((0.0, -1.0, 0.0), -4.0, base.objects[3].material, base.objects[3].colour)  # YPlane(4, -1): 4 - y
    objs[7] = Plane((0.0, 0.0, 1.0), -4.0, base.objects[7].material, base.objects[7].colour)   # ZPlane(-4, +1): z + 4
    sub = Scene(objects=objs, camera_position=base.camera_position, light=base.light)
    a, b = oracle.OracleEngine(base), oracle.OracleEngine(sub)
    pa, pb = _out3(), _out3()
    for obj in (3, 4, 7):
        for u in range(8):
            for v in range(8):
                lib.orc_kat_sample_pos(a.handle, obj, u, v, pa)
                lib.orc_kat_sample_pos(b.handle, obj, u, v, pb)
                assert tuple(pa) == tuple(pb), (obj, u, v)
    a.compute_lightmaps(4)
    b.compute_lightmaps(4)
    for i in range(8):
        assert np.array_equal(a.get_lightmap(i).view(np.uint32), b.get_lightmap(i).view(np.uint32)), i
    a.close()
    b.close()
    n = np.float32([-0.6, 0.0, -0.8])
    tilted = Scene(objects=[Plane(tuple(float(x) for x in n), -2.6, matt, red)], camera_position=base.camera_position, light=base.light)
    e = oracle.OracleEngine(tilted)
    pts = []
    for u in range(8):
        for v in range(8):
            lib.orc_kat_sample_pos(e.handle, 0, u, v, pa)
            pts.append(np.float32(tuple(pa)))
            assert abs(float(np.dot(pts[-1], n)) + 2.6) < 1e-5  # on the plane
    pts = np.array(pts).reshape(8, 8, 3)
    du, dv = pts[1, 0] - pts[0, 0], pts[0, 1] - pts[0, 0]
    assert abs(np.linalg.norm(du) - 1) < 1e-5 and abs(np.linalg.norm(dv) - 1) < 1e-5 and abs(float(np.dot(du, dv))) < 1e-5
    # a lightmap that encodes (u, v): sampling it at a sample point must return that point's own luxel
    lm = np.zeros((8, 8, 3), dtype=np.float32)
    for u in range(8):
        for v in range(8):
            lm[u, v] = (u + 1, v + 1, 1.0)
    e.set_lightmap(0, lm)
    out = _out3()
    for u in range(8):
        for v in range(8):
            lib.orc_kat_sample_lightmap(e.handle, 0, F3(*pts[u, v]), out)
            assert abs(out[0] - (u + 1)) < 1e-4 and abs(out[1] - (v + 1)) < 1e-4, (u, v, tuple(out))
    e.close()
