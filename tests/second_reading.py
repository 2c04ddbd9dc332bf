"""A second, independent reading of the reference's per-ray and per-patch arithmetic, in numpy float32.

Written from the Rust text of /root/reference/src (ray.rs, objects.rs, colour.rs, texture.rs,
radiosity.rs, engine.rs:129-244), not from oracle/rm_oracle.cpp: the C++ oracle and the HIP kernels
share one author's reading of that text, and the reference holds no golden data that could catch a
shared misreading.  This file is a different restatement - array-at-a-time over many rays with masks
instead of a loop per ray - that tests/test_second_reading.py holds against the oracle.

Every array is float32 and every operation is one IEEE binary32 operation in the reference's order
(numpy does not fuse).  `powf` and `atan2` come from numpy here and from glibc in the oracle, so
values that pass through them are compared with a tolerance; everything else bit for bit.
"""
import numpy as np

from raymarching_amd.scene import Plane, Sphere, XPlane, YPlane, ZPlane

F = np.float32
MAX_MARCH_DISTANCE = F(50.0)  # engine.rs:12
SMALL_DISTANCE = F(0.001)  # engine.rs:13
MAX_SHAD_IT = 64  # engine.rs:14
STEP_SIZE = F(0.0001)  # objects.rs:13
PI = F(np.pi)
TAU = F(2 * np.pi)


def sky_colour():  # engine.rs:15 with colour.rs:7-15
    return np.array([(F(c) / F(255.0)) * (F(c) / F(255.0)) for c in (135, 206, 235)], dtype=F)


# ---- vector.rs ------------------------------------------------------------------

def dot(a, b):  # vector.rs:32
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def mag_sqd(a):  # vector.rs:24: lanes x, y, z, w summed in order; w is 0
    return ((a[..., 0] * a[..., 0] + a[..., 1] * a[..., 1]) + a[..., 2] * a[..., 2]) + F(0.0)


def mag(a):  # vector.rs:22
    return np.sqrt(mag_sqd(a))


def normalized(a):  # vector.rs:27-30: the reciprocal, then a multiplication
    return a * (F(1.0) / mag(a))[..., None]


def reflect(v, n):  # vector.rs:38
    return v - n * (F(2.0) * dot(n, v))[..., None]


def vec(x, y, z):
    return np.stack([np.asarray(x, dtype=F), np.asarray(y, dtype=F), np.asarray(z, dtype=F)], axis=-1)


# ---- objects.rs -------------------------------------------------------------------

def sdf(obj, p):
    """objects.rs:261,275,295,315,334 for points p [N][3]."""
    if isinstance(obj, Sphere):
        return mag(p - np.asarray(obj.position, dtype=F)) - F(obj.radius)
    if isinstance(obj, XPlane):
        return F(obj.dir) * (p[..., 0] - F(obj.x))
    if isinstance(obj, YPlane):
        return F(obj.dir) * (p[..., 1] - F(obj.y))
    if isinstance(obj, ZPlane):
        return F(obj.dir) * (p[..., 2] - F(obj.z))
    if isinstance(obj, Plane):
        return dot(p, np.broadcast_to(np.asarray(obj.normal, dtype=F), p.shape)) - F(obj.distance)
    raise TypeError(obj)


def calculate_normal(obj, p):
    """objects.rs:135-141: central differences of the object's own distance, then normalized()."""
    zero = F(0.0)
    steps = (vec(STEP_SIZE, zero, zero), vec(zero, STEP_SIZE, zero), vec(zero, zero, STEP_SIZE))
    g = [sdf(obj, p + s) - sdf(obj, p - s) for s in steps]
    return normalized(vec(*g))


def sample_uv_from_pos(obj, pos, m):
    """objects.rs:285-290, 305-310, 325-330, 353-358."""
    half = F(m) / F(2.0)
    if isinstance(obj, YPlane):
        return pos[..., 0] + half - F(0.5), pos[..., 2] + half - F(0.5)
    if isinstance(obj, XPlane):
        return pos[..., 1] + half - F(0.5), pos[..., 2] + half - F(0.5)
    if isinstance(obj, ZPlane):
        return pos[..., 0] + half - F(0.5), pos[..., 1] + half - F(0.5)
    n = normalized(pos - np.asarray(obj.position, dtype=F))
    u = F(0.5) + np.arctan2(n[..., 0], -n[..., 2]) / TAU
    v = F(0.5) + (n[..., 1] / F(2.0))
    return u * F(m), v * F(m)


def get_sample_pos(obj, u, v, m):
    """objects.rs:278-284, 298-304, 318-324, 345-352 for one luxel."""
    a = F(u) - (F(m) / F(2.0)) + F(0.5)
    b = F(v) - (F(m) / F(2.0)) + F(0.5)
    if isinstance(obj, YPlane):
        return vec(a, F(obj.y), b)
    if isinstance(obj, XPlane):
        return vec(F(obj.x), a, b)
    if isinstance(obj, ZPlane):
        return vec(a, b, F(obj.z))
    theta = (((F(u) / F(m)) * F(2.0)) - F(1.0)) * PI
    phi = ((F(v) / F(m)) - F(0.5)) * PI
    n = vec(np.sin(theta), np.sin(phi), -np.cos(theta))
    return np.asarray(obj.position, dtype=F) + n * F(obj.radius)


def texture_sample(tex, u, v):
    """texture.rs:31-37 on texels squared as texture.rs:18-21 / colour.rs:7-15 do."""
    x = np.fmod(np.floor(np.abs(u * F(tex.uscale))), F(tex.width)).astype(np.int64)
    y = np.fmod(np.floor(np.abs(v * F(tex.vscale))), F(tex.height)).astype(np.int64)
    px = tex.rgb8.reshape(-1, 3)[x + y * tex.width].astype(F)
    return (px / F(255.0)) * (px / F(255.0))


def colour(obj, pos, m):
    """objects.rs:260, 267-274, 294, 314, 336."""
    if isinstance(obj, YPlane) and not (F(obj.y) > F(0.0)):
        u, v = sample_uv_from_pos(obj, pos, m)
        return texture_sample(obj.texture, u, v)
    return np.broadcast_to(np.asarray(obj.colour, dtype=F), pos.shape).copy()


def bilinear_interpolation(x, y, points):
    """colour.rs:47-71 for one sample: points = [(x, y, colour)] * 4, sorted as the text sorts them."""
    points = sorted(points, key=lambda p: p[1])  # stable, as sort_by
    points = sorted(points, key=lambda p: p[0])
    (x1, y1, q11), (_x1, y2, q12), (x2, _y1, q21), (_x2, _y2, q22) = points
    assert x1 == _x1 and x2 == _x2 and y1 == _y1 and y2 == _y2, "Points do not form a rectangle"
    acc = ((q11 * (x2 - x)) * (y2 - y) + (q21 * (x - x1)) * (y2 - y)) + (q12 * (x2 - x)) * (y - y1)
    acc = acc + (q22 * (x - x1)) * (y - y1)
    return acc / ((x2 - x1) * (y2 - y1))


def sample_lightmap(obj, lightmap, pos):
    """objects.rs:45-133 for one position; lightmap [M][M][3] indexed [u][v].  Returns None where the
    reference would index outside the map (a panic there)."""
    m = lightmap.shape[0]
    u, v = sample_uv_from_pos(obj, pos, m)
    u, v = F(u), F(v)
    u0, v0 = int(max(np.floor(u), F(0.0))), int(max(np.floor(v), F(0.0)))
    u1, v1 = u0 + 1, v0 + 1
    u0 = u0 - 1 if u0 >= m else u0
    v0 = v0 - 1 if v0 >= m else v0
    u1 = u0 if u1 >= m else u1
    v1 = v0 if v1 >= m else v1

    def fetch(a, b):
        if not (0 <= a < m and 0 <= b < m):
            raise IndexError
        return lightmap[a, b]

    def dark(c):
        return mag_sqd(c) == F(0.0)

    try:
        s00, s01, s10, s11 = fetch(u0, v0), fetch(u0, v1), fetch(u1, v0), fetch(u1, v1)
        resample = False
        if dark(s00):
            resample = True
            if dark(s10):
                v0, v1 = v0 + 1, v1 + 1
            elif dark(s11):
                v0, v1 = v0 - 1, v1 - 1
            if dark(s01):
                u0, u1 = u0 + 1, u1 + 1
            elif dark(s11):
                u0, u1 = u0 - 1, u1 - 1
        else:
            if dark(s10):
                resample = True
                u1 -= 1
            if dark(s01):
                resample = True
                v1 -= 1
        if resample:
            s00, s01, s10, s11 = fetch(u0, v0), fetch(u0, v1), fetch(u1, v0), fetch(u1, v1)
    except IndexError:
        return None
    if u0 == u1:
        u1 += 1
    if v0 == v1:
        v1 += 1
    pts = [(F(u0), F(v0), s00), (F(u0), F(v1), s01), (F(u1), F(v0), s10), (F(u1), F(v1), s11)]
    return bilinear_interpolation(u, v, pts)


# ---- ray.rs -------------------------------------------------------------------------

def march(objects, pos, direction, ignore=None):
    """Ray::march (ray.rs:15-42) for N rays at once.  ignore: None or an int array (-1 = none).
    Returns (hit index or -1, final position, iterations)."""
    pos = pos.astype(F).copy()
    n = pos.shape[0]
    travelled = np.zeros(n, dtype=F)
    hit = np.full(n, -1, dtype=np.int64)
    steps = np.zeros(n, dtype=np.int64)
    running = np.ones(n, dtype=bool)
    ignore = np.full(n, -1, dtype=np.int64) if ignore is None else np.asarray(ignore, dtype=np.int64)
    with np.errstate(all="ignore"):
        while True:
            running &= travelled < MAX_MARCH_DISTANCE
            idx = np.nonzero(running)[0]
            if idx.size == 0:
                break
            p = pos[idx]
            distance = np.full(idx.size, np.inf, dtype=F)
            closest = np.full(idx.size, -1, dtype=np.int64)
            for i, obj in enumerate(objects):
                d = sdf(obj, p)
                better = (d < distance) & (ignore[idx] != i)
                distance = np.where(better, d, distance)
                closest = np.where(better, i, closest)
            steps[idx] += 1
            done = distance < SMALL_DISTANCE
            hit[idx[done]] = closest[done]
            running[idx[done]] = False
            go = idx[~done]
            travelled[go] = travelled[go] + distance[~done]
            pos[go] = pos[go] + direction[go] * distance[~done][:, None]
    return hit, pos, steps


def smooth_shadow_march(objects, pos, direction, ignore, light_dist, k):
    """Ray::smooth_shadow_march (ray.rs:87-118) for N rays at once."""
    pos = pos.astype(F).copy()
    n = pos.shape[0]
    travelled = np.zeros(n, dtype=F)
    shade = np.ones(n, dtype=F)
    running = np.ones(n, dtype=bool)
    ignore = np.asarray(ignore, dtype=np.int64)
    with np.errstate(all="ignore"):
        for _ in range(MAX_SHAD_IT):
            idx = np.nonzero(running)[0]
            if idx.size == 0:
                break
            p = pos[idx]
            distance = np.full(idx.size, np.inf, dtype=F)
            for i, obj in enumerate(objects):
                d = sdf(obj, p)
                distance = np.where((d < distance) & (ignore[idx] != i), d, distance)
            x = np.clip(F(k) * distance / travelled[idx], F(0.0), F(1.0))  # f32::clamp keeps a NaN
            smooth = F(3.0) * (x * x) - F(2.0) * (x * (x * x))            # powi(2), powi(3)
            shade[idx] = np.fmin(shade[idx], smooth)                        # f32::min drops a NaN
            travelled[idx] = travelled[idx] + distance
            pos[idx] = p + direction[idx] * distance[:, None]
            running[idx] = ~((distance < SMALL_DISTANCE) | (travelled[idx] > light_dist[idx]))
    return np.clip(shade, F(0.0), F(1.0))


# ---- colour.rs / engine.rs ------------------------------------------------------------

def phong_ds(n, to_light, dist, intensity, mat, view):
    """colour.rs:74-98."""
    r = reflect(to_light, n)
    li = F(intensity) / (dist * dist)
    diffuse = F(mat.diffuse) * li * np.fmax(dot(to_light, n), F(0.0))
    with np.errstate(all="ignore"):
        spec = F(mat.specular) * li * np.power(np.fmax(dot(r, view), F(0.0)), F(mat.shininess))
    return diffuse, np.where(diffuse > F(0.0), spec, F(0.0)).astype(F)


def shade_object(scene, lightmaps, obj_index, pos, direction, depth=0):
    """Engine::shade_object (engine.rs:147-244) for N hits on ONE object, reflections followed through
    cast_sight_ray to any depth.  Returns colours [N][3] and a validity mask (False where
    sample_lightmap would have left the map)."""
    objects = scene.objects
    obj = objects[obj_index]
    mat = obj.material
    n_rays = pos.shape[0]
    m = lightmaps[obj_index].shape[0]
    object_colour = colour(obj, pos, m)
    n = calculate_normal(obj, pos)
    ambient = np.zeros((n_rays, 3), dtype=F)
    valid = np.ones(n_rays, dtype=bool)
    for j in range(n_rays):
        a = sample_lightmap(obj, lightmaps[obj_index], pos[j])  # every object of these scenes has a lightmap (engine.rs:159)
        if a is None:
            valid[j] = False
        else:
            ambient[j] = a
    with np.errstate(all="ignore"):
        low = mag_sqd(ambient) < F(mat.ambient) * F(mat.ambient)
        ambient = np.where(low[:, None], normalized(ambient) * F(mat.ambient), ambient)
    ambient = ambient * object_colour
    light = np.asarray(scene.light.position, dtype=F)
    to_light = light - pos
    dist = mag(to_light)
    to_light = to_light / dist[:, None]
    diffuse, specular = phong_ds(n, to_light, dist, scene.light.intensity, mat, direction)
    shade = smooth_shadow_march(objects, pos, to_light, np.full(n_rays, obj_index), dist, F(16.0))
    final = ambient + object_colour * (shade * (diffuse + specular))[:, None]
    if F(mat.reflectivity) > F(1e-3):
        x = np.clip(F(1.0) - dot(n, -direction), F(0.0), F(1.0))
        sq = x * x
        fresnel = x * (sq * sq)  # powi(5)
        rv = reflect(direction, n)
        rc, rvalid = cast_sight_ray(scene, lightmaps, pos + (rv * F(3.0)) * SMALL_DISTANCE, rv, depth + 1)
        final = final + (rc * object_colour) * np.clip(fresnel + F(mat.reflectivity), F(0.0), F(1.0))[:, None]
        valid &= rvalid
    return final.astype(F), valid


def cast_sight_ray(scene, lightmaps, pos, direction, depth=0):
    """Engine::cast_sight_ray (engine.rs:129-145) for N rays."""
    assert depth < 64, "runaway recursion"
    hit, end, _ = march(scene.objects, pos, direction)
    out = np.tile(sky_colour(), (pos.shape[0], 1))
    valid = np.ones(pos.shape[0], dtype=bool)
    for i in range(len(scene.objects)):
        sel = np.nonzero(hit == i)[0]
        if sel.size:
            out[sel], valid[sel] = shade_object(scene, lightmaps, i, end[sel], direction[sel], depth)
    return out, valid


# ---- radiosity.rs -----------------------------------------------------------------------

def clouds(scene, m):
    """engine.rs:259-281: position, colour and normal of every sample point, [n_obj][M][M][3]."""
    n = len(scene.objects)
    pos = np.zeros((n, m, m, 3), dtype=F)
    for c, obj in enumerate(scene.objects):
        for x in range(m):
            for y in range(m):
                pos[c, x, y] = get_sample_pos(obj, x, y, m)
    col = np.stack([colour(obj, pos[c].reshape(-1, 3), m).reshape(m, m, 3) for c, obj in enumerate(scene.objects)])
    nrm = np.stack([calculate_normal(obj, pos[c].reshape(-1, 3)).reshape(m, m, 3) for c, obj in enumerate(scene.objects)])
    return pos, col.astype(F), nrm.astype(F)


def compute_direct_lighting(scene, pos, col):
    """radiosity.rs:52-101 -> (emissive maps, lightmaps), [n_obj][M][M][3]."""
    n, m = pos.shape[0], pos.shape[1]
    light = np.asarray(scene.light.position, dtype=F)
    emissive = np.zeros_like(pos)
    lightmaps = np.zeros_like(pos)
    for c, obj in enumerate(scene.objects):
        origin = pos[c].reshape(-1, 3)
        to_light = light - origin
        dist = mag(to_light)
        to_light = to_light / dist[:, None]
        _, end, _ = march(scene.objects, origin, to_light, np.full(origin.shape[0], c))
        blocked = mag(end - origin) < (dist + F(3.0) * SMALL_DISTANCE)
        nn = calculate_normal(obj, origin)
        diffuse = np.fmax(dot(nn, to_light), F(0.0)) * F(scene.light.intensity) / (dist * dist)
        colr = col[c].reshape(-1, 3)
        em = colr * (diffuse + F(obj.material.emissive))[:, None]
        lm = colr * F(obj.material.emissive)
        emissive[c] = np.where(blocked[:, None], F(0.0), em).reshape(m, m, 3)
        lightmaps[c] = np.where(blocked[:, None], F(0.0), lm).reshape(m, m, 3)
    return emissive, lightmaps


def compute_object_radiosity(scene, c, pos, col, nrm, emissive, lightmap_c):
    """radiosity.rs:159-196 with compute_patch_radiosity (103-157) inlined, for cloud object c (every
    object of these scenes has a lightmap, so cloud index = engine index).  All (receiver, source) pairs
    are marched at once; the sum over the sources then runs in the reference's order per receiver."""
    n, m = pos.shape[0], pos.shape[1]
    recv = pos[c].reshape(-1, 3)
    recv_n = nrm[c].reshape(-1, 3)
    r = recv.shape[0]
    others = [j for j in range(n) if j != c]  # radiosity.rs:112
    src = np.concatenate([pos[j].reshape(-1, 3) for j in others])
    src_n = np.concatenate([nrm[j].reshape(-1, 3) for j in others])
    src_e = np.concatenate([emissive[j].reshape(-1, 3) for j in others])
    s = src.shape[0]
    with np.errstate(all="ignore"):
        to_light = src[None, :, :] - recv[:, None, :]  # [r][s][3]
        dist = mag(to_light)
        to_light = to_light / dist[..., None]
        back = -to_light
        start = src[None, :, :] + back * (F(3.0) * SMALL_DISTANCE)  # radiosity.rs:131: 3.0 * SMALL * -v
        hit, end, _ = march(scene.objects, start.reshape(-1, 3), back.reshape(-1, 3))
        hit, end = hit.reshape(r, s), end.reshape(r, s, 3)
        blocked = (hit >= 0) & (hit != c) & (mag_sqd(end - recv[:, None, :]) > F(0.1) * F(0.1))
        att = dot(np.broadcast_to(recv_n[:, None, :], to_light.shape), to_light) * -dot(np.broadcast_to(src_n[None, :, :], to_light.shape), to_light)
        diffuse = np.fmin(np.fmax(att, F(0.0)) / (dist * dist), F(2.0))
        incident = np.zeros((r, 3), dtype=F)
        for k in range(s):  # ordered sum (f32 addition does not associate)
            term = src_e[k][None, :] * diffuse[:, k][:, None]
            incident = np.where(blocked[:, k][:, None], incident, incident + term)
        incident = incident / PI
    new_lightmap = lightmap_c.reshape(-1, 3) + incident
    new_emissive = incident * col[c].reshape(-1, 3)
    return new_lightmap.reshape(m, m, 3).astype(F), new_emissive.reshape(m, m, 3).astype(F)


# ---- engine.rs:21-32, 63-126 and colour.rs:33-44: from a pixel to its four bytes --------------

def rot(v, sx, cx, sy, cy):  # engine.rs:24-32
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    y, z = y * cy - z * sy, y * sy + z * cy
    x, z = x * cx - z * sx, x * sx + z * cx
    return vec(x, y, z)


def pixel_directions(width, height, xs, y_invs, sx, cx, sy, cy, zdepth):
    """engine.rs:77-86 for pixels (xs, y_invs); the four trigonometric values and zdepth come from the caller."""
    y = (height - np.asarray(y_invs)).astype(F)  # y = HEIGHT - y_inv, in usize, then `as f32`
    u = F(2.0) * (np.asarray(xs).astype(F) - (F(0.5) * F(width))) / F(height)
    v = F(2.0) * (y - (F(0.5) * F(height))) / F(height)
    return normalized(rot(vec(u, v, np.full(u.shape, zdepth, dtype=F)), F(sx), F(cx), F(sy), F(cy)))


def aces_film(c):  # colour.rs:33-44
    a, b, cc, d, e = F(2.51), F(0.03), F(2.43), F(0.59), F(0.14)
    return (c * (a * c + b)) / (c * (cc * c + d) + e)


def to_pixel_range(i):  # engine.rs:21-22: round half away from zero, clamp, saturating cast (NaN -> 0)
    with np.errstate(all="ignore"):
        x = F(255.0) * i
        t = np.trunc(x)
        r = t + np.where(np.abs(x - t) >= F(0.5), np.sign(x), F(0.0)).astype(F)
        r = np.clip(r, F(0.0), F(255.0))
        return np.where(np.isnan(r), 0, r).astype(np.uint8)


def pack_pixels(linear, exposure=F(1.0)):
    """engine.rs:109-125: ACESFilm(colour * exposure).sqrt(), bytes b0 = z, b1 = y, b2 = x, b3 = 0."""
    with np.errstate(all="ignore"):
        srgb = np.sqrt(aces_film(linear * exposure))
    out = np.zeros(linear.shape[:-1] + (4,), dtype=np.uint8)
    out[..., 0] = to_pixel_range(srgb[..., 2])
    out[..., 1] = to_pixel_range(srgb[..., 1])
    out[..., 2] = to_pixel_range(srgb[..., 0])
    return out
