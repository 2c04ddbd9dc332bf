"""Algorithmic flop count of one frame from event counts.

Counting rule: each f32 add, subtract, multiply, compare, min, max, abs, floor
and round is 1 flop; each divide, reciprocal and square root is 1 flop; each
powf or atan2 call is 1 flop (they are also reported separately).  Vector
operations count per lane used (3).  Nothing is counted for address arithmetic,
integer work or loads.

The event counts come from the counting build of the CPU oracle; they are a
property of the workload (scene, pose, resolution), not of the kernel that
renders it.
"""

# sdf evaluations (objects.rs:334 / 275,295,315 / 261)
SPHERE_SDF = 10  # 3 sub, 3 mul, 2 add, sqrt, sub
AXIS_PLANE_SDF = 2  # sub, mul
PLANE_SDF = 6  # 3 mul, 2 add, sub
# extensions of the reference's object set (the stress scene): the box of objects.rs:370-380 - 3 sub,
# 3 abs, 3 sub, 3 max, length (3 mul, 2 add, sqrt), max, max, min, add - and, with a smooth union, the
# fold of one distance into the running value where the two are at least k apart (difference, magnitude,
# compare besides the minimum that is counted anyway; the eight operations of an actual blend are not counted)
BOX_SDF = 22
SMOOTH_FOLD_EXTRA = 3
# Ray::march per iteration besides the sdfs: one compare per object, hit test,
# loop test, travelled add, position update (3 mul + 3 add)      (ray.rs:19-39)
MARCH_STEP = 9
# smooth_shadow_march per iteration besides the sdfs and their compares: k*d, /t,
# clamp (2), smoothstep (6), min, travelled add, position update (6), 2 exit tests
SHADOW_STEP = 20
# calculate_normal: six offset positions (18 adds), 3 differences, normalise (10)
NORMAL_EXTRA = 31
RAY_SETUP = 28  # engine.rs:79-84: u, v, rot, normalise
SHADE_CONST = 61  # engine.rs:163-194 + phong_ds without powf
LIGHTMAP_SAMPLE = 80  # plane uv, floors, four zero tests, bilinear
SPHERE_UV_EXTRA = 19  # normalise + the arithmetic around atan2
TEXTURE_SAMPLE = 12
REFLECTION = 47  # engine.rs:231-241 besides the recursive ray
TONEMAP = 42  # exposure, ACES, sqrt, to_pixel_range for 3 channels


def frame_flops(c, smooth_union=False):
    """c: dict of oracle counters for one frame.  Returns the flop total."""
    march_box, shadow_box = c.get("march_box_sdf", 0), c.get("shadow_box_sdf", 0)
    march_sdf = c["march_sphere_sdf"] + c["march_plane_sdf"] + c["march_gplane_sdf"] + march_box
    shadow_sdf = c["shadow_sphere_sdf"] + c["shadow_plane_sdf"] + c["shadow_gplane_sdf"] + shadow_box
    total = BOX_SDF * (march_box + shadow_box)
    if smooth_union:
        total += SMOOTH_FOLD_EXTRA * (march_sdf + shadow_sdf)
    total += SPHERE_SDF * (c["march_sphere_sdf"] + c["shadow_sphere_sdf"])
    total += AXIS_PLANE_SDF * (c["march_plane_sdf"] + c["shadow_plane_sdf"])
    total += PLANE_SDF * (c["march_gplane_sdf"] + c["shadow_gplane_sdf"])
    total += march_sdf + MARCH_STEP * c["march_steps"]
    total += shadow_sdf + SHADOW_STEP * c["shadow_steps"]
    total += (6 * SPHERE_SDF + NORMAL_EXTRA) * c["normals_sphere"]
    total += (6 * AXIS_PLANE_SDF + NORMAL_EXTRA) * c["normals_plane"]
    total += (6 * PLANE_SDF + NORMAL_EXTRA) * c["normals_gplane"]
    total += RAY_SETUP * c["pixels"] + TONEMAP * c["pixels"]
    total += SHADE_CONST * c["shade_calls"]
    total += LIGHTMAP_SAMPLE * c["lightmap_samples"] + SPHERE_UV_EXTRA * c["atan2_calls"]
    total += TEXTURE_SAMPLE * c["texture_samples"]
    total += REFLECTION * c["reflection_rays"]
    total += c["powf_calls"] + c["atan2_calls"]
    return int(total)
