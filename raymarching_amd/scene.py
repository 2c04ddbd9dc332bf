"""Scene description types with the reference's names and fields.

These are plain descriptions (no arithmetic on the render path): the reference's
`Sphere`, `XPlane`, `YPlane`, `ZPlane`, `Plane`, `PointLight`, `Material`,
`Texture` and the `rgb!` macro, plus the default scene of main.rs.  An `Engine`
(engine.py) flattens them through the C ABI onto the GPU.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .bmp import load_bmp_rgb8

Vec3 = Tuple[float, float, float]

_f32 = np.float32


def rgb(r, g, b) -> Vec3:
    """colour.rs:7-15  rgb![r,g,b] = ((c/255)^2) per channel, evaluated in f32."""
    out = []
    for c in (r, g, b):
        q = _f32(c) / _f32(255.0)
        out.append(float(q * q))
    return tuple(out)


# colour.rs:17-22
WHITE = rgb(255, 255, 255)
SOFT_RED = rgb(214, 81, 81)
SOFT_GREEN = rgb(81, 214, 81)
SOFT_GRAY = rgb(214, 214, 214)
SOFT_YELLOW = rgb(230, 230, 127)
# engine.rs:15
SKY_COLOUR = rgb(135, 206, 235)


@dataclass(frozen=True)
class Material:
    """material.rs:2-9"""

    ambient: float = 0.25
    diffuse: float = 1.0
    specular: float = 0.0
    shininess: float = 4.0
    reflectivity: float = 0.0
    emissive: float = 0.0

    @staticmethod
    def basic() -> "Material":
        """material.rs:12-21"""
        return Material()

    def as_tuple(self):
        return (self.ambient, self.diffuse, self.specular, self.shininess, self.reflectivity, self.emissive)


@dataclass
class Texture:
    """texture.rs:4-29: decoded RGB8 rows (top-down) plus the two scale factors."""

    rgb8: np.ndarray  # [h][w][3] uint8
    uscale: float
    vscale: float

    @staticmethod
    def new(path: str, uscale: float, vscale: float) -> "Texture":
        """texture.rs:13"""
        pixels, _, _ = load_bmp_rgb8(path)
        return Texture(pixels, uscale, vscale)

    @property
    def width(self):
        return int(self.rgb8.shape[1])

    @property
    def height(self):
        return int(self.rgb8.shape[0])


@dataclass
class Sphere:
    """objects.rs:158-165"""

    position: Vec3
    radius: float
    material: Material
    colour: Vec3


@dataclass
class Plane:
    """objects.rs:167-175"""

    normal: Vec3
    distance: float
    material: Material
    colour: Vec3


@dataclass
class XPlane:
    """objects.rs:177-184, XPlane::new objects.rs:225"""

    x: float
    dir: float
    material: Material
    colour: Vec3


@dataclass
class YPlane:
    """objects.rs:186-194, YPlane::new objects.rs:236 (takes a texture)"""

    y: float
    dir: float
    material: Material
    colour: Vec3
    texture: Texture


@dataclass
class ZPlane:
    """objects.rs:196-203, ZPlane::new objects.rs:248"""

    z: float
    dir: float
    material: Material
    colour: Vec3


@dataclass
class Box:
    """Extension: the commented-out box of objects.rs:370-380 with centre and half extents."""

    position: Vec3
    half_extent: Vec3
    material: Material
    colour: Vec3


@dataclass
class PointLight:
    """objects.rs:205-209"""

    position: Vec3
    intensity: float


@dataclass
class Scene:
    """What main.rs:64-71 puts into `Engine { objects, camera_position, light }`."""

    objects: List[object]
    camera_position: Vec3
    light: PointLight
    smooth_union: float = 0.0  # extension: > 0 blends object distances (0 = the reference's hard minimum)


def construct_objects(texture_path: str) -> List[object]:
    """main.rs:154-193"""
    wood_tex = Texture.new(texture_path, 32.0, 32.0)
    basic = Material.basic()
    return [
        Sphere((-1.2, -1.0, 0.1), 1.0, Material(0.05, 0.03, 0.2, 16.0, 1.0, 0.0), WHITE),
        Sphere((1.0, -1.0, -0.7), 1.0, Material(0.1, 1.0, 0.9, 32.0, 0.25, 0.0), SOFT_YELLOW),
        YPlane(-2.0, 1.0, basic, SOFT_GRAY, wood_tex),
        YPlane(4.0, -1.0, basic, SOFT_GRAY, wood_tex),
        XPlane(-3.0, 1.0, basic, SOFT_RED),
        XPlane(3.0, -1.0, basic, SOFT_GREEN),
        ZPlane(2.0, -1.0, basic, SOFT_GRAY),
        ZPlane(-4.0, 1.0, basic, SOFT_GRAY),
    ]


def default_scene(texture_path: Optional[str] = None) -> Scene:
    """main.rs:62-71: the scene, camera and light the reference starts with."""
    if texture_path is None:
        texture_path = default_texture_path()
    return Scene(
        objects=construct_objects(texture_path),
        camera_position=(0.0, 0.5, -3.5),
        light=PointLight((2.0, -1.0, 1.5), 3.5),
    )


def default_texture_path() -> str:
    import os

    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "textures", "Floor128.bmp")


def build_scene(backend, scene: Scene):
    """Feed `scene` into a backend exposing texture/add_*/set_light/set_camera.

    Both the GPU engine's C ABI wrapper and the test oracle implement this small
    protocol, so one description drives both.
    """
    tex_ids = {}
    for obj in scene.objects:
        if isinstance(obj, YPlane):
            key = id(obj.texture)
            if key not in tex_ids:
                tex_ids[key] = backend.texture(obj.texture)
    for obj in scene.objects:
        mat = obj.material.as_tuple()
        if isinstance(obj, Sphere):
            backend.add_sphere(obj.position, obj.radius, mat, obj.colour)
        elif isinstance(obj, XPlane):
            backend.add_xplane(obj.x, obj.dir, mat, obj.colour)
        elif isinstance(obj, YPlane):
            backend.add_yplane(obj.y, obj.dir, mat, obj.colour, tex_ids[id(obj.texture)])
        elif isinstance(obj, ZPlane):
            backend.add_zplane(obj.z, obj.dir, mat, obj.colour)
        elif isinstance(obj, Plane):
            backend.add_plane(obj.normal, obj.distance, mat, obj.colour)
        elif isinstance(obj, Box):
            backend.add_box(obj.position, obj.half_extent, mat, obj.colour)
        else:
            raise TypeError(f"unknown object type {type(obj).__name__}")
    backend.set_light(scene.light.position, scene.light.intensity)
    backend.set_camera(scene.camera_position)
    if scene.smooth_union:
        backend.set_smooth_union(scene.smooth_union)


def stress_scene(n_spheres: int = 250, seed: int = 0xC0FFEE, texture_path: Optional[str] = None, boxes: bool = False,
                 smooth_union: float = 0.0) -> Scene:
    """A seed-fixed synthetic scene: the default room (six axis planes, textured floor) filled with
    `n_spheres` small primitives of mixed materials (every third one a box if `boxes`), optionally
    blended with a smooth union.  It exercises the large-scene and ordered paths and deep
    divergence; it has no counterpart in the reference."""
    if texture_path is None:
        texture_path = default_texture_path()
    rng = np.random.Generator(np.random.PCG64(seed))
    wood = Texture.new(texture_path, 32.0, 32.0)
    basic = Material.basic()
    objects: List[object] = []
    palette = [SOFT_RED, SOFT_GREEN, SOFT_GRAY, SOFT_YELLOW]
    placed = []  # (centre, radius): spheres keep a gap, so no surface point lies inside another sphere
    attempts = 0
    while len(placed) < n_spheres and attempts < 200000:
        attempts += 1
        c = rng.uniform([-2.6, -1.6, -1.0], [2.6, 3.4, 1.8])
        r = float(rng.uniform(0.05, 0.4))
        if all(np.linalg.norm(c - pc) > r + pr + 0.02 for pc, pr in placed):
            placed.append((c, r))
    # A reflection ray starts 3e-3 off the surface (engine.rs:237).  Inside the blend zone of a smooth
    # union that is still "on" the surface, so the ray hits its own object again and the recursion runs
    # hundreds deep.  In a blended scene only primitives that stand clear of every blend zone keep their
    # reflective material: everything else (primitives and walls) at least 2k + 0.01 away from them, so
    # that the ordered fold returns their own distance unchanged all around their surface (a running
    # value is never more than k below what was folded before).
    centres = np.array([c for c, _ in placed])
    radii = np.array([r for _, r in placed])
    if len(placed):
        gaps = np.linalg.norm(centres[:, None] - centres[None], axis=-1) - radii[:, None] - radii[None]
        np.fill_diagonal(gaps, np.inf)
        walls = np.minimum.reduce([centres[:, 0] + 3.0, 3.0 - centres[:, 0], centres[:, 1] + 2.0, 4.0 - centres[:, 1],
                                   centres[:, 2] + 4.0, 2.0 - centres[:, 2]]) - radii
        clear = np.minimum(gaps.min(axis=1), walls) >= 2.0 * smooth_union + 0.01
    for i, (c, r) in enumerate(placed):
        kind = i % 4
        is_box = boxes and i % 3 == 2
        # (and no box among them: a reflection ray that grazes a flat face starts less than SMALL_DISTANCE
        # above it, hits the face again 3e-3 further on, and so crawls along the whole face, one level of
        # recursion per step - a hundred levels on these boxes)
        refl = 1.0 if (smooth_union <= 0.0 or (clear[i] and not is_box)) else 0.0
        if kind == 0:
            mat = Material(0.05, 0.3, 0.2, 16.0, 0.5 * refl, 0.0)  # half mirror
        elif kind == 1:
            mat = Material(0.1, 1.0, 0.9, 32.0, 0.25 * refl, 0.0)  # glossy
        else:
            mat = Material(0.25, 1.0, 0.0, 4.0, 0.0, 0.0)  # matt
        centre = (float(c[0]), float(c[1]), float(c[2]))
        if boxes and i % 3 == 2:
            # inscribed in the sphere that was placed, so the gaps between primitives stay
            h = r / np.sqrt(3.0)
            objects.append(Box(centre, (float(h * 1.2), float(h * 0.9), float(h * 0.8)), mat, palette[i % len(palette)]))
        else:
            objects.append(Sphere(centre, r, mat, palette[i % len(palette)]))
    objects += [
        YPlane(-2.0, 1.0, basic, SOFT_GRAY, wood),
        YPlane(4.0, -1.0, basic, SOFT_GRAY, wood),
        XPlane(-3.0, 1.0, basic, SOFT_RED),
        XPlane(3.0, -1.0, basic, SOFT_GREEN),
        ZPlane(2.0, -1.0, basic, SOFT_GRAY),
        ZPlane(-4.0, 1.0, basic, SOFT_GRAY),
    ]
    return Scene(objects=objects, camera_position=(0.0, 0.5, -3.5), light=PointLight((2.0, -1.0, 1.5), 3.5),
                 smooth_union=smooth_union)


def stress_lightmap_pattern(map_size: int = 8) -> np.ndarray:
    """A fixed smooth [M][M][3] pattern that stands in for the lightmap precompute in scenes whose
    boxes have no lightmap parameterisation (bench.py --workload stress and its parity test)."""
    u, v = np.meshgrid(np.arange(map_size), np.arange(map_size), indexing="ij")
    return np.stack([0.10 + 0.02 * u, 0.12 + 0.01 * v, 0.08 + 0.01 * (u + v)], axis=-1).astype(np.float32)
