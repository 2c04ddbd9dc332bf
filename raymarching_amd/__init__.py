"""MI355X-native ray-marching hot path: scene types, Engine and multi-GPU helpers."""
from .scene import (  # noqa: F401
    Box,
    Material,
    Plane,
    PointLight,
    Scene,
    Sphere,
    Texture,
    XPlane,
    YPlane,
    ZPlane,
    construct_objects,
    default_scene,
    rgb,
)

__all__ = [
    "Box",
    "Engine",
    "Material",
    "Plane",
    "PointLight",
    "Scene",
    "Sphere",
    "Texture",
    "XPlane",
    "YPlane",
    "ZPlane",
    "construct_objects",
    "default_scene",
    "rgb",
]


def __getattr__(name):
    # Engine needs librm_hip.so; load it on first use so that importing the
    # scene types alone never touches the GPU library.
    if name == "Engine":
        from .engine import Engine

        return Engine
    raise AttributeError(name)
