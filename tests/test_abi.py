"""The C ABI library without a GPU: it loads, exports every symbol include/rm.h declares,
builds scenes on the host, and refuses loudly to create an engine when no device exists."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import gpu_available
from raymarching_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in rm.h but missing from librm_hip.so"
    # and the Python binding table covers the header one to one
    assert sorted(_lib.SIGNATURES) == names
    assert lib.rm_version() >= 1


def test_scene_construction_and_argument_errors():
    lib = _lib.load()
    scene = ctypes.c_void_p(lib.rm_scene_create())
    mat = _lib.Material(0.25, 1.0, 0.0, 4.0, 0.0, 0.0)
    white = _lib.f3((1, 1, 1))
    assert lib.rm_scene_add_sphere(scene, _lib.f3((0, 0, 0)), 1.0, mat, white) == 0
    assert lib.rm_scene_add_xplane(scene, -3.0, 1.0, mat, white) == 1
    # YPlane::new takes a texture: refused without one
    assert lib.rm_scene_add_yplane(scene, -2.0, 1.0, mat, white, 0) < 0
    assert b"texture" in lib.rm_last_error()
    px = np.zeros((2, 2, 3), dtype=np.uint8)
    tex = lib.rm_texture_create_rgb8(scene, px.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), 2, 2, 1.0, 1.0)
    assert tex == 0
    assert lib.rm_scene_add_yplane(scene, -2.0, 1.0, mat, white, tex) == 2
    assert lib.rm_scene_add_zplane(scene, 2.0, -1.0, mat, white) == 3
    assert lib.rm_scene_add_plane(scene, _lib.f3((0, 1, 0)), 0.0, mat, white) == 4
    assert lib.rm_texture_create_rgb8(scene, None, 2, 2, 1.0, 1.0) < 0
    assert lib.rm_scene_add_sphere(scene, None, 1.0, mat, white) < 0
    assert lib.rm_scene_set_light(scene, _lib.f3((0, 1, 0)), 2.0) == 0
    assert lib.rm_scene_set_camera(scene, None) != 0
    # bad engine arguments never reach the device
    assert not lib.rm_engine_create(scene, 0, 10, 8, 0)
    assert not lib.rm_engine_create(None, 10, 10, 8, 0)
    assert lib.rm_engine_render_rows_device(None, None, 0, 0, 0) != 0
    lib.rm_scene_destroy(scene)


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU error path")
def test_engine_creation_fails_loudly_without_gpu(scene):
    from raymarching_amd import Engine

    with pytest.raises(_lib.RMError, match="HIP device|hip"):
        Engine(scene, 64, 64)


def test_rust_binding_matches_the_header():
    """bindings/rust/rm-sys/src/lib.rs is generated from include/rm.h (tools/gen_rm_sys.py): the
    committed file is current, and it declares exactly the functions the ctypes table binds."""
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rm_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    text = open(os.path.join(root, "bindings", "rust", "rm-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"pub fn (rm_\w+)\(", text))
    from raymarching_amd import _lib

    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_a_missing_rccl_is_reported_not_crashed_on():
    """rm_group_unique_id and rm_group_create look RCCL up at run time; where it cannot be loaded they
    must come back with RM_ERR_UNSUPPORTED and a message (a fresh process: the lookup is cached)."""
    import subprocess
    import sys

    code = (
        "import ctypes, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from raymarching_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = (ctypes.c_uint8 * _lib.RM_GROUP_ID_BYTES)()\n"
        "rc1 = lib.rm_group_unique_id(buf)\n"
        "msg1 = lib.rm_last_error().decode()\n"
        "rc2 = lib.rm_group_unique_id(buf)  # the cached failure\n"
        "msg2 = lib.rm_last_error().decode()\n"
        "g = lib.rm_group_create(None, buf, 0, 1, 1)\n"
        "print(rc1, rc2, bool(g)); print(msg1); print(msg2)\n"
    )
    env = dict(os.environ, RM_RCCL_LIBRARY="/nonexistent/librccl-not-there.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "3 3 False", lines  # RM_ERR_UNSUPPORTED twice, no group
    assert "could not be loaded" in lines[1] and "librccl-not-there" in lines[1]
    assert lines[2] == lines[1]
