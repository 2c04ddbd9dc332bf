"""Edge cases of the frame path: empty and sky-only scenes, frames smaller than one wave tile,
ragged sizes, extreme poses, and a camera that is not a number."""
import numpy as np
import pytest

from conftest import assert_frames_match, make_oracle
from raymarching_amd.scene import SKY_COLOUR, Material, PointLight, Scene, Sphere, default_scene

pytestmark = pytest.mark.gpu


def _pair(scene, w, h, lightmap_value=0.1):
    from raymarching_amd import Engine

    eng = Engine(scene, w, h)
    ora = make_oracle(scene)
    lm = np.full((8, 8, 3), lightmap_value, dtype=np.float32)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm)
        ora.set_lightmap(i, lm)
    return eng, ora


def test_empty_scene_is_sky():
    """No objects at all: every march runs out at MAX_MARCH_DISTANCE and returns the sky colour
    (engine.rs:141)."""
    scene = Scene(objects=[], camera_position=(0.0, 0.5, -3.5), light=PointLight((2.0, -1.0, 1.5), 3.5))
    eng, ora = _pair(scene, 40, 24)
    got, ref = eng.render(None), ora.render(40, 24)
    assert np.array_equal(got, ref)
    assert len(np.unique(got.reshape(-1, 4), axis=0)) == 1
    assert got[0, 0, 3] == 0 and tuple(got[0, 0, :3]) != (0, 0, 0)  # B,G,R of the tone-mapped sky, A = 0
    assert SKY_COLOUR[2] > SKY_COLOUR[0]  # a blue sky: byte 0 (B) is the largest
    assert got[0, 0, 0] > got[0, 0, 2]
    eng.close()


def test_single_sphere_against_the_sky():
    """One mirror sphere and nothing else: reflection rays leave the scene."""
    scene = Scene(objects=[Sphere((0.0, 0.5, 0.0), 1.0, Material(0.05, 0.03, 0.2, 16.0, 1.0, 0.0), (1.0, 1.0, 1.0))],
                  camera_position=(0.0, 0.5, -3.5), light=PointLight((2.0, -1.0, 1.5), 3.5))
    eng, ora = _pair(scene, 96, 72)
    assert_frames_match(eng.render(None), ora.render(96, 72), "single sphere")
    hit, _, _ = eng.trace_primary()
    assert (hit == 0).any() and (hit < 0).any()
    eng.close()


@pytest.mark.parametrize("w,h", [(1, 1), (3, 2), (8, 8), (9, 7), (61, 13), (130, 5)])
def test_small_and_ragged_frames(w, h, oracle_lightmaps):
    """Frames smaller than a tile, and sizes that leave partial tiles on both edges."""
    from raymarching_amd import Engine

    scene = default_scene()
    lm, _ = oracle_lightmaps
    eng = Engine(scene, w, h)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[-1][i])
    ora = make_oracle(scene, lm[-1])
    got, ref = eng.render(None), ora.render(w, h)
    assert got.shape == (h, w, 4)
    if w * h >= 64:
        assert_frames_match(got, ref, f"{w}x{h}")
    else:  # too few pixels for a fraction: every one within the per-channel bound
        assert np.abs(got.astype(int) - ref.astype(int)).max() <= 2
    eng.close()


@pytest.mark.parametrize("pose", [
    (5000, -3000, (0.0, 0.0, 0.0), 0),     # many turns of the mouse
    (0, 0, (0.0, 0.0, 0.0), 17),           # fov 5 degrees
    (0, 0, (0.0, 0.0, 0.0), -17),          # fov 175 degrees
    (0, 942, (0.0, 0.0, 0.0), 0),          # looking almost straight up
    (300, 100, (0.0, 0.0, 9.0), 0),        # camera moved through the back wall, outside the room
])
def test_extreme_poses(pose, oracle_lightmaps):
    from raymarching_amd import Engine

    scene = default_scene()
    lm, _ = oracle_lightmaps
    w, h = 160, 120
    eng = Engine(scene, w, h)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[-1][i])
    ora = make_oracle(scene, lm[-1])
    got, ref = eng.render(None, *pose), ora.render(w, h, *pose)
    mx, exact = assert_frames_match(got, ref, f"pose {pose}")
    print(f"pose {pose}: max|d|={mx} exact={exact:.4%} caps={eng.stats()}")
    assert np.array_equal(np.float32(eng.camera_position), np.float32(ora.camera_position))
    eng.close()


@pytest.mark.parametrize("cam", [(float("nan"), 0.5, -3.5), (float("nan"),) * 3, (float("inf"), 0.5, -3.5)])
def test_camera_that_is_not_a_number(cam, oracle_lightmaps):
    """NaN and infinite camera coordinates: distances through that coordinate are NaN, which
    `f32::min` drops and every comparison rejects (ray.rs:29-35); NaN colours become 0 bytes
    (engine.rs:21-22).  Nothing faults or hangs, and the frame equals the oracle's."""
    from raymarching_amd import Engine

    base = default_scene()
    scene = Scene(objects=base.objects, camera_position=cam, light=base.light)
    lm, _ = oracle_lightmaps
    w, h = 64, 48
    eng = Engine(scene, w, h)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[-1][i])
    ora = make_oracle(scene, lm[-1])
    got, ref = eng.render(None), ora.render(w, h)
    mx, exact = assert_frames_match(got, ref, f"camera {cam}")
    print(f"camera {cam}: max|d|={mx} exact={exact:.4%} caps={eng.stats()}")
    eng.close()


def test_large_frame_leaves_in_bands(oracle_lightmaps):
    """Frames of 4 Mpixel and more are copied to the host in row bands while later bands render:
    into pageable memory, into page-locked memory, and with a padded pitch the bytes equal the frame
    left on the device (1032 rows: the bands do not divide evenly)."""
    import ctypes

    import torch

    from raymarching_amd import Engine, _lib

    scene = default_scene()
    lm, _ = oracle_lightmaps
    w, h = 4104, 1032
    eng = Engine(scene, w, h)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[-1][i])
    eng.input(-60, 25, (0.1, 0.0, 0.2), 0)  # move once; the renders below keep the pose (zero movement)
    pageable = eng.render(None, -60, 25, (0.0, 0.0, 0.0), 0)
    dev = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    eng.render_rows_device(dev.data_ptr(), 0, h)
    eng.synchronize()
    ref = dev.cpu().numpy()
    assert np.array_equal(pageable, ref)
    lib = _lib.load()
    pitch = w * 4 + 256
    for p, what in ((w * 4, "page-locked"), (pitch, "page-locked, padded pitch")):
        ptr = lib.rm_host_alloc(p * h)
        view = np.ctypeslib.as_array((ctypes.c_uint8 * (p * h)).from_address(ptr)).reshape(h, p)
        view[:] = 0xAB
        _lib.check(lib.rm_engine_render(eng.handle, ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), p, -60, 25, _lib.f3((0.0, 0.0, 0.0)), 0), what)
        assert np.array_equal(view[:, : w * 4].reshape(h, w, 4), ref), what
        assert (view[:, w * 4:] == 0xAB).all(), what
        lib.rm_host_free(ptr)
    padded = np.full((h, pitch), 0xCD, dtype=np.uint8)  # pageable with a pitch
    _lib.check(lib.rm_engine_render(eng.handle, padded.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), pitch, -60, 25, _lib.f3((0.0, 0.0, 0.0)), 0), "pageable, padded pitch")
    assert np.array_equal(padded[:, : w * 4].reshape(h, w, 4), ref) and (padded[:, w * 4:] == 0xCD).all()
    eng.close()


def test_bad_arguments_are_refused_with_a_message():
    """Every entry point answers a bad call with a status and a message instead of the reference's
    abort: sizes and modes outside what a kernel assumes never reach a launch."""
    import torch

    from raymarching_amd import Engine, _lib, default_scene

    eng = Engine(default_scene(), 66, 40)  # a width that is not a multiple of 4
    buf = torch.zeros((40, 66, 4), dtype=torch.uint8, device="cuda")
    out = torch.zeros_like(buf)
    with pytest.raises(_lib.RMError, match="multiple of 4"):
        eng.pack_bgr_device(buf.data_ptr(), out.data_ptr(), 66 * 40 - 2)
    with pytest.raises(_lib.RMError, match="multiple of 4"):
        eng.unpack_bgr_strips_device(buf.data_ptr(), out.data_ptr(), 1, 5)
    with pytest.raises(_lib.RMError, match="out of place"):
        eng.fxaa_device(buf.data_ptr(), buf.data_ptr(), 0, 40)
    with pytest.raises(_lib.RMError, match="row range"):
        eng.fxaa_device(buf.data_ptr(), out.data_ptr(), 0, 41)
    with pytest.raises(_lib.RMError, match="1 or 4"):
        eng.set_ssaa(3)
    eng.close()
    eng = Engine(default_scene(), 64, 40)
    with pytest.raises(_lib.RMError, match="do not cover"):
        eng.unpack_bgr_strips_device(buf.data_ptr(), out.data_ptr(), 2, 2)  # 4 strips for a frame of 5
    eng.close()
