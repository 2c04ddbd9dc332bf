"""GPU parity tests for the per-pixel path: HIP kernels (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

from conftest import assert_frames_match, frame_stats, make_oracle

pytestmark = pytest.mark.gpu

W, H = 800, 600  # engine.rs:9-10


@pytest.fixture(scope="module")
def gpu_engine(scene, oracle_lightmaps):
    from raymarching_amd import Engine

    e = Engine(scene, W, H)
    lm, _ = oracle_lightmaps
    for i in range(e.n_objects):
        e.set_lightmap(i, lm[-1][i])
    yield e
    e.close()


def test_primary_march_bit_exact(scene, gpu_engine):
    """Ray::march from the camera: hit index, iteration count and end position are
    pure IEEE add/mul/sqrt/compare work, so they must agree bit for bit."""
    ora = make_oracle(scene)
    hit_o, steps_o, pos_o = ora.trace_primary(W, H)
    gpu_engine.input(0, 0, (0, 0, 0), 0)
    hit_g, steps_g, pos_g = gpu_engine.trace_primary()
    assert np.array_equal(hit_g, hit_o)
    assert np.array_equal(steps_g, steps_o)
    assert np.array_equal(pos_g.view(np.uint32), pos_o.view(np.uint32))
    # centre pixel: dir = (0,0,1) from (0,0.5,-3.5) reaches the back wall z=2 (object 6)
    assert hit_g[300, 400] == 6


def test_frame_default_pose(scene, gpu_engine, oracle_lightmaps):
    lm, _ = oracle_lightmaps
    ora = make_oracle(scene, lm[-1])
    ref = ora.render(W, H)
    gpu_engine.camera_position = scene.camera_position
    got = gpu_engine.render()
    mx, exact = assert_frames_match(got, ref, "default pose")
    print(f"default pose: max|d|={mx} exact={exact:.6%}")
    assert (got[:, :, 3] == 0).all()
    st = gpu_engine.stats()
    assert st["depth_cap_hits"] == 0 and st["step_cap_hits"] == 0 and st["lightmap_oob"] == 0


@pytest.mark.parametrize(
    "mouse_x,mouse_y,move,scroll",
    [
        (150, -60, (0.0, 0.0, 0.0), 0),
        (-420, 90, (0.05, 0.0, 0.15), 2),
        (900, 200, (-0.15, 0.0, -0.05), -3),
    ],
)
def test_frame_other_poses(scene, gpu_engine, oracle_lightmaps, mouse_x, mouse_y, move, scroll):
    """Engine::render is stateful in the camera: both sides start from the scene's
    camera and apply the same input once."""
    lm, _ = oracle_lightmaps
    ora = make_oracle(scene, lm[-1])
    ref = ora.render(W, H, mouse_x, mouse_y, move, scroll)
    gpu_engine.camera_position = scene.camera_position
    got = gpu_engine.render(None, mouse_x, mouse_y, move, scroll)
    assert np.array_equal(np.float32(gpu_engine.camera_position), np.float32(ora.camera_position))
    mx, exact = assert_frames_match(got, ref, f"pose {mouse_x},{mouse_y},{scroll}")
    print(f"pose ({mouse_x},{mouse_y},{move},{scroll}): max|d|={mx} exact={exact:.6%}")


def test_stateful_camera_sequence(scene, gpu_engine, oracle_lightmaps):
    """Three consecutive render() calls with movement accumulate the camera exactly as the reference."""
    lm, _ = oracle_lightmaps
    ora = make_oracle(scene, lm[-1])
    gpu_engine.camera_position = scene.camera_position
    inputs = [(30, 10, (0.0, 0.0, 0.05), 0), (60, 20, (0.05, 0.0, 0.05), 1), (90, 25, (0.0, 0.0, -0.15), 1)]
    for mx_, my_, mv, sc in inputs:
        ref = ora.render(320, 240, mx_, my_, mv, sc)
        del ref
    cam_o = np.float32(ora.camera_position)
    for mx_, my_, mv, sc in inputs:
        gpu_engine.input(mx_, my_, mv, sc)
    assert np.array_equal(np.float32(gpu_engine.camera_position), cam_o)


def test_rows_equal_full_frame(scene, gpu_engine):
    """A band rendered on its own equals the same rows of the full frame (band sharding)."""
    import torch

    gpu_engine.camera_position = scene.camera_position
    gpu_engine.input(0, 0, (0, 0, 0), 0)
    full = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    gpu_engine.render_rows_device(full.data_ptr(), 0, H)
    for r0, r1 in [(0, 75), (75, 150), (293, 301), (599, 600)]:
        band = torch.empty((r1 - r0, W, 4), dtype=torch.uint8, device="cuda")
        gpu_engine.render_rows_device(band.data_ptr(), r0, r1)
        gpu_engine.synchronize()
        assert torch.equal(band, full[r0:r1])


def test_strips_equal_full_frame(scene, gpu_engine):
    """Every n-th 8-row strip rendered compactly equals those rows of the full frame, and the
    strips of all ranks reassemble to it (interleaved sharding over GPUs)."""
    import torch

    from raymarching_amd import dist as rdist

    gpu_engine.camera_position = scene.camera_position
    gpu_engine.input(0, 0, (0, 0, 0), 0)
    full = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    gpu_engine.render_rows_device(full.data_ptr(), 0, H)
    for world in (1, 3, 8):
        parts = []
        for rank in range(world):
            first, stride, per = rdist.strip_plan(H, rank, world)
            buf = torch.zeros((per * 8, W, 4), dtype=torch.uint8, device="cuda")
            gpu_engine.render_strips_device(buf.data_ptr(), first, stride, per)
            parts.append(buf)
        gpu_engine.synchronize()
        _, _, per = rdist.strip_plan(H, 0, world)
        frame = rdist.deinterleave_strips(torch.stack(parts), world, per, W, H)
        assert torch.equal(frame, full), world


def test_packed_strips_reassemble_to_the_frame(scene, gpu_engine):
    """Strips travel between GPUs as 3 bytes per pixel (the fourth byte of a rendered pixel is 0):
    every rank's strips packed, laid out as an all-gather leaves them and unpacked again give the
    full frame bit for bit, for rank counts that do and do not divide the number of strips."""
    import torch

    from raymarching_amd import dist as rdist

    gpu_engine.camera_position = scene.camera_position
    gpu_engine.input(0, 0, (0, 0, 0), 0)
    full = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    gpu_engine.render_rows_device(full.data_ptr(), 0, H)
    gpu_engine.synchronize()
    assert int(full[..., 3].max()) == 0
    for world in (1, 2, 3, 8):
        _, _, per = rdist.strip_plan(H, 0, world)
        gathered = torch.full((world, per * 8, W, 3), 0xAB, dtype=torch.uint8, device="cuda")
        for rank in range(world):
            first, stride, per = rdist.strip_plan(H, rank, world)
            buf = torch.zeros((per * 8, W, 4), dtype=torch.uint8, device="cuda")
            gpu_engine.render_strips_device(buf.data_ptr(), first, stride, per)
            gpu_engine.pack_bgr_device(buf.data_ptr(), gathered[rank].data_ptr(), per * 8 * W)
            gpu_engine.synchronize()
            assert torch.equal(gathered[rank], buf[..., :3]), (world, rank)
        frame = torch.full((H, W, 4), 0xCD, dtype=torch.uint8, device="cuda")
        gpu_engine.unpack_bgr_strips_device(gathered.data_ptr(), frame.data_ptr(), world, per)
        gpu_engine.synchronize()
        assert torch.equal(frame, full), world
    # rows already in frame order (contiguous bands): world = 1
    packed = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    gpu_engine.pack_bgr_device(full.data_ptr(), packed.data_ptr(), H * W)
    frame = torch.empty_like(full)
    gpu_engine.unpack_bgr_strips_device(packed.data_ptr(), frame.data_ptr(), 1, (H + 7) // 8)
    gpu_engine.synchronize()
    assert torch.equal(frame, full)


def test_probe_shade_matches_oracle(scene, gpu_engine, oracle_lightmaps):
    """Engine::shade_object at scattered surface points, including the reflective spheres."""
    lm, _ = oracle_lightmaps
    ora = make_oracle(scene, lm[-1])
    hit, _, pos = ora.trace_primary(W, H)
    rng = np.random.default_rng(7)
    ys = rng.integers(0, H, 400)
    xs = rng.integers(0, W, 400)
    objs = hit[ys, xs].astype(np.int32)
    pts = pos[ys, xs]
    pose = ora.pose()
    dirs = np.empty_like(pts)
    import ctypes

    for k, (x, y) in enumerate(zip(xs, ys)):
        out = (ctypes.c_float * 3)()
        ora.lib.orc_kat_pixel_direction(W, H, int(x), int(y), (ctypes.c_float * 5)(*pose[:5]), out)
        dirs[k] = out
    got = gpu_engine.probe_shade(objs, pts, dirs)
    ref = np.stack([ora.shade(o, p, d) for o, p, d in zip(objs, pts, dirs)])
    # linear colours: identical up to the last-bit freedom of powf/atan2f
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("intensity", [3.5, 2.0 ** 41, 2.0 ** -41])
def test_probe_shade_outside_the_usual_ranges(scene, oracle_lightmaps, intensity):
    """The light terms of shade_object (distance, direction, intensity / distance^2) where the
    kernel's shared-work roots and quotients do not apply and the IEEE expansions run instead: a
    light intensity outside [2^-40, 2^40), points whose offset from the light has a component that is
    tiny but not zero, exactly zero, or huge, and points at the light itself."""
    import copy

    from raymarching_amd import Engine

    lm, _ = oracle_lightmaps
    sc = copy.deepcopy(scene)
    sc.light.intensity = float(np.float32(intensity))
    lx, ly, lz = (float(np.float32(v)) for v in sc.light.position)
    pts = np.array([
        [lx + 1e-33, -1.0, 0.5],      # a component below 2^-90 that is not zero
        [lx, ly + 1.0, lz],           # two components exactly zero
        [lx - 1e-33, ly, lz + 2e-36],  # every component tiny: the squared distance underflows
        [3e11, 0.5, -2e11],           # squared distance above 2^40
        [1e20, 1e20, 1e20],           # squared distance overflows
        [lx, ly, lz],                 # at the light: distance zero
        [0.3, -1.0, 0.2], [-1.0, 0.4, 0.1], [0.0, 0.5, 2.0],  # ordinary points
    ], dtype=np.float32)
    dirs = np.tile(np.array([[0.0, 0.0, 1.0]], dtype=np.float32), (len(pts), 1))
    eng = Engine(sc, 64, 64)
    ora = make_oracle(sc, lm[-1])
    for i, values in enumerate(lm[-1]):
        if values is not None:
            eng.set_lightmap(i, values)
    # a sphere and three untextured planes (at the absurd coordinates the floor's texture lookup is
    # not a number in different channels here and in the oracle, whatever the light terms do)
    chosen = [i for i, o in enumerate(sc.objects) if type(o).__name__ == "Sphere"][:1]
    chosen += [i for i, o in enumerate(sc.objects) if type(o).__name__ != "Sphere" and getattr(o, "texture", None) is None][:3]
    assert len(chosen) == 4
    worst = 0.0
    for obj in chosen:
        objs = np.full(len(pts), obj, dtype=np.int32)
        got = eng.probe_shade(objs, pts, dirs)
        ref = np.stack([ora.shade(obj, p, d) for p, d in zip(pts, dirs)])
        assert np.array_equal(np.isnan(got), np.isnan(ref)), (obj, got, ref)
        assert np.array_equal(np.isinf(got), np.isinf(ref)), (obj, got, ref)
        ok = np.isfinite(ref)
        # linear colours: identical up to the last-bit freedom of powf/atan2f
        np.testing.assert_allclose(got[ok], ref[ok], rtol=2e-5, atol=1e-30)
        worst = max(worst, float(np.max(np.abs(got[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-30), initial=0.0)))
    print(f"intensity {intensity:g}: largest relative difference {worst:.2e}")
    eng.close()
    ora.close()


@pytest.mark.parametrize("threshold_us,size", [(None, (1280, 720)), (5.0, (1280, 720)), (0.2, (1280, 720)), (5.0, (1001, 699))])
def test_slow_tiles_in_quarters_render_the_same_frame(scene, oracle_lightmaps, threshold_us, size, monkeypatch):
    """Launches planned from the tiles' last timings (slow tiles first, as four waves of 4x4 pixels):
    every frame equals the one rendered with the plans switched off, for whole frames, row ranges that
    do not start on a tile boundary, strips, and a camera that has moved since the timings were taken.
    Thresholds: the default, 5 us (a third of the tiles) and 0.2 us (every tile: the list overflows)."""
    import torch

    from raymarching_amd import Engine

    if threshold_us is not None:
        monkeypatch.setenv("RM_SPLIT_US", str(threshold_us))
        monkeypatch.setenv("RM_SPLIT_TIMES_MEAN", "0")  # the absolute threshold alone decides
    lm, _ = oracle_lightmaps
    w, h = size  # (the second size ends in partial tiles on both axes)

    def engine(split):
        e = Engine(scene, w, h)
        for i, values in enumerate(lm[-1]):
            e.set_lightmap(i, values)
        e.set_tile_splitting(split)
        return e

    plain, planned = engine(False), engine(True)

    def rows(e, r0, r1):
        buf = torch.zeros((r1 - r0, w, 4), dtype=torch.uint8, device="cuda")
        e.render_rows_device(buf.data_ptr(), r0, r1)
        e.synchronize()
        return buf

    def strips(e, first, stride, n):
        buf = torch.zeros((n * 8, w, 4), dtype=torch.uint8, device="cuda")
        e.render_strips_device(buf.data_ptr(), first, stride, n)
        e.synchronize()
        return buf

    split_counts = []
    first_pose_frame = None
    for pose in ((0, 0, (0.0, 0.0, 0.0), 0), (35, -20, (0.25, 0.0, 0.5), 1), (-60, 10, (-0.4, 0.1, 0.2), -1)):
        plain.input(*pose)
        planned.input(*pose)
        want = rows(plain, 0, h)
        for frame in range(3):
            got = rows(planned, 0, h)
            assert torch.equal(got, want), (pose, frame)
            split_counts.append(planned.split_tiles())
            if first_pose_frame is None:
                first_pose_frame = got.cpu().numpy()
        assert torch.equal(rows(planned, 100, 613), want[100:613])
        for first, stride, n in ((1, 3, 30), (2, 4, 23)):  # (the last strips named lie below the frame)
            got = strips(planned, first, stride, n)
            for k in range(n):
                r = (first + stride * k) * 8
                inside = max(0, min(8, h - r))
                assert torch.equal(got[k * 8:k * 8 + inside], want[r:r + inside]), (pose, first, k)
                below = got[k * 8 + inside:(k + 1) * 8]
                assert below.numel() == 0 or int(below.max()) == 0
    print(f"threshold {threshold_us}: tiles in quarters per whole-frame launch {split_counts}")
    assert split_counts[0] == 0  # nothing is known before the first frame
    assert max(split_counts) > 0
    if threshold_us == 0.2:
        assert max(split_counts) == 900  # the cap of this launch: 14400 tiles / 16
    assert plain.split_tiles() == 0
    ref = make_oracle(scene, lm[-1]).render(w, h)
    assert_frames_match(first_pose_frame, ref, "first pose")
    plain.close()
    planned.close()


def test_planned_launches_on_many_streams(scene, oracle_lightmaps):
    """Each stream that launches frames of an engine has plans of its own (an engine keeps 32; the
    launches of a stream that finds none free go unplanned): 36 streams render the frame three times
    each, launches in flight side by side, and every frame equals the one rendered with the plans
    switched off."""
    import torch

    from raymarching_amd import Engine

    lm, _ = oracle_lightmaps
    w, h = 1280, 720
    e = Engine(scene, w, h)
    for i, values in enumerate(lm[-1]):
        e.set_lightmap(i, values)
    e.set_tile_splitting(False)
    main = torch.cuda.current_stream()
    e.set_stream(main.cuda_stream)
    want = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    e.render_rows_device(want.data_ptr(), 0, h)
    e.synchronize()
    e.set_tile_splitting(True)
    streams = [torch.cuda.Stream() for _ in range(36)]
    frames = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in streams]
    torch.cuda.synchronize()
    for turn in range(3):
        for st, buf in zip(streams, frames):
            e.set_stream(st.cuda_stream)
            e.render_rows_device(buf.data_ptr(), 0, h)
        torch.cuda.synchronize()
        for k, buf in enumerate(frames):
            assert torch.equal(buf, want), (turn, k)
            buf.zero_()
        torch.cuda.synchronize()
    e.set_stream(main.cuda_stream)
    e.close()


def test_one_stream_that_cycles_through_more_sets_of_tiles_than_plans_are_kept(scene, oracle_lightmaps):
    """A plan names the tiles of one launch shape; a stream that cycles through 40 row ranges (more than
    the 32 plans an engine keeps) gives up the plan it has not used for longest each time, and every
    range comes out as in the frame rendered with the plans switched off - three times round, so that
    every plan has been given up and made again."""
    import torch

    from raymarching_amd import Engine

    lm, _ = oracle_lightmaps
    w, h = 1280, 720
    e = Engine(scene, w, h)
    for i, values in enumerate(lm[-1]):
        e.set_lightmap(i, values)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_tile_splitting(False)
    want = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    e.render_rows_device(want.data_ptr(), 0, h)
    e.synchronize()
    e.set_tile_splitting(True)
    ranges = [(8 * k, h - 8 * (k % 7)) for k in range(40)]
    got = torch.zeros_like(want)
    for turn in range(3):
        for r0, r1 in ranges:
            got.zero_()
            e.render_rows_device(got[r0:].data_ptr(), r0, r1)  # (the pointer names the range's first row)
            e.synchronize()
            assert torch.equal(got[r0:r1], want[r0:r1]), (turn, r0, r1)
            assert not got[:r0].any() and not got[r1:].any(), (turn, r0, r1)
    e.close()


def test_headless_host_harness(tmp_path, scene, oracle_lightmaps):
    """The C++ host mirror of main.rs (host/main.cpp) renders the default frame through the C ABI."""
    import os
    import struct
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "rm_headless")
    if not os.path.exists(exe):
        pytest.skip("host/rm_headless is not built")
    out = tmp_path / "frame.bmp"
    res = subprocess.run([exe, "--frames", "2", "--out", str(out), "--texture", os.path.join(root, "assets", "textures", "Floor128.bmp")],
                         capture_output=True, text=True, timeout=100)
    assert res.returncode == 0, res.stderr
    assert res.stdout.count("Elapsed:") == 2
    data = out.read_bytes()
    w, h = struct.unpack_from("<ii", data, 18)
    assert (w, -h) == (W, H)
    got = np.frombuffer(data, dtype=np.uint8, offset=54).reshape(H, W, 4)
    lm, _ = oracle_lightmaps
    ref = make_oracle(scene, lm[-1]).render(W, H)
    assert_frames_match(got, ref, "headless harness")
    # the same host with the frame and the lightmap patches spread over "three GPUs" (this one, three
    # times): the image file is the same byte for byte
    out3 = tmp_path / "frame3.bmp"
    res = subprocess.run([exe, "--frames", "2", "--out", str(out3), "--devices", "0,0,0", "--texture",
                          os.path.join(root, "assets", "textures", "Floor128.bmp")], capture_output=True, text=True, timeout=100)
    assert res.returncode == 0, res.stderr
    assert '"gpus": 3' in res.stdout
    assert out3.read_bytes() == data


def test_supersampled_frame(scene, oracle_lightmaps):
    """The four-ray supersampling the reference carries commented out (engine.rs:95-108): same
    tolerance as the live path, at two poses."""
    from raymarching_amd import Engine

    w, h = 240, 180
    lm, _ = oracle_lightmaps
    ora = make_oracle(scene, lm[-1])
    ora.set_ssaa(4)
    eng = Engine(scene, w, h)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[-1][i])
    eng.set_ssaa(4)
    for pose in [(0, 0, (0.0, 0.0, 0.0), 0), (-140, 60, (0.3, 0.0, 0.6), 2)]:
        got = eng.render(None, *pose)
        ref = ora.render(w, h, *pose)
        mx, exact = assert_frames_match(got, ref, f"ssaa {pose}")
        print(f"ssaa pose {pose}: max|d|={mx} exact={exact:.5%}")
    eng.set_ssaa(1)
    ora.set_ssaa(1)
    assert_frames_match(eng.render(None), ora.render(w, h), "ssaa off again")
    with pytest.raises(Exception, match="1 or 4"):
        eng.set_ssaa(3)
    eng.close()


def test_tetrahedral_normals(scene):
    """The four-point normal the reference carries commented out (objects.rs:142-149), through the
    lightmap precompute (cloud and direct-light normals) and a frame."""
    from raymarching_amd import Engine

    w, h = 240, 180
    ora = make_oracle(scene)
    ora.set_tetrahedral_normals(True)
    ora.compute_lightmaps(2)
    eng = Engine(scene, w, h)
    eng.set_tetrahedral_normals(True)
    eng.compute_lightmaps(2)
    for i in range(eng.n_objects):
        assert np.array_equal(eng.get_lightmap(i).view(np.uint32), ora.get_lightmap(i).view(np.uint32)), i
    pose = (-90, 40, (0.2, 0.0, 0.4), 1)
    mx, exact = assert_frames_match(eng.render(None, *pose), ora.render(w, h, *pose), "tetrahedral normals")
    print(f"tetrahedral normals: max|d|={mx} exact={exact:.5%}")
    eng.close()


def test_refraction(scene, oracle_lightmaps):
    """The refraction block the reference carries commented out (engine.rs:196-228, ray.rs:44-56,
    vector.rs:40-48): frames of the default scene (both spheres refract), one of them supersampled,
    and shade_object at points on the spheres.  No recursion is cut on either side, so the frames are
    the unbounded recursion's."""
    from raymarching_amd import Engine

    w, h = 320, 240
    lm, _ = oracle_lightmaps
    ora = make_oracle(scene, lm[-1])
    ora.set_refraction(True)
    eng = Engine(scene, w, h)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[-1][i])
    plain = eng.render(None, 0, 0, (0.0, 0.0, 0.0), 0)
    eng.set_refraction(True)
    poses = [(0, 0, (0.0, 0.0, 0.0), 0), (310, 120, (0.9, -0.1, 1.6), 3), (-420, 60, (-0.6, 0.2, 1.1), -2)]
    for k, pose in enumerate(poses):
        got, ref = eng.render(None, *pose), ora.render(w, h, *pose)
        mx, exact = assert_frames_match(got, ref, f"refraction pose {k}")
        print(f"refraction pose {k}: max|d|={mx} exact={exact:.5%}")
        if k == 0:
            assert (got != plain).any(axis=-1).mean() > 0.02  # the spheres look different with it
    eng.set_ssaa(4)
    ora.set_ssaa(4)
    mx, exact = assert_frames_match(eng.render(None, *poses[1]), ora.render(w, h, *poses[1]), "refraction, supersampled")
    print(f"refraction + ssaa: max|d|={mx} exact={exact:.5%}")
    ora.set_ssaa(1)

    # shade_object on sphere points seen from the last pose
    hit, _, pos = ora.trace_primary(w, h)
    ys, xs = np.nonzero(hit <= 1)
    pick = np.random.default_rng(5).choice(len(ys), 200, replace=False)
    objs, pts = hit[ys[pick], xs[pick]].astype(np.int32), pos[ys[pick], xs[pick]]
    dirs = pts - np.asarray(ora.camera_position, dtype=np.float32)
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    got = eng.probe_shade(objs, pts, dirs)
    ref = np.stack([ora.shade(o, p, d) for o, p, d in zip(objs, pts, dirs)])
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-6)
    assert eng.stats()["depth_cap_hits"] == 0 and ora.refraction_depth_cuts() == 0
    eng.close()


def test_frame_sequence_with_recorded_input(scene, oracle_lightmaps):
    """A recorded input trace (mouse look, WASD movement, scroll zoom; main.rs:105-138) played
    through the pipelined rm_engine_render_async path: every frame and the final camera match the
    oracle, which renders the same trace frame by frame."""
    import ctypes

    from raymarching_amd import Engine, _lib

    w, h = 200, 150
    lm, _ = oracle_lightmaps
    ora = make_oracle(scene, lm[-1])
    eng = Engine(scene, w, h)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[-1][i])
    rng = np.random.default_rng(5)
    trace, mx, my, sc = [], 0, 0, 0
    for f in range(24):
        mx += int(rng.integers(-40, 41))
        my += int(rng.integers(-15, 16))
        sc += int(rng.integers(-1, 2)) if f % 5 == 0 else 0
        speed = 0.15 if f % 7 == 0 else 0.05  # LShift, main.rs:124-128
        move = (float(rng.integers(-1, 2)) * speed, 0.0, float(rng.integers(-1, 2)) * speed)
        trace.append((mx, my, move, sc))
    lib = _lib.load()
    n_bytes = w * h * 4
    bufs = [lib.rm_host_alloc(n_bytes) for _ in range(2)]
    views = [np.ctypeslib.as_array((ctypes.c_uint8 * n_bytes).from_address(b)).reshape(h, w, 4) for b in bufs]
    got = []
    pending = []
    for f, (a, b, mv, s_) in enumerate(trace):
        if len(pending) == 2:  # the frame that used this buffer two submissions ago must be read first
            eng.render_wait()
            got.append(views[pending.pop(0)].copy())
        eng.render_async(bufs[f & 1], a, b, mv, s_)
        pending.append(f & 1)
    while pending:
        eng.render_wait()
        got.append(views[pending.pop(0)].copy())
    worst, inexact = 0, 0.0
    for f, (a, b, mv, s_) in enumerate(trace):
        ref = ora.render(w, h, a, b, mv, s_)
        mxd, exact = assert_frames_match(got[f], ref, f"frame {f}")
        worst, inexact = max(worst, mxd), max(inexact, 1.0 - exact)
    print(f"24-frame trace: worst max|d|={worst}, worst inexact fraction={inexact:.2e}")
    assert np.array_equal(np.float32(eng.camera_position), np.float32(ora.camera_position))
    # a pageable destination is refused (the copy could not overlap the next frame)
    with pytest.raises(_lib.RMError, match="page-locked"):
        eng.render_async(np.empty(n_bytes, dtype=np.uint8).ctypes.data)
    with pytest.raises(_lib.RMError, match="pending"):
        eng.render_wait()
    for b in bufs:
        lib.rm_host_free(b)
    eng.close()


def test_settings_change_between_frames_in_flight(scene, oracle_lightmaps):
    """Switches and lightmaps changed between two rm_engine_render_async frames: the frame already in
    flight keeps what it was submitted with (there is one device copy of the settings for all frames),
    the next one gets the new values.  Each frame is compared with a synchronous render at its settings."""
    import ctypes

    from raymarching_amd import Engine, _lib

    w, h = 1920, 1080  # large enough for frame 1 to be still rendering when the switch is flipped
    lm, _ = oracle_lightmaps
    eng = Engine(scene, w, h)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[-1][i])
    plain = eng.render()
    eng.set_refraction(True)
    refr = eng.render()
    eng.set_refraction(False)
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[1][i])  # after one bounce: darker
    dark = eng.render()
    for i in range(eng.n_objects):
        eng.set_lightmap(i, lm[-1][i])
    assert not np.array_equal(plain, refr) and not np.array_equal(plain, dark)
    lib = _lib.load()
    n_bytes = w * h * 4
    bufs = [lib.rm_host_alloc(n_bytes) for _ in range(2)]
    views = [np.ctypeslib.as_array((ctypes.c_uint8 * n_bytes).from_address(b)).reshape(h, w, 4) for b in bufs]
    for second, change, undo in (
        (refr, lambda: eng.set_refraction(True), lambda: eng.set_refraction(False)),
        (dark, lambda: [eng.set_lightmap(i, lm[1][i]) for i in range(eng.n_objects)],
         lambda: [eng.set_lightmap(i, lm[-1][i]) for i in range(eng.n_objects)]),
    ):
        eng.render_async(bufs[0])
        change()
        eng.render_async(bufs[1])
        eng.render_wait()
        eng.render_wait()
        assert np.array_equal(views[0], plain)
        assert np.array_equal(views[1], second)
        undo()
    for b in bufs:
        lib.rm_host_free(b)
    eng.close()


def test_tone_map_estimate_never_disagrees_with_the_exact_operations(gpu_engine):
    """engine.rs:109-125: the frames pack a pixel from one-ulp estimates of the division and the
    square root and take the exact operations only near a rounding boundary.  Both paths side by side
    on 12 million colours: random ones over the whole range, every float around the 255 boundaries of
    each byte (found by bisection on the exact path), zeros, negatives, infinities and NaNs - and the
    exact path against the oracle's restatement on a sample."""
    import ctypes

    import oracle as orc

    rng = np.random.default_rng(42)
    parts = [rng.uniform(0.0, 1.2, (4_000_000, 3)), 10.0 ** rng.uniform(-12, 3, (2_000_000, 3)),
             np.array([[0.0, -0.0, 1.0], [np.inf, -np.inf, np.nan], [-1.0, -1e-30, 1e-45], [3e38, 1e30, 1e-38]])]
    # the inputs where the exact byte changes: bisect x for each boundary k - 1/2 -> k, then take a
    # window of neighbouring floats on both sides
    lo = np.zeros(255, dtype=np.float32)
    hi = np.full(255, 64.0, dtype=np.float32)
    target = np.arange(1, 256, dtype=np.uint32)
    for _ in range(40):
        mid = ((lo.astype(np.float64) + hi.astype(np.float64)) / 2).astype(np.float32)
        _, exact = gpu_engine.probe_pack(np.stack([mid, mid, mid], axis=1))
        below = (exact & 255) < target
        lo = np.where(below, mid, lo)
        hi = np.where(below, hi, mid)
    window = np.arange(-4000, 4001, dtype=np.int64)
    near = (hi.view(np.uint32).astype(np.int64)[:, None] + window[None, :]).astype(np.uint32).view(np.float32).reshape(-1)
    parts.append(np.stack([near, near[::-1], np.roll(near, 12345)], axis=1))
    colours = np.concatenate(parts).astype(np.float32)
    fast, exact = gpu_engine.probe_pack(colours)
    assert np.array_equal(fast, exact), np.flatnonzero(fast != exact)[:10]
    assert len(np.unique(exact[-len(near):] & 255)) == 256  # the window really crosses every boundary
    # the exact path is the oracle's arithmetic (colour.rs:33-44, engine.rs:21-22, 110)
    lib = orc.load()
    sample = colours[rng.integers(0, len(colours), 20000)]
    want = np.empty(len(sample), dtype=np.uint32)
    lib.orc_kat_pack_pixels(len(sample), sample.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), want.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    _, got = gpu_engine.probe_pack(sample)
    assert np.array_equal(got, want)


def test_two_pass_frame_is_the_same_frame(scene, oracle_lightmaps, monkeypatch):
    """The experimental two-pass frame (RM_TWO_PASS: primary hits and local shading first, the queued
    reflection rays in full waves afterwards) does each pixel's arithmetic in the same order: same bytes,
    whole frame and a row range, with a pass-2 grid smaller and larger than the queue."""
    from raymarching_amd import Engine

    w, h = 640, 360
    lm, _ = oracle_lightmaps

    def frames(env):
        if env is None:
            monkeypatch.delenv("RM_TWO_PASS", raising=False)
        else:
            monkeypatch.setenv("RM_TWO_PASS", env)
        eng = Engine(scene, w, h)
        for i in range(eng.n_objects):
            eng.set_lightmap(i, lm[-1][i])
        out = [eng.render(None, 0, 0, (0.0, 0.0, 0.0), 0), eng.render(None, -300, 40, (0.2, 0.0, 0.4), 1)]
        st = eng.stats()
        eng.close()
        return out, st

    want, st0 = frames(None)
    for env in ("1", "7", "100000"):
        got, st = frames(env)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), env
        assert st == st0
