"""GPU parity tests for the radiosity precompute (Engine::compute_lightmaps)."""
import numpy as np
import pytest

from conftest import LIGHTMAP_RTOL

pytestmark = pytest.mark.gpu


def _rel_err(got, ref):
    scale = max(float(np.abs(ref).max()), 1e-30)  # all-zero maps (no emissive objects, 0 bounces) compare as 0
    return float(np.abs(got - ref).max() / scale)


@pytest.mark.parametrize("bounces", [0, 1, 4])
def test_lightmaps_match_oracle(scene, oracle_lightmaps, bounces):
    from raymarching_amd import Engine

    lm, _ = oracle_lightmaps
    e = Engine(scene, 64, 64)
    e.compute_lightmaps(bounces)
    got = np.stack([e.get_lightmap(i) for i in range(e.n_objects)])
    e.close()
    ref = lm[bounces]
    assert _rel_err(got, ref) <= LIGHTMAP_RTOL
    # no library call with a platform-defined last bit is involved: expect identity
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"max rel err {_rel_err(got, ref):.3e}"
    # luxels the oracle leaves empty (sample points outside the room) stay empty
    assert np.array_equal((got == 0).all(-1), (ref == 0).all(-1))


def test_form_factor_launch_in_chunks(scene, oracle_lightmaps, monkeypatch):
    """Sources beyond one launch's grid are taken in chunks (more than 65535 patches need it); a
    tiny chunk size exercises the same code on the default maps."""
    from raymarching_amd import Engine

    lm, _ = oracle_lightmaps
    monkeypatch.setenv("RM_RAD_CHUNK", "100")
    e = Engine(scene, 64, 64)
    e.compute_lightmaps(4)
    got = np.stack([e.get_lightmap(i) for i in range(e.n_objects)])
    e.close()
    assert np.array_equal(got.view(np.uint32), lm[4].view(np.uint32))


@pytest.mark.parametrize("per,chunk", [("3", "100"), ("16", None)])
def test_form_factors_several_sources_per_workgroup(scene, oracle_lightmaps, monkeypatch, per, chunk):
    """Large launches give a workgroup several sources, one after the other (rad_form_multi_kernel,
    chosen by launch size); forced here on the default maps, with a count that does not divide a
    launch's rows and with launches in chunks - the lightmaps equal the oracle's bit for bit."""
    from raymarching_amd import Engine

    lm, _ = oracle_lightmaps
    monkeypatch.setenv("RM_RAD_PER", per)
    if chunk:
        monkeypatch.setenv("RM_RAD_CHUNK", chunk)
    e = Engine(scene, 64, 64)
    e.compute_lightmaps(4)
    got = np.stack([e.get_lightmap(i) for i in range(e.n_objects)])
    e.close()
    assert np.array_equal(got.view(np.uint32), lm[4].view(np.uint32))


def test_stepwise_calls_after_a_whole_precompute(scene, oracle_lightmaps):
    """The one-call precompute lets the two emissive maps trade places from bounce to bounce; the
    stepwise calls made on the same engine afterwards (odd and even bounce counts before them) still
    deliver the oracle's lightmaps."""
    from raymarching_amd import Engine

    lm, _ = oracle_lightmaps
    e = Engine(scene, 64, 64)
    for before in (1, 4):
        e.compute_lightmaps(before)
        e.radiosity_begin()
        _, _, n = e.radiosity_buffers()
        e.radiosity_form_factors(0, n)
        for _ in range(4):
            e.radiosity_bounce(0, n)
            e.radiosity_commit()
        e.synchronize()
        got = np.stack([e.get_lightmap(i) for i in range(e.n_objects)])
        assert np.array_equal(got.view(np.uint32), lm[4].view(np.uint32)), before
    e.close()


def test_large_maps_run(scene):
    """MAP_SIZE 32 (8192 patches, 67 M form factors): finite, non-negative, and the same luxels
    empty as geometry dictates.  (The CPU oracle would need hours at this size.)"""
    from raymarching_amd import Engine

    e = Engine(scene, 64, 64, map_size=32)
    e.compute_lightmaps(4)
    got = np.stack([e.get_lightmap(i) for i in range(e.n_objects)])
    e.close()
    assert got.shape == (8, 32, 32, 3)
    assert np.isfinite(got).all() and (got >= 0).all()
    assert (got.reshape(8, -1).max(axis=1) > 0).all()


def test_sharded_bounces_equal_whole(scene):
    """Patch-sharded precompute (two halves, then an explicit exchange) equals the one-call result."""
    import torch

    from raymarching_amd import Engine

    whole = Engine(scene, 64, 64)
    whole.compute_lightmaps(4)
    ref = np.stack([whole.get_lightmap(i) for i in range(whole.n_objects)])
    whole.close()

    ranks = [Engine(scene, 64, 64) for _ in range(2)]
    for e in ranks:
        e.radiosity_begin()
    _, _, n = ranks[0].radiosity_buffers()
    cuts = [0, n // 2 + 17, n]
    for r, e in enumerate(ranks):
        e.radiosity_form_factors(cuts[r], cuts[r + 1])
    for _ in range(4):
        for r, e in enumerate(ranks):
            e.radiosity_bounce(cuts[r], cuts[r + 1])
            e.synchronize()
        # exchange: every rank receives the other's slice of next-emissive and lightmaps
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")
        for dst in range(2):
            src = 1 - dst
            em_s, lm_s, _ = ranks[src].radiosity_buffers()
            em_d, lm_d, _ = ranks[dst].radiosity_buffers()
            off = cuts[src] * 12
            size = (cuts[src + 1] - cuts[src]) * 12
            for a, b in ((em_s, em_d), (lm_s, lm_d)):
                rc = hip.hipMemcpy(ctypes.c_void_p(b + off), ctypes.c_void_p(a + off), ctypes.c_size_t(size), 3)
                assert rc == 0
        for e in ranks:
            e.radiosity_commit()
            e.synchronize()
    for e in ranks:
        got = np.stack([e.get_lightmap(i) for i in range(e.n_objects)])
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        e.close()


def _sharded_worker(rank, world, port, out_dir):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raymarching_amd import Engine, default_scene
    from raymarching_amd import dist as rdist

    torch.cuda.set_device(0)
    eng = Engine(default_scene(), 64, 64, device=0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    rdist.compute_lightmaps_sharded(eng, 4)
    got = np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])
    np.save(os.path.join(out_dir, f"lm_{rank}.npy"), got)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def test_patch_sharded_precompute_two_processes(tmp_path, oracle_lightmaps):
    """compute_lightmaps_sharded with two ranks (both on GPU 0, exchange over gloo): every rank ends
    up with the oracle's lightmaps, bit for bit."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref = oracle_lightmaps[0][4]
    for r in range(2):
        got = np.load(str(tmp_path / f"lm_{r}.npy"))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_lightmaps_saved_and_loaded(tmp_path, scene):
    """rm_engine_save_lightmaps / _load_lightmaps: a second engine that loads the file renders the same
    frame without running the precompute; files that do not fit are refused and change nothing."""
    from raymarching_amd import Engine, _lib

    eng = Engine(scene, 160, 120)
    eng.compute_lightmaps(4)
    want = np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])
    frame = eng.render()
    path = tmp_path / "lightmaps.rmlm"
    eng.save_lightmaps(path)
    assert path.stat().st_size == 16 + 8 * (4 + 8 * 8 * 3 * 4)
    eng.close()
    other = Engine(scene, 160, 120)
    other.load_lightmaps(path)
    got = np.stack([other.get_lightmap(i) for i in range(other.n_objects)])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(other.render(), frame)
    # refused: another map size, a truncated file, something that is no lightmap file
    big = Engine(scene, 64, 64, map_size=16)
    with pytest.raises(_lib.RMError, match="of size 8, the engine 8 of size 16"):
        big.load_lightmaps(path)
    big.close()
    cut = tmp_path / "cut.rmlm"
    cut.write_bytes(path.read_bytes()[:-100])
    with pytest.raises(_lib.RMError, match="ends early"):
        other.load_lightmaps(cut)
    junk = tmp_path / "junk.rmlm"
    junk.write_bytes(b"BM" + bytes(64))
    with pytest.raises(_lib.RMError, match="not a lightmap file"):
        other.load_lightmaps(junk)
    with pytest.raises(_lib.RMError, match="cannot read"):
        other.load_lightmaps(tmp_path / "missing.rmlm")
    got = np.stack([other.get_lightmap(i) for i in range(other.n_objects)])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    other.close()


def _scene_with_emission(scene, obj, emissive):
    import dataclasses

    objects = list(scene.objects)
    objects[obj] = dataclasses.replace(objects[obj], material=dataclasses.replace(objects[obj].material, emissive=emissive))
    return dataclasses.replace(scene, objects=objects)


@pytest.mark.parametrize("emissive", [float("inf"), float("nan")])
def test_emission_that_is_not_finite(scene, emissive):
    """radiosity.rs:130-152 with emission values that overflow: the reference leaves a blocked patch
    pair out of the sum and adds emission * 0 = NaN for an open pair that faces away.  With
    rm_engine_set_exact_occlusion the precompute does exactly that (every pair marched, blocked pairs
    skipped): same values, same NaNs.  The default takes two shortcuts that are invisible for finite
    emission (facing-away pairs are not marched, blocked pairs add emission * 0) - this test is the
    definition of what they do otherwise: a value may differ from the reference's only by being NaN."""
    from raymarching_amd import Engine
    from conftest import make_oracle

    glowing = _scene_with_emission(scene, 5, emissive)  # the green wall (main.rs:185)
    ora = make_oracle(glowing)
    stages, _ = ora.compute_lightmaps(4, stages=True)
    ora.close()
    for bounces in (1, 4):
        ref = stages[bounces]
        assert np.isnan(ref).any() and np.isfinite(ref).any()  # the case is not trivial: NaN has spread, not everywhere
        e = Engine(glowing, 64, 64)
        e.set_exact_occlusion(True)
        e.compute_lightmaps(bounces)
        exact = np.stack([e.get_lightmap(i) for i in range(e.n_objects)])
        e.set_exact_occlusion(False)
        e.compute_lightmaps(bounces)
        default = np.stack([e.get_lightmap(i) for i in range(e.n_objects)])
        e.close()
        assert np.array_equal(exact, ref, equal_nan=True), f"{bounces} bounces, exact occlusion"
        same = (default == ref) | (np.isnan(default) & np.isnan(ref))
        assert np.isnan(default[~same]).all(), f"{bounces} bounces: the shortcuts may only add NaNs"
    # finite emission: the switch changes nothing
    warm = _scene_with_emission(scene, 5, 0.75)
    ora = make_oracle(warm)
    ref = ora.compute_lightmaps(4, stages=True)[0][4]
    ora.close()
    e = Engine(warm, 64, 64)
    for on in (False, True):
        e.set_exact_occlusion(on)
        e.compute_lightmaps(4)
        got = np.stack([e.get_lightmap(i) for i in range(e.n_objects)])
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"finite emission, exact occlusion {on}"
    e.close()
