"""Several GPUs behind the C ABI, rehearsed on one: rm_group_* over RCCL with a one-rank communicator,
rm_group_* with 2 and 3 ranks on GPU 0 (one host thread per rank, the all-gather supplied by the test),
and rm_multi_* with one device listed several times.  Every gathered frame and every sharded lightmap
must equal the single-engine result bit for bit: sharding moves work, it changes no arithmetic."""
import ctypes
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSES = [(0, 0, (0.0, 0.0, 0.0), 0), (35, -12, (0.05, 0.0, 0.05), 0), (35, -12, (0.0, 0.0, 0.1), 1), (-80, 20, (0.0, 0.0, 0.0), 1)]


def _alias_u8(ptr, nbytes):
    import torch

    class _Mem:
        __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    return torch.as_tensor(_Mem(), device="cuda")


def _reference_frames(scene, w, h, fxaa, lightmaps=None):
    """The frames of POSES from one engine (and its FXAA), as [n][h][w][4] host arrays, plus its lightmaps."""
    import torch

    from raymarching_amd import Engine

    eng = Engine(scene, w, h)
    eng.compute_lightmaps(4)
    lm = np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])
    out = []
    raw = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    filt = torch.zeros_like(raw)
    for mx, my, mv, sc in POSES:
        eng.input(mx, my, mv, sc)
        eng.render_rows_device(raw.data_ptr(), 0, h)
        if fxaa:
            eng.fxaa_device(raw.data_ptr(), filt.data_ptr(), 0, h)
        eng.synchronize()
        out.append((filt if fxaa else raw).cpu().numpy().copy())
    cam = eng.camera_position
    eng.close()
    return out, lm, cam


@pytest.mark.parametrize("fxaa", [False, True])
def test_group_over_rccl_with_one_rank(scene, fxaa):
    """The RCCL path end to end (library lookup, communicator, all-gather on the frame's stream, frames
    in flight) with the only communicator a one-GPU box allows."""
    import torch

    from raymarching_amd import Engine
    from raymarching_amd.group import Group, unique_id

    w, h = 640, 360
    ref, ref_lm, ref_cam = _reference_frames(scene, w, h, fxaa)
    eng = Engine(scene, w, h)
    grp = Group(eng, 0, 1, unique_id(), frames_in_flight=3)
    assert grp.info() == {"rank": 0, "world": 1, "strips_per_rank": 45, "bytes_per_pixel_exchanged": 3}
    grp.compute_lightmaps(4)
    lm = np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])
    assert np.array_equal(lm.view(np.uint32), ref_lm.view(np.uint32))
    outs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in POSES]
    grp.set_timing(True)
    for (mx, my, mv, sc), out in zip(POSES, outs):
        grp.frame(out.data_ptr(), fxaa, mx, my, mv, sc)
    grp.wait()
    ms, n = grp.render_time()
    assert n == len(POSES) and 0.0 < ms < 50.0
    for f, out in enumerate(outs):
        assert np.array_equal(out.cpu().numpy(), ref[f]), f"frame {f}"
    assert np.array_equal(np.float32(eng.camera_position), np.float32(ref_cam))
    grp.close()
    eng.close()


class _ThreadExchange:
    """An all-gather between host threads that share one GPU: every rank waits for its stream, publishes
    its send buffer, and copies every rank's part into its own receive buffer."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=60)
        self.send = [None] * world

    def for_rank(self, rank):
        import torch

        def all_gather(d_send, d_recv, nbytes, stream):
            torch.cuda.synchronize()
            self.send[rank] = d_send
            self.barrier.wait()
            for r in range(self.world):
                _alias_u8(d_recv + r * nbytes, nbytes).copy_(_alias_u8(self.send[r], nbytes))
            torch.cuda.synchronize()
            self.barrier.wait()
            return 0

        return all_gather


@pytest.mark.parametrize("world,w,h,fxaa", [(2, 640, 360, False), (3, 640, 360, True), (3, 402, 250, False), (4, 320, 100, True)])
def test_group_ranks_as_threads_on_one_gpu(scene, world, w, h, fxaa):
    """2, 3 and 4 ranks (strip counts that do and do not divide, a width that cannot be packed, a
    height that cuts the last strip): every rank ends up with the single-engine frames and lightmaps."""
    import torch

    from raymarching_amd import Engine
    from raymarching_amd.group import Group

    ref, ref_lm, ref_cam = _reference_frames(scene, w, h, fxaa)
    ex = _ThreadExchange(world)
    engines = [Engine(scene, w, h) for _ in range(world)]
    groups = [Group(engines[r], r, world, frames_in_flight=2, all_gather=ex.for_rank(r)) for r in range(world)]
    assert groups[0].info()["bytes_per_pixel_exchanged"] == (3 if w % 4 == 0 else 4)
    outs = [[torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in POSES] for _ in range(world)]
    errors = []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            groups[r].compute_lightmaps(4)
            for (mx, my, mv, sc), out in zip(POSES, outs[r]):
                groups[r].frame(out.data_ptr(), fxaa, mx, my, mv, sc)
            groups[r].wait()
        except Exception as exc:  # noqa: BLE001
            errors.append((r, exc))
            ex.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=110)
    assert not errors, errors
    for r in range(world):
        lm = np.stack([engines[r].get_lightmap(i) for i in range(engines[r].n_objects)])
        assert np.array_equal(lm.view(np.uint32), ref_lm.view(np.uint32)), f"rank {r}: lightmaps"
        for f in range(len(POSES)):
            assert np.array_equal(outs[r][f].cpu().numpy(), ref[f]), f"rank {r} frame {f}"
        assert np.array_equal(np.float32(engines[r].camera_position), np.float32(ref_cam))
    for g in groups:
        g.close()
    for e in engines:
        e.close()


@pytest.mark.parametrize("n,w,h", [(1, 320, 240), (3, 640, 360), (4, 402, 250)])
def test_multi_engine_on_one_gpu(scene, n, w, h):
    """rm_multi_* with GPU 0 listed n times: sharded lightmaps and frames in host memory (pageable,
    page-locked, with a pitch) equal the single engine's."""
    from raymarching_amd import _lib
    from raymarching_amd.group import MultiEngine

    ref, ref_lm, ref_cam = _reference_frames(scene, w, h, False)
    multi = MultiEngine(scene, w, h, [0] * n)
    multi.compute_lightmaps(4)
    for k in range(n):
        eng = multi.engine(k)
        lm = np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])
        assert np.array_equal(lm.view(np.uint32), ref_lm.view(np.uint32)), f"engine {k}: lightmaps"
    lib = _lib.load()
    pitch = w * 4 + 64
    pinned = lib.rm_host_alloc(pitch * h)
    pinned_view = np.ctypeslib.as_array((ctypes.c_uint8 * (pitch * h)).from_address(pinned)).reshape(h, pitch)
    for f, (mx, my, mv, sc) in enumerate(POSES):
        if f % 2 == 0:
            got = multi.render(None, mx, my, mv, sc)  # pageable destination
            assert np.array_equal(got, ref[f]), f"frame {f}"
        else:
            pinned_view[:] = 0xAB
            multi.render(None, mx, my, mv, sc, host_ptr=pinned, pitch=pitch)
            assert np.array_equal(pinned_view[:, : w * 4].reshape(h, w, 4), ref[f]), f"frame {f} (page-locked, pitch)"
            assert (pinned_view[:, w * 4:] == 0xAB).all()
    assert np.array_equal(np.float32(multi.engine(0).camera_position), np.float32(ref_cam))
    lib.rm_host_free(pinned)
    multi.close()
