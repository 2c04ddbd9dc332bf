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


@pytest.mark.parametrize("fxaa,exchange", [(False, "all_gather"), (True, "all_gather"), (False, "presenter"), (True, "presenter")])
def test_group_over_rccl_with_one_rank(scene, fxaa, exchange):
    """The RCCL path end to end (library lookup, communicator, one more split off it per frame in flight,
    all-gather or gather to the presenter on the frame's stream, frames in flight) with the only
    communicator a one-GPU box allows."""
    import torch

    from raymarching_amd import Engine
    from raymarching_amd.group import Group, unique_id

    w, h = 640, 360
    ref, ref_lm, ref_cam = _reference_frames(scene, w, h, fxaa)
    eng = Engine(scene, w, h)
    grp = Group(eng, 0, 1, unique_id(), frames_in_flight=3)
    assert grp.info() == {"rank": 0, "world": 1, "strips_per_rank": 45, "bytes_per_pixel_exchanged": 3}
    assert grp.communicators() == 3
    grp.compute_lightmaps(4)
    lm = np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])
    assert np.array_equal(lm.view(np.uint32), ref_lm.view(np.uint32))
    grp.set_exchange(exchange)
    assert grp.next_presenter() == (0 if exchange == "presenter" else -1)
    outs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in POSES]
    grp.set_timing(True)
    for (mx, my, mv, sc), out in zip(POSES, outs):
        grp.frame(out.data_ptr(), fxaa, mx, my, mv, sc)
    grp.wait()
    ms, n = grp.render_time()
    assert n == len(POSES) and 0.0 < ms < 50.0
    xms, xn = grp.exchange_time()
    assert xn == len(POSES) and 0.0 < xms < 50.0
    for f, out in enumerate(outs):
        assert np.array_equal(out.cpu().numpy(), ref[f]), f"frame {f}"
    assert np.array_equal(np.float32(eng.camera_position), np.float32(ref_cam))
    grp.close()
    eng.close()


def test_group_leaves_the_engine_as_it_found_it(scene):
    """A group retunes how its engine launches frames (tile splitting, the 8-wave build) and takes
    launch-plan slots for its streams: both are given back when it goes, also after many groups."""
    import torch

    from raymarching_amd import Engine
    from raymarching_amd.group import Group, unique_id

    w, h = 1920, 1080
    eng = Engine(scene, w, h)
    eng.compute_lightmaps(4)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    eng.render_rows_device(out.data_ptr(), 0, h)
    eng.render_rows_device(out.data_ptr(), 0, h)
    eng.render_rows_device(out.data_ptr(), 0, h)
    before = out.cpu().numpy().copy()
    listed_before = eng.split_tiles()
    assert listed_before > 0  # the mirror-sphere tiles of the default scene are slow
    for _ in range(4):  # more groups than the engine has plan slots for their streams
        grp = Group(eng, 0, 1, unique_id(), frames_in_flight=3)
        for _ in range(4):
            grp.frame(out.data_ptr(), False)
        grp.wait()
        assert np.array_equal(out.cpu().numpy(), before)
        grp.close()
    for _ in range(3):
        eng.render_rows_device(out.data_ptr(), 0, h)
    assert eng.split_tiles() > 0, "tile splitting was left switched off"
    assert np.array_equal(out.cpu().numpy(), before)
    # closing the engine first closes the group with it
    grp = Group(eng, 0, 1, unique_id(), frames_in_flight=2)
    eng.close()
    assert grp.handle is None
    grp.close()


class _ThreadExchange:
    """An all-gather (and a gather to one rank) between host threads that share one GPU: every rank
    waits for its stream, publishes its send buffer, and copies every rank's part into its own
    receive buffer."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=100)
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

    def gather_for_rank(self, rank):
        import torch

        def gather(d_send, d_recv, nbytes, root, stream):
            torch.cuda.synchronize()
            self.send[rank] = d_send
            self.barrier.wait()
            assert (d_recv is not None) == (rank == root)
            if rank == root:
                for r in range(self.world):
                    _alias_u8(d_recv + r * nbytes, nbytes).copy_(_alias_u8(self.send[r], nbytes))
            torch.cuda.synchronize()
            self.barrier.wait()
            return 0

        return gather


def _run_ranks(world, rank_main, ex, join_timeout=110):
    errors = []

    def guarded(r):
        try:
            rank_main(r)
        except Exception as exc:  # noqa: BLE001
            errors.append((r, exc))
            ex.barrier.abort()

    threads = [threading.Thread(target=guarded, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=join_timeout)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a rank did not finish"


@pytest.mark.parametrize("fxaa", [False, True])
def test_eight_ranks_at_3840x2160_three_frames_in_flight(scene, fxaa):
    """The benchmark's eight-GPU shape on one GPU: 270 strips over 8 ranks are 34 each with the last
    two of rank 6's and 7's beyond the frame, 3 bytes per pixel on the wire, three frames in flight -
    first through the all-gather (every rank must hold every frame), then with the frames gathered to a
    rotating presenter (frame f on rank f mod 8 alone, the other ranks' buffers untouched)."""
    import torch

    from raymarching_amd import Engine
    from raymarching_amd.group import Group

    world, w, h = 8, 3840, 2160
    ref, ref_lm, ref_cam = _reference_frames(scene, w, h, fxaa)
    ex = _ThreadExchange(world)
    engines = [Engine(scene, w, h) for _ in range(world)]
    groups = [Group(engines[r], r, world, frames_in_flight=3, all_gather=ex.for_rank(r), gather=ex.gather_for_rank(r)) for r in range(world)]
    assert groups[7].info() == {"rank": 7, "world": 8, "strips_per_rank": 34, "bytes_per_pixel_exchanged": 3}
    # one set of frame buffers per rank, reused by both exchanges; a presenter frame is compared on its rank
    outs = [[torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in POSES] for _ in range(world)]
    ref_dev = [torch.from_numpy(f).cuda() for f in ref]

    def all_gather_main(r):
        torch.cuda.set_device(0)
        groups[r].compute_lightmaps(4)
        for (mx, my, mv, sc), out in zip(POSES, outs[r]):
            groups[r].frame(out.data_ptr(), fxaa, mx, my, mv, sc)
        groups[r].wait()

    _run_ranks(world, all_gather_main, ex)
    for r in range(world):
        lm = np.stack([engines[r].get_lightmap(i) for i in range(engines[r].n_objects)])
        assert np.array_equal(lm.view(np.uint32), ref_lm.view(np.uint32)), f"rank {r}: lightmaps"
        for f in range(len(POSES)):
            assert torch.equal(outs[r][f], ref_dev[f]), f"all-gather: rank {r} frame {f}"
            outs[r][f].zero_()
        assert np.array_equal(np.float32(engines[r].camera_position), np.float32(ref_cam))

    # the same poses again, gathered to a rotating presenter; 11 frames, so that the rotation wraps
    # and slots, presenters and poses fall out of step with each other
    n_frames = 11
    landed = [[None] * n_frames for _ in range(world)]

    def presenter_main(r):
        torch.cuda.set_device(0)
        engines[r].camera_position = scene.camera_position  # main.rs:66: every rank restarts the recorded path
        groups[r].set_exchange("presenter")
        for f in range(n_frames):
            mx, my, mv, sc = POSES[f % len(POSES)]
            p = groups[r].next_presenter()
            assert p == f % world
            if p == r:
                landed[r][f] = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
            groups[r].frame(landed[r][f].data_ptr() if p == r else 0, fxaa, mx, my, mv, sc)
        groups[r].wait()

    # what a single engine renders along the same 11-frame path
    one = Engine(scene, w, h)
    one.compute_lightmaps(4)
    raw = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    filt = torch.zeros_like(raw)
    want = []
    for f in range(n_frames):
        one.input(*POSES[f % len(POSES)])
        one.render_rows_device(raw.data_ptr(), 0, h)
        if fxaa:
            one.fxaa_device(raw.data_ptr(), filt.data_ptr(), 0, h)
        one.synchronize()
        want.append((filt if fxaa else raw).clone())
    one.close()

    _run_ranks(world, presenter_main, ex)
    for f in range(n_frames):
        for r in range(world):
            if r == f % world:
                assert torch.equal(landed[r][f], want[f]), f"presenter: frame {f} on rank {r}"
            else:
                assert landed[r][f] is None
    for g in groups:
        g.close()
    for e in engines:
        e.close()


def test_an_exception_in_the_hosts_exchange_comes_back_out(scene):
    import torch

    from raymarching_amd import Engine, _lib
    from raymarching_amd.group import Group

    def broken(d_send, d_recv, nbytes, stream):
        raise ValueError("the transport fell over")

    eng = Engine(scene, 64, 64)
    grp = Group(eng, 0, 1, frames_in_flight=1, all_gather=broken)
    out = torch.zeros((64, 64, 4), dtype=torch.uint8, device="cuda")
    with pytest.raises(_lib.RMError) as info:
        grp.frame(out.data_ptr())
    assert isinstance(info.value.__cause__, ValueError) and "fell over" in str(info.value.__cause__)
    grp.close()
    eng.close()


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

    def rank_main(r):
        torch.cuda.set_device(0)
        groups[r].compute_lightmaps(4)
        for (mx, my, mv, sc), out in zip(POSES, outs[r]):
            groups[r].frame(out.data_ptr(), fxaa, mx, my, mv, sc)
        groups[r].wait()

    _run_ranks(world, rank_main, ex)
    for r in range(world):
        lm = np.stack([engines[r].get_lightmap(i) for i in range(engines[r].n_objects)])
        assert np.array_equal(lm.view(np.uint32), ref_lm.view(np.uint32)), f"rank {r}: lightmaps"
        for f in range(len(POSES)):
            assert np.array_equal(outs[r][f].cpu().numpy(), ref[f]), f"rank {r} frame {f}"
        assert np.array_equal(np.float32(engines[r].camera_position), np.float32(ref_cam))

    # the same frames gathered to a rotating presenter, by a host that supplied an all-gather only
    # (every rank receives, the presenter alone goes on to use what it received)
    landed = [[None] * len(POSES) for _ in range(world)]

    def presenter_main(r):
        torch.cuda.set_device(0)
        engines[r].camera_position = scene.camera_position
        groups[r].set_exchange("presenter")
        for f, (mx, my, mv, sc) in enumerate(POSES):
            if f % world == r:
                landed[r][f] = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
            groups[r].frame(landed[r][f].data_ptr() if f % world == r else 0, fxaa, mx, my, mv, sc)
        groups[r].wait()

    _run_ranks(world, presenter_main, ex)
    for f in range(len(POSES)):
        assert np.array_equal(landed[f % world][f].cpu().numpy(), ref[f]), f"presenter: frame {f}"
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
