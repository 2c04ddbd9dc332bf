"""Two-rank CPU rehearsal (gloo) of the multi-GPU decomposition: band-sharded frames and
patch-sharded exchanges assemble to exactly the single-process result.  The per-rank work
is done by the CPU oracle here; on the GPU the same helpers run over RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 96, 70  # height not divisible by the world size: the last band is short


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from raymarching_amd import dist as rdist
    from raymarching_amd.scene import default_scene

    scene = default_scene()
    ora = oracle.OracleEngine(scene)
    ora.compute_lightmaps(1)

    def render_band(r0, r1):
        return torch.from_numpy(ora.render(W, H, 40, -20, (0.0, 0.0, 0.0), 1, rows=(r0, r1), threads=2))

    frame = rdist.gather_frame(render_band, W, H)
    np.save(os.path.join(out_dir, f"frame_{rank}.npy"), frame.numpy())

    def render_strips(first, stride, n):
        out = torch.zeros((n * 8, W, 4), dtype=torch.uint8)
        for k in range(n):
            r0 = (first + k * stride) * 8
            if r0 >= H:
                break
            r1 = min(r0 + 8, H)
            out[k * 8 : k * 8 + (r1 - r0)] = render_band(r0, r1)
        return out

    frame2 = rdist.gather_frame_interleaved(render_strips, W, H)
    np.save(os.path.join(out_dir, f"frame_strips_{rank}.npy"), frame2.numpy())

    # patch-sharded exchange: every rank owns a slice of a [n_patches][3] array
    n = 8 * 8 * 8
    full = torch.arange(n * 3, dtype=torch.float32).view(n, 3) * 0.5
    p0, p1 = rdist.patch_range(n, rank, world)
    got = rdist.gather_patches(full[p0:p1].clone(), n)
    assert torch.equal(got, full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_and_patch_gather(tmp_path, scene):
    import oracle

    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ora = oracle.OracleEngine(scene)
    ora.compute_lightmaps(1)
    ref = ora.render(W, H, 40, -20, (0.0, 0.0, 0.0), 1)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"frame_{r}.npy"))
        assert got.shape == ref.shape and np.array_equal(got, ref)
        got2 = np.load(os.path.join(str(tmp_path), f"frame_strips_{r}.npy"))
        assert got2.shape == ref.shape and np.array_equal(got2, ref)


def test_strip_plan_covers_every_strip_once():
    from raymarching_amd import dist as rdist

    for height, world in [(2160, 8), (2160, 3), (70, 2), (600, 7), (8, 4)]:
        n_strips = (height + 7) // 8
        seen = []
        counts = set()
        for r in range(world):
            first, stride, per = rdist.strip_plan(height, r, world)
            counts.add(per)
            seen += [first + k * stride for k in range(per) if first + k * stride < n_strips]
        assert sorted(seen) == list(range(n_strips)) and len(counts) == 1


def test_deinterleave_strips():
    from raymarching_amd import dist as rdist

    world, per, w, h = 3, 4, 5, 90  # 12 strip slots for 11.25 strips of rows
    frame = torch.arange(per * world * 8 * w * 4, dtype=torch.int64).remainder(251).to(torch.uint8).view(per * world * 8, w, 4)
    gathered = torch.empty((world, per * 8, w, 4), dtype=torch.uint8)
    for r in range(world):
        for k in range(per):
            s = k * world + r
            gathered[r, k * 8 : (k + 1) * 8] = frame[s * 8 : (s + 1) * 8]
    out = rdist.deinterleave_strips(gathered, world, per, w, h)
    assert out.shape == (h, w, 4) and torch.equal(out, frame[:h])
