"""Multi-GPU decomposition: one process per GPU, torch.distributed (RCCL) between them.

Frames shard by screen band: rank r renders rows band_rows(H, r, world) and one
all-gather leaves the whole frame on every rank.  Radiosity shards by receiver
patch: each rank computes its rows of the form-factor matrix once, then per
bounce its slice of the next emissive map, and one all-gather per bounce
exchanges the slices (Jacobi: bounce k reads only bounce k-1, engine.rs:299-325).

Nothing here touches the GPU directly; callers pass tensors on whatever device
their process group uses (cuda for nccl, cpu for gloo).
"""
from typing import Callable, Tuple

import torch
import torch.distributed as dist


def band_rows(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [r0, r1) of rank `rank`: equal bands of ceil(height/world) rows, the last ones clipped."""
    band = (height + world - 1) // world
    r0 = min(rank * band, height)
    return r0, min(r0 + band, height)


def patch_range(n_patches: int, rank: int, world: int) -> Tuple[int, int]:
    """Receiver patches [p0, p1) of rank `rank` (contiguous, equal except the tail)."""
    per = (n_patches + world - 1) // world
    p0 = min(rank * per, n_patches)
    return p0, min(p0 + per, n_patches)


STRIP_ROWS = 8  # RM_STRIP_ROWS: one row of 8x8 wave tiles


def strip_plan(height: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(first_strip, strip_stride, strips_per_rank) of rank `rank`: it renders strips
    rank, rank+world, ... of 8 rows each.  Every rank gets the same count (strips past the
    frame are skipped by the renderer), so one all-gather of equal parts moves them."""
    n_strips = (height + STRIP_ROWS - 1) // STRIP_ROWS
    per_rank = (n_strips + world - 1) // world
    return rank, world, per_rank


def deinterleave_strips(gathered: torch.Tensor, world: int, per_rank: int, width: int, height: int) -> torch.Tensor:
    """[world][per_rank*8][width][4] as all-gathered -> the frame [height][width][4]:
    strip k of rank r is frame strip k*world + r."""
    g = gathered.view(world, per_rank, STRIP_ROWS, width, 4)
    frame = g.permute(1, 0, 2, 3, 4).reshape(per_rank * world * STRIP_ROWS, width, 4)
    return frame[:height]


def gather_frame_interleaved(render_strips: Callable[[int, int, int], torch.Tensor], width: int, height: int,
                             group=None) -> torch.Tensor:
    """Load-balanced variant of gather_frame: `render_strips(first, stride, n)` returns the
    compact uint8 [n*8][width][4] image of this rank's strips; the result is the whole frame."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, stride, per_rank = strip_plan(height, rank, world)
    mine = render_strips(first, stride, per_rank)
    if world == 1:
        return mine[:height]
    gathered = torch.empty((world,) + tuple(mine.shape), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(gathered.view(-1), mine.reshape(-1), group=group)
    return deinterleave_strips(gathered, world, per_rank, width, height)


def gather_frame(render_band: Callable[[int, int], torch.Tensor], width: int, height: int, group=None) -> torch.Tensor:
    """Render this rank's band with `render_band(r0, r1) -> uint8 [r1-r0][width][4]` and
    all-gather the bands into the full [height][width][4] frame on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    r0, r1 = band_rows(height, rank, world)
    band = render_band(r0, r1)
    if world == 1:
        return band
    band_h = (height + world - 1) // world
    padded = torch.zeros((band_h, width, 4), dtype=torch.uint8, device=band.device)
    padded[: r1 - r0] = band
    full = torch.empty((world * band_h, width, 4), dtype=torch.uint8, device=band.device)
    dist.all_gather_into_tensor(full.view(-1), padded.view(-1), group=group)
    return full[:height]


def _all_gather_flat(out: torch.Tensor, inp: torch.Tensor, group=None):
    """all_gather_into_tensor; with the gloo backend (CPU rehearsals) device tensors go through host memory."""
    if out.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def gather_patches(local: torch.Tensor, n_patches: int, group=None) -> torch.Tensor:
    """All-gather per-patch rows: `local` is this rank's [p1-p0][k] slice; returns [n_patches][k]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    per = (n_patches + world - 1) // world
    padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    full = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    _all_gather_flat(full.view(-1), padded.view(-1), group)
    return full[:n_patches]


def compute_lightmaps_sharded(engine, bounces: int = 4, group=None):
    """Engine::compute_lightmaps (engine.rs:246-326) with receiver patches split over the
    ranks of `group`.  `engine` is this rank's raymarching_amd.Engine; tensors alias its
    device buffers, so the gathers write straight into them."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    engine.radiosity_begin()
    em_ptr, lm_ptr, n = engine.radiosity_buffers()
    p0, p1 = patch_range(n, rank, world)
    if bounces > 0:
        engine.radiosity_form_factors(p0, p1)
    if world == 1:
        for _ in range(bounces):
            engine.radiosity_bounce(p0, p1)
            engine.radiosity_commit()
        engine.synchronize()
        return
    emissive = _alias_f32(em_ptr, n * 3, engine.device).view(n, 3)
    lightmaps = _alias_f32(lm_ptr, n * 3, engine.device).view(n, 3)
    for _ in range(bounces):
        engine.radiosity_bounce(p0, p1)
        engine.synchronize()
        emissive.copy_(gather_patches(emissive[p0:p1].clone(), n, group))
        # the gathered values were written on torch's current stream, the commit reads them on the
        # engine's: whatever the two are, the copy has to have landed first
        torch.cuda.current_stream().synchronize()
        engine.radiosity_commit()
    engine.synchronize()
    lightmaps.copy_(gather_patches(lightmaps[p0:p1].clone(), n, group))
    torch.cuda.current_stream().synchronize()
    engine.synchronize()


def _alias_f32(ptr: int, count: int, device: int) -> torch.Tensor:
    """A float32 CUDA tensor over `count` floats of existing device memory at `ptr`."""

    class _Mem:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}

    return torch.as_tensor(_Mem(), device=torch.device("cuda", device))


def _alias_u8(ptr: int, count: int, device: int) -> torch.Tensor:
    """A uint8 CUDA tensor over `count` bytes of existing device memory at `ptr`."""

    class _Mem:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    return torch.as_tensor(_Mem(), device=torch.device("cuda", device))
