#!/usr/bin/env python3
"""Headline benchmark: primary Mrays/s of the default scene at 3840x2160 on N MI355X.

One "step" is one frame: every primary ray of the 4K image through march, shading
(texture, lightmap, Phong, soft shadow, reflections), tone map and pack, written
as B,G,R,0 bytes into HBM.  With N > 1 the frame is split into N row bands, one
per rank, and the bands are all-gathered over RCCL so that every rank ends up with
the whole frame (strong scaling: the work per frame is fixed).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak (counts an FMA as 2 flops)
PEAK_FP32_NO_FMA_TFLOPS = 78.65  # same issue rate without fused multiply-add


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--fxaa", action="store_true", help="add the FXAA pass to every frame")
    ap.add_argument("--ssaa", action="store_true", help="four rays per pixel (the supersampling the reference carries commented out)")
    ap.add_argument("--refraction", action="store_true", help="the refraction block the reference carries commented out (engine.rs:196-228)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time budget of the CPU baseline leg")
    ap.add_argument("--no-pack", action="store_true", help="exchange 4 bytes per pixel instead of 3 (the fourth byte of a pixel is always 0)")
    ap.add_argument("--no-overlap", action="store_true", help="wait for each frame's all-gather before the next render")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="1 = frames one after the other; n = consecutive frames on n streams, so that the long-running "
                         "tiles at the end of one frame overlap the start of the next ones (default: 1 on one GPU, 3 on several)")
    ap.add_argument("--workload", choices=["render", "radiosity", "stress"], default="render",
                    help="render: the headline frame benchmark; radiosity: Engine::compute_lightmaps, patch-sharded; "
                    "stress: 256 mixed primitives with smooth unions at 7680x4320, march cap 1024")
    ap.add_argument("--map-size", type=int, default=8, help="lightmap edge for --workload radiosity (8 in the reference)")
    return ap.parse_args()


def load_counters(width, height):
    path = os.path.join(ROOT, "tests", "golden", "counters.json")
    try:
        with open(path) as f:
            return json.load(f).get(f"frame_pose0_{width}x{height}")
    except OSError:
        return None


def measured_traffic(width, height):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC passes (4K only)."""
    if (width, height) != (3840, 2160):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "round1_traffic.json")) as f:
            return int(json.load(f)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(scene, lightmaps, width, height, budget_s, ssaa=False, refraction=False):
    """Times the CPU oracle (test infrastructure) on whole frames of the same workload
    until about `budget_s` seconds of CPU work have been spent (at least 3 frames)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    e = oracle.OracleEngine(scene)
    for i in range(e.n_objects):
        e.set_lightmap(i, lightmaps[i])
    if ssaa:
        e.set_ssaa(4)
    e.set_refraction(refraction)
    e.render(width, height, row_step=16, threads=threads)  # warm-up: thread pool, caches
    times = []
    frame = None
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        frame = e.render(width, height, threads=threads)
        times.append(time.perf_counter() - t0)
        if len(times) >= 200:
            break
    e.close()
    times.sort()
    median = times[len(times) // 2]
    rays = width * height * (4 if ssaa else 1)
    return {
        "value": rays / median / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{len(times)} whole {width}x{height} frames ({rays} primary rays each), median {median * 1e3:.1f} ms/frame, "
        f"{sum(times):.1f} s of wall time, OpenMP over rows as the reference's rayon",
    }, frame


def radiosity_workload(args, world, rank, local_rank, backend):
    """One step = Engine::compute_lightmaps (engine.rs:246-326): clouds, direct lighting, the
    form-factor matrix of all patch pairs, 4 bounces; receiver patches sharded over the ranks with
    one all-gather per bounce."""
    import torch
    import torch.distributed as dist

    from raymarching_amd import Engine, default_scene
    from raymarching_amd.dist import compute_lightmaps_sharded

    M = args.map_size
    eng = Engine(default_scene(), 64, 64, map_size=M, device=local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    steps, warmup = min(args.steps, 20), min(args.warmup, 3)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        compute_lightmaps_sharded(eng, 4)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        compute_lightmaps_sharded(eng, 4)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n = 8 * M * M
        pairs = n * (n - M * M)
        print(json.dumps({
            "metric": "radiosity occlusion marches per second (Engine::compute_lightmaps, 4 bounces)",
            "value": pairs / (elapsed / steps) / 1e6,
            "unit": "Mmarches/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: the reference's default scene",
            "config": {"workload": f"compute_lightmaps, default scene, MAP_SIZE={M}, 4 bounces", "map_size": M,
                       "patches": n, "patch_pairs": pairs, "sharding": "receiver patches, all-gather per bounce"},
        }), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the ray-marching kernels need an MI355X", file=sys.stderr)
        sys.exit(2)
    # Rehearsal switches (not used by the driver): RM_BENCH_BACKEND=gloo RM_BENCH_SAME_DEVICE=1 run the
    # N-rank code path with every rank on GPU 0, for boxes with a single GPU (RCCL refuses that).
    backend = os.environ.get("RM_BENCH_BACKEND", "nccl")
    if os.environ.get("RM_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # RM_BENCH_GATHER_ALONE=1 (rehearsal, not used by the driver): one rank forms an RCCL group by itself
    # and runs the several-GPU path - strips, asynchronous all-gather, re-ordering, frames in flight -
    # on the real backend; the exchange then is a copy.
    gathering = world > 1 or os.environ.get("RM_BENCH_GATHER_ALONE") == "1"
    if gathering and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if gathering:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from raymarching_amd import Engine, default_scene
    from raymarching_amd.dist import STRIP_ROWS, band_rows, deinterleave_strips, strip_plan
    from raymarching_amd.flops import frame_flops

    if args.workload == "radiosity":
        radiosity_workload(args, world, rank, local_rank, backend)
        return

    W, H = args.width, args.height
    scene = default_scene()
    if args.workload == "stress":
        from raymarching_amd.scene import stress_scene

        if (W, H) == (3840, 2160):
            W, H = 7680, 4320
        scene = stress_scene(250, boxes=True, smooth_union=0.08)
        args.no_cpu_baseline = True  # minutes per frame on the CPU; parity is covered by the GPU tests at small sizes
    eng = Engine(scene, W, H, device=local_rank)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    if args.workload == "stress":
        # boxes have no lightmap parameterisation: a fixed smooth pattern stands in for the precompute
        u, v = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
        lm = np.stack([0.10 + 0.02 * u, 0.12 + 0.01 * v, 0.08 + 0.01 * (u + v)], axis=-1).astype(np.float32)
        from raymarching_amd.scene import Box

        for i, obj in enumerate(scene.objects):
            if not isinstance(obj, Box):
                eng.set_lightmap(i, lm)
        eng.set_march_cap(1024)
    else:
        eng.compute_lightmaps(4)  # main.rs:73: once, before the frame loop (not timed)
    eng.input(0, 0, (0.0, 0.0, 0.0), 0)  # first-frame pose (main.rs:78-80,122)
    if args.ssaa:
        eng.set_ssaa(4)
    if args.refraction:
        eng.set_refraction(True)

    # Sharding: without FXAA rank r renders the 8-row strips r, r+N, r+2N, ... (even load however
    # the cost is spread over the image); with FXAA it renders one contiguous band plus the few rows
    # the filter reads around it.  Either way every rank contributes the same number of rows.
    interleaved = gathering and not args.fxaa
    if interleaved:
        first_strip, strip_stride, strips_per_rank = strip_plan(H, rank, world)
        band_h = strips_per_rank * STRIP_ROWS
        my_rows = sum(min(STRIP_ROWS, H - (first_strip + k * strip_stride) * STRIP_ROWS)
                      for k in range(strips_per_rank) if (first_strip + k * strip_stride) * STRIP_ROWS < H)
        r0, r1 = 0, my_rows  # only the row count is used below
    else:
        band_h = (H + world - 1) // world
        r0, r1 = band_rows(H, rank, world)
    # A rank's share of a frame is bounded by its few slowest tiles, not by its size, so several
    # frames are kept in flight: measured on one GPU, an eighth of the 3840x2160 frame takes 0.132 ms
    # alone, 0.066 ms per frame with two in flight and 0.045 ms with three (0.039 would be an eighth
    # of the whole frame's time); a fourth stream gains nothing.
    in_flight = args.frames_in_flight or (3 if gathering else 1)
    if args.no_overlap:
        in_flight = 1
    n_buf = max(in_flight, 2)  # one set of buffers per frame in flight (two at least: gather f beside render f+1)
    # Rows cross xGMI as 3 bytes per pixel (byte 3 of a rendered pixel is 0, engine.rs:116-126): a rank
    # packs its rows before the all-gather and one kernel unpacks and orders the gathered rows.
    packing = gathering and W % 4 == 0 and not args.no_pack
    wire = 3 if packing else 4
    bands = [torch.zeros((band_h, W, 4), dtype=torch.uint8, device="cuda") for _ in range(n_buf)]
    wire_bands = [torch.zeros((band_h, W, 3), dtype=torch.uint8, device="cuda") for _ in range(n_buf)] if packing else bands
    frames = [torch.zeros((world * band_h, W, wire), dtype=torch.uint8, device="cuda") for _ in range(n_buf)]
    final = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(n_buf)] if interleaved or packing else None
    fx_srcs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(n_buf)] if args.fxaa else None

    kernel_events = []
    # frame s is rendered, gathered and put in order on streams[s % in_flight]
    streams = [stream] + [torch.cuda.Stream() for _ in range(in_flight - 1)]

    def render_band(buf, timed):
        fx_src = fx_srcs[next(i for i, b in enumerate(bands) if b is buf)] if args.fxaa else None  # one per frame in flight
        stream = torch.cuda.current_stream()
        eng.set_stream(stream.cuda_stream)
        if timed:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
        if args.fxaa:
            # band plus the rows FXAA reads around it (one above, four below)
            h0, h1 = max(r0 - 1, 0), min(r1 + 4, H)
            eng.render_rows_device(fx_src[h0].data_ptr(), h0, h1)
            if timed:
                ev1.record(stream)
            # the band buffer stands in for rows r0.. of a frame-shaped destination
            eng.fxaa_device(fx_src.data_ptr(), buf.data_ptr() - r0 * W * 4, r0, r1)
        elif interleaved:
            eng.render_strips_device(buf.data_ptr(), first_strip, strip_stride, strips_per_rank)
            if timed:
                ev1.record(stream)
        else:
            eng.render_rows_device(buf.data_ptr(), r0, r1)
            if timed:
                ev1.record(stream)
        if timed:
            kernel_events.append((ev0, ev1))
        if packing:
            eng.pack_bgr_device(buf.data_ptr(), wire_bands[next(i for i, b in enumerate(bands) if b is buf)].data_ptr(), band_h * W)

    class _HostStagedGather:
        """gloo rehearsal only: all-gather through host memory, same interface as an async work handle."""

        def __init__(self, out, inp):
            cpu_in = inp.cpu()
            cpu_out = torch.empty(out.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(cpu_out, cpu_in)
            out.copy_(cpu_out)

        def wait(self):
            pass

    def start_gather(out, inp):
        if backend == "nccl":
            return dist.all_gather_into_tensor(out, inp, async_op=True)
        return _HostStagedGather(out, inp)

    def finish(pending):
        """Frame f is complete on this rank once its gather has landed (and, for strips, been put in row order)."""
        work, b = pending
        work.wait()
        if packing:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            if interleaved:
                eng.unpack_bgr_strips_device(frames[b].data_ptr(), final[b].data_ptr(), world, strips_per_rank)
            else:  # contiguous bands are in row order already
                eng.unpack_bgr_strips_device(frames[b].data_ptr(), final[b].data_ptr(), 1, (H + STRIP_ROWS - 1) // STRIP_ROWS)
        elif interleaved:
            final[b].copy_(deinterleave_strips(frames[b], world, strips_per_rank, W, H))

    def run(steps, timed):
        if in_flight == 1:
            pending = None
            for s in range(steps):
                b = s & 1
                render_band(bands[b], timed)
                if gathering:
                    if pending is not None:
                        finish(pending)
                    pending = (start_gather(frames[b].view(-1), wire_bands[b].view(-1)), b)
                    if args.no_overlap:
                        finish(pending)
                        pending = None
            if pending is not None:
                finish(pending)
            return
        # Several frames in flight: frame s lives on streams[s % in_flight] from its render to the
        # re-ordering of its gathered strips, so the renders of the next frames run beside the tail of
        # frame s.  Buffer set b is reused by frame s + in_flight on the same stream, after frame s has
        # been finished there.
        pending = [None] * in_flight
        for st in streams:
            st.wait_stream(stream)
        for s in range(steps):
            b = s % in_flight
            with torch.cuda.stream(streams[b]):
                if pending[b] is not None:
                    finish(pending[b])
                    pending[b] = None
                render_band(bands[b], timed)
                if gathering:
                    pending[b] = (start_gather(frames[b].view(-1), wire_bands[b].view(-1)), b)
        for s in range(max(steps - in_flight, 0), steps):  # oldest frame first
            b = s % in_flight
            if pending[b] is not None:
                with torch.cuda.stream(streams[b]):
                    finish(pending[b])
                pending[b] = None
        for st in streams[1:]:
            stream.wait_stream(st)
        eng.set_stream(stream.cuda_stream)

    def fence():
        torch.cuda.synchronize()
        if gathering:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup, False)
    fence()
    t0 = time.perf_counter()
    run(args.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / max(len(kernel_events), 1)
    stats = eng.stats()

    result = None
    last_buf = (args.steps - 1) % in_flight if in_flight > 1 else (args.steps - 1) & 1  # where the last frame landed
    if rank == 0:
        rays = W * H * (4 if args.ssaa else 1)
        ms_per_step = elapsed / args.steps * 1e3
        counters = load_counters(W, H) if args.workload == "render" and not args.ssaa and not args.refraction else None
        roofline = None
        if counters is not None:
            band_share = (r1 - r0) / H  # flops are counted for the whole frame; a band is its share by rows
            flops = frame_flops(counters)
            achieved = flops * band_share / (kernel_ms * 1e-3) / 1e12
            roofline = {
                "bound": "valu",
                "achieved": achieved,
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS,
                "frac_of_no_fma_peak": achieved / PEAK_FP32_NO_FMA_TFLOPS,
                "traffic": int(measured_traffic(W, H) * band_share) if measured_traffic(W, H) else None,
                "traffic_source": "rocprofv3 PMC passes committed under profiles/ (not collected live)",
                "kernel": "render_kernel",
                "kernel_ms": kernel_ms,
                "algorithmic_flops_per_launch": int(flops * band_share),
                "note": "FP32 vector ALU bound (no MFMA, HBM traffic is 4 B per ray); strict IEEE parity forbids FMA contraction",
            }
        result = {
            "metric": "primary Mrays/s at 4K" if (W, H) == (3840, 2160) and args.workload == "render" else f"primary Mrays/s at {W}x{H}",
            "value": rays / (elapsed / args.steps) / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: the reference's default scene (main.rs:154-193), first-frame pose, lightmaps from 4 radiosity bounces",
            "config": {
                "workload": (f"default scene {W}x{H}, march + shading + texture + lightmap + reflections" if args.workload == "render"
                             else f"stress scene: 250 spheres and boxes + room, smooth union k=0.08, {W}x{H}, march cap 1024")
                + (" + FXAA" if args.fxaa else "") + (", 4 rays per pixel" if args.ssaa else "") + (", refraction on" if args.refraction else ""),
                "width": W,
                "height": H,
                "bands": world,
                "sharding": "interleaved 8-row strips" if interleaved else "contiguous row bands",
                "fxaa": bool(args.fxaa),
                "ssaa": 4 if args.ssaa else 1,
                "refraction": bool(args.refraction),
                "map_size": 8,
                "overlap_gather": gathering and not args.no_overlap,
                "bytes_per_pixel_exchanged": wire if gathering else None,
                "frames_in_flight": in_flight,
            },
            "roofline": roofline,
            "cap_hits": stats,
        }
        if gathering:
            # the gathered frame against the same frame rendered (and filtered) whole on this GPU, bit for bit
            gathered = (final if final is not None else frames)[last_buf][:H]
            local = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            eng.render_rows_device(local.data_ptr(), 0, H)
            if args.fxaa:
                filtered = torch.zeros_like(local)
                eng.fxaa_device(local.data_ptr(), filtered.data_ptr(), 0, H)
                local = filtered
            torch.cuda.synchronize()
            result["gathered_frame_equals_single_gpu_render"] = bool(torch.equal(gathered, local))
        if not args.no_cpu_baseline and not gathering:  # the CPU leg runs at N=1 only
            lightmaps = [eng.get_lightmap(i) for i in range(eng.n_objects)]
            base, cpu_rows = cpu_baseline(scene, lightmaps, W, H, args.cpu_seconds, args.ssaa, args.refraction)
            result["cpu_baseline"] = base
            if not args.fxaa:
                # the same frame from the GPU: a parity check at the benchmarked size
                full = (final if final is not None else frames)[last_buf] if gathering else None
                if full is None:
                    full = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
                    eng.render_rows_device(full.data_ptr(), 0, H)
                    torch.cuda.synchronize()
                gpu_rows = full[:H].cpu().numpy()
                diff = np.abs(gpu_rows.astype(np.int16) - cpu_rows.astype(np.int16))
                result["parity_vs_cpu_frame"] = {
                    "max_abs_diff": int(diff.max()),
                    "exact_pixel_fraction": float((diff.reshape(-1, 4).max(axis=1) == 0).mean()),
                }
        print(json.dumps(result), flush=True)

    eng.close()
    if gathering:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
