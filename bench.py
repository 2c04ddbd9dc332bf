#!/usr/bin/env python3
"""Headline benchmark: primary Mrays/s of the default scene at 3840x2160 on N MI355X.

One "step" is one frame: every primary ray of the 4K image through march, shading
(texture, lightmap, Phong, soft shadow, reflections), tone map and pack, written
as B,G,R,0 bytes into HBM.  With N > 1 the frame is split into interleaved 8-row
strips, rank r rendering strips r, r+N, ..., and one all-gather per frame (RCCL,
issued inside the library by rm_group_frame) leaves the whole frame on every rank
(strong scaling: the work per frame is fixed).

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
    ap.add_argument("--no-overlap", action="store_true", help="wait for each frame of the group before the next is issued")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="1 = frames one after the other; n = consecutive frames on n streams, so that the long-running "
                         "tiles at the end of one frame overlap the start of the next ones (default: 1 on one GPU, 3 on several)")
    ap.add_argument("--workload", choices=["render", "radiosity", "stress"], default="render",
                    help="render: the headline frame benchmark; radiosity: Engine::compute_lightmaps, patch-sharded; "
                    "stress: 256 mixed primitives with smooth unions at 7680x4320, march cap 1024")
    ap.add_argument("--map-size", type=int, default=8, help="lightmap edge for --workload radiosity (8 in the reference)")
    ap.add_argument("--exchange", choices=["all_gather", "presenter"], default="all_gather",
                    help="several GPUs: where a frame ends up in the timed region - on every rank (one all-gather per frame, the "
                         "default) or on rank f mod N alone (the strips gathered to a rotating presenter); the other one is "
                         "measured after the timed region and reported beside it")
    ap.add_argument("--preroll-seconds", type=float, default=0.3,
                    help="untimed frames rendered before --warmup so that the timed region does not sit on the clock ramp")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements after the timed region")
    ap.add_argument("--fast", action="store_true",
                    help="the frame kernel's fast tier (rm_engine_set_precision: sight-ray chain strict, colour arithmetic relaxed; "
                         "within 2/255 and 99.9 %% identical pixels of the reference instead of bit-identical)")
    return ap.parse_args()


def load_counters(width, height):
    path = os.path.join(ROOT, "tests", "golden", "counters.json")
    try:
        with open(path) as f:
            return json.load(f).get(f"frame_pose0_{width}x{height}")
    except OSError:
        return None


def stress_roofline(width, height, kernel_ms):
    """The stress frame against the FP32 vector peak.  The numerator of `achieved` is what the kernel
    executed: vector lane-operations per launch from the committed rocprofv3 PMC pass
    (SQ_THREAD_CYCLES_VALU), each counted as one flop, divided by the kernel time measured live.  The
    algorithmic flops - every object evaluated at every step, as the reference's loop does - are counted
    by the oracle on a sample of rows and reported beside it (`algorithmic`): the spatial index
    legitimately avoids nearly all of them, so they say nothing about how busy the vector units are."""
    try:
        pmc = None
        for name in ("round3_stress_pmc.json", "round2_stress_pmc.json"):
            if os.path.exists(os.path.join(ROOT, "profiles", name)):
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pmc = json.load(f)
                break
        if pmc is None:
            return None
    except (OSError, ValueError):
        return None
    if (width, height) != (pmc.get("width"), pmc.get("height")) or not kernel_ms:
        return None
    achieved = pmc["valu_lane_ops_per_launch"] / (kernel_ms * 1e-3) / 1e12
    algorithmic = None
    try:
        from raymarching_amd.flops import frame_flops

        with open(os.path.join(ROOT, "tests", "golden", "counters.json")) as f:
            c = json.load(f)[f"stress_rows_{width}x{height}"]
        per_ray = frame_flops(c, smooth_union=True) / c["pixels"]
        algorithmic = {
            "flops_per_primary_ray": per_ray,
            "flops_per_launch": per_ray * width * height,
            "tflops_at_this_frame_time": per_ray * width * height / (kernel_ms * 1e-3) / 1e12,
            "sample": f"the CPU oracle's counting build on rows {c['rows']} of the frame ({c['pixels']} pixels)",
            "note": "the flops of the reference's algorithm, which evaluates every object at every step (ray.rs:19-40: 256 distances and a "
                    "smooth-minimum fold per step); the kernel's spatial index proves most of them irrelevant and never computes them, "
                    "so this rate exceeds the machine's peak and measures the index, not the vector units",
        }
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass
    return {
        "bound": "valu",
        "achieved": achieved,
        "peak": PEAK_FP32_TFLOPS,
        "unit": "TFLOP/s",
        "frac": achieved / PEAK_FP32_TFLOPS,
        "frac_of_no_fma_peak": achieved / PEAK_FP32_NO_FMA_TFLOPS,
        "traffic": pmc.get("hbm_bytes_per_launch"),
        "traffic_source": "rocprofv3 PMC passes committed under profiles/ (not collected live)",
        "kernel": "render_kernel<GroupedLarge>",
        "kernel_ms": kernel_ms,
        "executed_lane_ops_per_launch": pmc["valu_lane_ops_per_launch"],
        "valu_instructions_per_launch": pmc.get("valu_instructions_per_launch"),
        "active_lanes_per_valu_instruction": pmc.get("active_lanes_per_valu_instruction"),
        "algorithmic": algorithmic,
        "note": "`achieved` counts executed vector lane-operations (SQ_THREAD_CYCLES_VALU), one flop each; the algorithm's own flops stand beside "
                "it under `algorithmic`.  The vector units are busy about 85 % of the launch (SQ_ACTIVE_INST_VALU against its duration) with "
                "48 of 64 lanes at work per instruction: what is left is fewer instructions and fuller waves, not more overlap",
    }


def measured_traffic(width, height):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC passes (4K only)."""
    if (width, height) != (3840, 2160):
        return None
    for name in ("round3_traffic.json", "round2_traffic.json", "round1_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return int(json.load(f)["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            continue
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


def radiosity_flops(map_size, bounces=4):
    """Algorithmic flops of Engine::compute_lightmaps as the reference runs it (every bounce marches
    every patch pair again, radiosity.rs:119-153): from the oracle's event counts at MAP_SIZE 8;
    other sizes are scaled by the number of patch pairs (M^4), which is exact for the per-pair
    arithmetic and an estimate for the march steps."""
    from raymarching_amd.flops import frame_flops

    try:
        with open(os.path.join(ROOT, "tests", "golden", "counters.json")) as f:
            c = json.load(f)["radiosity_map8_bounces4"]
    except (OSError, KeyError):
        return None
    pairs8 = 512 * 448
    per_pair = 40  # radiosity.rs:126-152 besides the march: difference, length, division, two dots, product, max, division, min, 3 multiply-adds
    at8 = frame_flops(c) + per_pair * pairs8 * 4
    return int(at8 * (map_size / 8.0) ** 4 * bounces / 4)


def radiosity_cpu_baseline(scene, budget_s):
    """The CPU oracle's compute_lightmaps (single thread, as the reference's) at MAP_SIZE 8, repeated
    until about `budget_s` seconds are spent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle

    e = oracle.OracleEngine(scene)
    times = []
    t_start = time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        e.compute_lightmaps(4)
        times.append(time.perf_counter() - t0)
    e.close()
    times.sort()
    median = times[len(times) // 2]
    marches = 512 * 448 * 4 + 512  # the reference marches every pair again in every bounce, plus direct lighting
    return {
        "value": marches / median / 1e6,
        "unit": "Mmarches/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{len(times)} runs of compute_lightmaps at MAP_SIZE 8 (512 patches, {marches} occlusion marches each, 4 bounces), "
        f"median {median:.2f} s, single thread as the reference (main.rs:73 runs it before the thread pool exists)",
    }


def make_group(eng, rank, world, local_rank, backend, in_flight):
    """The rank's rm_group over `eng`: RCCL inside the library where a communicator can be built on every
    rank, else (a rehearsal with every rank on one GPU, or RCCL missing) the library's custom-exchange
    form with an all-gather through host memory and gloo.  Returns (group, backend label)."""
    import torch
    import torch.distributed as dist

    group = None
    from raymarching_amd.group import Group, unique_id

    if backend == "nccl":
        # the 128-byte id is the only thing the ranks exchange outside the library
        failure = None
        try:
            ids = [unique_id() if rank == 0 else None]
        except Exception as exc:  # (no RCCL to be found: every rank must still reach the broadcast)
            ids, failure = [None], f"{type(exc).__name__}: {exc}"
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        if ids[0] is None:
            failure = failure or "rank 0 could not make a communicator id"
        else:
            try:
                if os.environ.get("RM_BENCH_FAIL_RCCL") == "1":  # rehearsal of the fallback below
                    raise RuntimeError("failure requested by RM_BENCH_FAIL_RCCL")
                group = Group(eng, rank, world, ids[0], frames_in_flight=in_flight)
            except Exception as exc:
                failure = f"{type(exc).__name__}: {exc}"
        # The transport is the ranks' common choice: if the communicator could not be built on any
        # rank, all of them hand the library the host all-gather below instead (same frames, a
        # slower exchange), and the line says so.  The rendering is the library's either way.
        failures = [None] * world
        if world > 1:
            dist.all_gather_object(failures, failure)
        else:
            failures = [failure]
        if any(failures):
            if group is not None:
                group.close()
                group = None
            backend = "gloo (RCCL communicator failed: " + next(f for f in failures if f)[:200] + ")"
            print(f"[bench] rank {rank}: {backend}", file=sys.stderr)
    if group is None:
        # rehearsal on one GPU: RCCL refuses two ranks on one device, so the all-gather goes
        # through host memory and gloo; everything else is the path above
        def host_all_gather(d_send, d_recv, nbytes, hip_stream):
            from raymarching_amd.dist import _alias_u8

            torch.cuda.synchronize()
            mine = _alias_u8(d_send, nbytes, local_rank).cpu()
            if world > 1:
                everyone = torch.empty(nbytes * world, dtype=torch.uint8)
                dist.all_gather_into_tensor(everyone, mine)
            else:
                everyone = mine
            _alias_u8(d_recv, nbytes * world, local_rank).copy_(everyone)
            torch.cuda.synchronize()
            return 0

        def host_gather(d_send, d_recv, nbytes, root, hip_stream):
            from raymarching_amd.dist import _alias_u8

            torch.cuda.synchronize()
            mine = _alias_u8(d_send, nbytes, local_rank).cpu()
            if world > 1:
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)] if rank == root else None
                dist.gather(mine, parts, dst=root)
                everyone = torch.cat(parts) if rank == root else None
            else:
                everyone = mine
            if rank == root:
                _alias_u8(d_recv, nbytes * world, local_rank).copy_(everyone)
            torch.cuda.synchronize()
            return 0

        group = Group(eng, rank, world, frames_in_flight=in_flight, all_gather=host_all_gather, gather=host_gather)
    return group, backend


def radiosity_workload(args, world, rank, local_rank, backend):
    """One step = Engine::compute_lightmaps (engine.rs:246-326): clouds, direct lighting, the
    form-factor matrix of all patch pairs, 4 bounces; receiver patches sharded over the ranks with
    one all-gather per bounce (inside the library, rm_group_compute_lightmaps)."""
    import torch
    import torch.distributed as dist

    from raymarching_amd import Engine, default_scene

    M = args.map_size
    scene = default_scene()
    eng = Engine(scene, 64, 64, map_size=M, device=local_rank)
    gathering = world > 1 or os.environ.get("RM_BENCH_GATHER_ALONE") == "1"
    group = None
    if gathering:
        group, backend = make_group(eng, rank, world, local_rank, backend, 1)
    steps, warmup = min(args.steps, 20), min(args.warmup, 3)
    n = 8 * M * M
    phase_ms = {"form_factors": 0.0, "bounces": 0.0}

    def step():
        if gathering:
            group.compute_lightmaps(4)
        else:
            eng.compute_lightmaps(4)

    def phases():
        # the stepwise calls (what a patch-sharded host makes), with events between the phases (null
        # stream): the kernels' own times for the roofline, taken after the timed region
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        eng.radiosity_begin()
        ev[0].record()
        eng.radiosity_form_factors(0, n)
        ev[1].record()
        for _ in range(4):
            eng.radiosity_bounce(0, n)
            eng.radiosity_commit()
        ev[2].record()
        eng.synchronize()
        torch.cuda.synchronize()
        phase_ms["form_factors"] += ev[0].elapsed_time(ev[1])
        phase_ms["bounces"] += ev[1].elapsed_time(ev[2])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if not gathering:
        for _ in range(steps):
            phases()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        pairs = n * (n - M * M)
        result = {
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
                       "patches": n, "patch_pairs": pairs, "sharding": "receiver patches, all-gather per bounce" if gathering else "none",
                       "note": "value counts each patch pair once (the form factors are computed once); the reference marches every pair again in every bounce"},
        }
        if not gathering:
            form_ms = phase_ms["form_factors"] / steps
            bounce_ms = phase_ms["bounces"] / steps / 4
            # per launch the form-factor kernel marches every patch pair once: the flops of ONE of the reference's bounces
            flops = radiosity_flops(M, bounces=1)
            hbm = {"bound": "hbm", "achieved": 4.0 * n * n / (bounce_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                   "traffic": None, "kernel": "rad_bounce_kernel", "kernel_ms": bounce_ms,
                   "algorithmic_bytes_per_launch": 4 * n * n,
                   "note": "one bounce reads the [patches][patches] f32 form-factor matrix once: 4 bytes per (receiver, source) term"}
            hbm["frac"] = hbm["achieved"] / hbm["peak"]
            valu = None
            if flops is not None:
                achieved = flops / (form_ms * 1e-3) / 1e12
                valu = {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                        "frac_of_no_fma_peak": achieved / PEAK_FP32_NO_FMA_TFLOPS, "traffic": None, "kernel": "rad_form_kernel",
                        "kernel_ms": form_ms, "algorithmic_flops_per_launch": flops,
                        "note": "algorithmic flops of one occlusion march and one geometric term per patch pair (one bounce of the reference, "
                        "radiosity.rs:119-153), counted by the oracle at MAP_SIZE 8" + ("" if M == 8 else " and scaled by the patch-pair count (an estimate)")
                        + "; pairs that face away from each other are counted although the kernel proves their march away"}
            dominant_is_form = form_ms >= 4 * bounce_ms
            result["roofline"] = valu if (dominant_is_form and valu) else hbm
            result["roofline_other_kernel"] = hbm if (dominant_is_form and valu) else valu
            result["phase_ms"] = {"form_factors": form_ms, "one_bounce": bounce_ms}
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = radiosity_cpu_baseline(scene, args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if group is not None:
        group.close()
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
    # and runs the several-GPU path - strips, all-gather, re-ordering, frames in flight - on the real
    # backend; the exchange then is a copy.
    gathering = world > 1 or os.environ.get("RM_BENCH_GATHER_ALONE") == "1"
    if world > 1:
        # torch.distributed only carries the 128-byte group id, the barriers around the timed region and
        # the maximum over the ranks' times: gloo does.  The frames and lightmaps travel inside the library.
        dist.init_process_group("gloo")

    from raymarching_amd import Engine, default_scene
    from raymarching_amd.dist import STRIP_ROWS
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
        eng.set_march_cap(1024)
    if not gathering:
        # main.rs:73: once, before the frame loop (not timed).  On the stress scene the 173 spheres and
        # planes carry the 11072 patches; boxes occlude and receive nothing.
        eng.compute_lightmaps(4)
    eng.input(0, 0, (0.0, 0.0, 0.0), 0)  # first-frame pose (main.rs:78-80,122)
    if args.ssaa:
        eng.set_ssaa(4)
    if args.refraction:
        eng.set_refraction(True)
    if args.fast:
        eng.set_precision("fast")

    # Several GPUs: rank r renders the 8-row strips r, r+N, r+2N, ... (even load however the cost is
    # spread over the image) and the library does the rest of a frame in one C call per rank
    # (rm_group_frame): strips -> 3 bytes per pixel -> all-gather on the frame's own stream (RCCL) ->
    # rows in order -> FXAA on the gathered frame.  A rank's share of a frame is bounded by its few
    # slowest tiles, not by its size, so three frames are kept in flight (measured on one GPU: an
    # eighth of the 3840x2160 frame takes 0.132 ms alone, 0.045 ms per frame with three in flight).
    in_flight = args.frames_in_flight or (3 if gathering else 1)
    if args.no_overlap:
        in_flight = 1
    n_buf = max(in_flight, 2)
    group = None
    if gathering:
        group, backend = make_group(eng, rank, world, local_rank, backend, in_flight)
        group.compute_lightmaps(4)  # the same precompute, patch-sharded (a rehearsal of that exchange too)
        info = group.info()
        strips_per_rank, wire = info["strips_per_rank"], info["bytes_per_pixel_exchanged"]
        my_rows = sum(min(STRIP_ROWS, H - (rank + k * world) * STRIP_ROWS) for k in range(strips_per_rank) if (rank + k * world) * STRIP_ROWS < H)
        r0, r1 = 0, my_rows  # only the row count is used below
    else:
        r0, r1 = 0, H
        wire = None
    final = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(n_buf)]
    fx_srcs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(n_buf)] if args.fxaa and not gathering else None

    kernel_events = []
    # one GPU: frame s is rendered (and filtered) on streams[s % in_flight]
    streams = [stream] + [torch.cuda.Stream() for _ in range(in_flight - 1)]
    if in_flight > 1:
        # launches overlap: the machine is busy while a frame's slow tiles finish, and rendering them
        # as quarter waves (rm_engine_set_tile_splitting) would only cost throughput
        eng.set_tile_splitting(False)

    def render_whole(b, timed):
        st = torch.cuda.current_stream()
        eng.set_stream(st.cuda_stream)
        if timed:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(st)
        dst = fx_srcs[b] if args.fxaa else final[b]
        eng.render_rows_device(dst.data_ptr(), 0, H)
        if timed:
            ev1.record(st)
            kernel_events.append((ev0, ev1))
        if args.fxaa:
            eng.fxaa_device(fx_srcs[b].data_ptr(), final[b].data_ptr(), 0, H)

    def run(steps, timed):
        if gathering:
            group.set_timing(timed)
            for s in range(steps):
                group.frame(final[s % n_buf].data_ptr(), args.fxaa)
                if args.no_overlap:
                    group.wait()
            group.wait()
            return
        if in_flight == 1:
            for s in range(steps):
                render_whole(s & 1, timed)
            return
        for st in streams:
            st.wait_stream(stream)
        for s in range(steps):
            with torch.cuda.stream(streams[s % in_flight]):
                render_whole(s % in_flight, timed)
        for st in streams[1:]:
            stream.wait_stream(st)
        eng.set_stream(stream.cuda_stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(steps):
        """`steps` frames between two fences; the maximum over the ranks' wall times, in seconds."""
        fence()
        t0 = time.perf_counter()
        run(steps, True)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    if gathering and args.exchange == "presenter":
        group.set_exchange("presenter")
    # Pre-roll (untimed set-up, like the precompute): a GPU that has idled starts below the clock it
    # settles at under load; a timed region of a few milliseconds would sit on that ramp.  Every rank
    # runs the same number of frames (the group's frames are collective).
    preroll_frames = 0
    if args.preroll_seconds > 0:
        t_pre = time.perf_counter()
        run(8, False)
        torch.cuda.synchronize()
        per8 = max(time.perf_counter() - t_pre, 1e-4)
        rounds = int(min(max(args.preroll_seconds / per8, 1.0), 400.0))
        if world > 1:
            t = torch.tensor([rounds], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            rounds = int(t.item())
        for _ in range(rounds):
            run(8, False)
        torch.cuda.synchronize()
        preroll_frames = 8 * (rounds + 1)
    run(args.warmup, False)
    elapsed = timed_region(args.steps)

    exchange_ms = None
    if gathering:
        kernel_ms, _ = group.render_time()  # events around the render launch on each frame's own stream (spans overlap with frames in flight)
        exchange_ms_overlapped, _ = group.exchange_time()
        per_frame_kernel = []
    else:
        per_frame_kernel = [a.elapsed_time(b) for a, b in kernel_events]
        kernel_ms = sum(per_frame_kernel) / max(len(per_frame_kernel), 1)
    stats = eng.stats()
    last_buf = (args.steps - 1) % n_buf if gathering or in_flight > 1 else (args.steps - 1) & 1  # where the last frame landed
    # (the presenter rotates with every frame the group has issued in this mode, pre-roll and warm-up included)
    last_presenter = (group.next_presenter() - 1) % world if gathering and args.exchange == "presenter" else None

    # ---- secondary measurements, after the timed region (none of them is `value`) ----
    extras = {}
    whole = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    whole_fx = torch.zeros_like(whole) if args.fxaa else None

    def render_local():
        eng.render_rows_device(whole.data_ptr(), 0, H)
        if args.fxaa:
            eng.fxaa_device(whole.data_ptr(), whole_fx.data_ptr(), 0, H)

    checks = {}
    if gathering:
        # the gathered frame against the same frame rendered (and filtered) whole on this GPU, bit for bit
        render_local()
        torch.cuda.synchronize()
        local = whole_fx if args.fxaa else whole
        if last_presenter is None or last_presenter == rank:
            checks[args.exchange] = bool(torch.equal(final[last_buf], local))
    if gathering and not args.no_extras:
        # (none of this is `value`: a failure here - a build of RCCL without a gather, say - must not cost the line)
        try:
            calib = 16
            # (a) this rank's share rendered alone and the collective alone: one frame at a time, nothing else on the GPU
            group.set_exchange("all_gather")
            group.set_timing(True)
            for s in range(calib):
                group.frame(final[s % n_buf].data_ptr(), args.fxaa)
                group.wait()
            share_ms, _ = group.render_time()
            ag_ms, _ = group.exchange_time()
            group.set_exchange("presenter")
            group.set_timing(True)
            for s in range(calib):
                group.frame(final[s % n_buf].data_ptr(), args.fxaa)
                group.wait()
            pr_ms, _ = group.exchange_time()
            group.set_timing(False)
            kernel_alone_ms = share_ms
            # (b) the whole frame on this rank alone, every rank at the same time: what one GPU does without the group
            fence()
            t0 = time.perf_counter()
            for _ in range(calib):
                render_local()
            torch.cuda.synchronize()
            n1_ms = (time.perf_counter() - t0) / calib * 1e3
            # (c) the other exchange, timed like the headline
            other = "presenter" if args.exchange == "all_gather" else "all_gather"
            group.set_exchange(other)
            run(args.warmup, False)
            other_elapsed = timed_region(args.steps)
            other_last = (args.steps - 1) % n_buf
            if other == "all_gather" or (group.next_presenter() - 1) % world == rank:
                checks[other] = bool(torch.equal(final[other_last], local))
            if world > 1:
                t = torch.tensor([share_ms, ag_ms, pr_ms, n1_ms], dtype=torch.float64)
                tmax, tmean = t.clone(), t.clone()
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dist.all_reduce(tmean, op=dist.ReduceOp.SUM)
                tmean /= world
                gathered_checks = [None] * world
                dist.all_gather_object(gathered_checks, checks)
                checks = {k: all(c[k] for c in gathered_checks if k in c) for k in ("all_gather", "presenter") if any(k in c for c in gathered_checks)}
            else:
                tmax = tmean = torch.tensor([share_ms, ag_ms, pr_ms, n1_ms], dtype=torch.float64)
            ms_per_step = elapsed / args.steps * 1e3
            other_ms = other_elapsed / args.steps * 1e3
            extras = {
                "render_share_ms": {"mean": float(tmean[0]), "max_over_ranks": float(tmax[0]),
                                    "note": "this rank's strips rendered alone (one frame at a time, events around the launch)"},
                "exchange_ms": {"all_gather": {"mean": float(tmean[1]), "max_over_ranks": float(tmax[1])},
                                "presenter_gather": {"mean": float(tmean[2]), "max_over_ranks": float(tmax[2])},
                                "note": "the collective alone, one frame at a time (events around it on the frame's stream)"},
                "n1_ms_per_frame": {"mean": float(tmean[3]), "max_over_ranks": float(tmax[3]),
                                    "note": "the whole frame on one rank's GPU without the group, every rank at the same time"},
                "efficiency_vs_n1": float(tmean[3]) / (world * ms_per_step),
                "other_exchange": {"exchange": other, "ms_per_step": other_ms, "value": W * H * (4 if args.ssaa else 1) / (other_ms * 1e-3) / 1e6,
                                   "unit": "Mrays/s", "efficiency_vs_n1": float(tmean[3]) / (world * other_ms), "steps": args.steps,
                                   "note": "the same frames through the other exchange, timed like the headline"},
                "communicators": group.communicators(),
            }
            exchange_ms = extras["exchange_ms"]
        except Exception as exc:  # noqa: BLE001
            extras = {"extras_failed": f"{type(exc).__name__}: {exc}"[:300]}
            print(f"[bench] rank {rank}: secondary measurements failed: {extras['extras_failed']}", file=sys.stderr)
    elif not gathering and not args.no_extras and args.workload == "render" and not args.ssaa and not args.refraction:
        # the shader clock under this load: the probe wave runs beside frames that are enqueued first
        run(60 if W * H <= 1920 * 1080 else 30, False)
        clock_load = eng.shader_clock_mhz(200)
        torch.cuda.synchronize()
        eng.set_stream(stream.cuda_stream)
        fast_frame = None
        fast_tier = None
        if not args.fast and not args.fxaa and in_flight == 1:
            # the same frames in the fast tier, timed like the headline (never `value`)
            eng.set_precision("fast")
            run(max(args.warmup, 40), False)  # (the GPU is still warm from the frames above; the fast kernel's code is loaded here)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps, False)
            torch.cuda.synchronize()
            fast_ms = (time.perf_counter() - t0) / args.steps * 1e3
            fast_frame = final[(args.steps - 1) & 1].cpu().numpy()
            eng.set_precision("strict")
            render_whole(last_buf, False)  # the strict frame back where the parity check below reads it
            torch.cuda.synchronize()
            fast_tier = {"ms_per_step": fast_ms, "value": W * H / (fast_ms * 1e-3) / 1e6, "unit": "Mrays/s", "steps": args.steps,
                         "note": "rm_engine_set_precision(RM_PRECISION_FAST): sight-ray chain strict, colour arithmetic and soft-shadow "
                                 "rays relaxed (FMA, one-ulp root/reciprocal/exp2/log2); opt-in, the headline is the strict tier"}
        # the reference-shaped call: Engine::render into host memory (main.rs:141-145), PCIe included
        from raymarching_amd import _lib as rm_lib
        import ctypes

        lib = rm_lib.load()
        n_abi = 8
        pinned = lib.rm_host_alloc(W * H * 4)
        abi = {}
        if pinned:
            ptr = ctypes.cast(ctypes.c_void_p(pinned), ctypes.POINTER(ctypes.c_uint8))
            zero = rm_lib.f3((0.0, 0.0, 0.0))
            for _ in range(2):
                rm_lib.check(lib.rm_engine_render(eng.handle, ptr, W * 4, 0, 0, zero, 0), "rm_engine_render")
            t0 = time.perf_counter()
            for _ in range(n_abi):
                rm_lib.check(lib.rm_engine_render(eng.handle, ptr, W * 4, 0, 0, zero, 0), "rm_engine_render")
            abi["page_locked_ms"] = (time.perf_counter() - t0) / n_abi * 1e3
            lib.rm_host_free(pinned)
        pageable = np.empty((H, W, 4), dtype=np.uint8)
        for _ in range(2):
            eng.render(pageable)
        t0 = time.perf_counter()
        for _ in range(n_abi):
            eng.render(pageable)
        abi["pageable_ms"] = (time.perf_counter() - t0) / n_abi * 1e3
        abi["frames"] = n_abi
        abi["note"] = "rm_engine_render: input block, frame, copy into the caller's host buffer (PCIe-inclusive; never `value`)"
        for k in ("page_locked_ms", "pageable_ms"):
            if k in abi:
                abi[k.replace("_ms", "_mrays_per_s")] = W * H / (abi[k] * 1e-3) / 1e6
        extras = {"shader_clock_mhz": {"under_load": clock_load, "nominal": 2400.0,
                                       "note": "one wave counting shader cycles over 200 us of the constant-rate counter while frames render"},
                  "abi_render": abi}
        if fast_tier:
            extras["fast_tier"] = fast_tier

    result = None
    if rank == 0:
        rays = W * H * (4 if args.ssaa else 1)
        ms_per_step = elapsed / args.steps * 1e3
        counters = load_counters(W, H) if args.workload == "render" and not args.ssaa and not args.refraction else None
        roofline = None
        roof_kernel_ms = kernel_ms
        if gathering:
            # with frames in flight the events around one launch span other launches too: the roofline is
            # that of this rank's share rendered alone, or nothing
            roof_kernel_ms = extras["render_share_ms"]["mean"] if "render_share_ms" in extras else None
        if counters is not None and roof_kernel_ms:
            band_share = (r1 - r0) / H  # flops are counted for the whole frame; a band is its share by rows
            flops = frame_flops(counters)
            achieved = flops * band_share / (roof_kernel_ms * 1e-3) / 1e12
            tiles = ((W + 7) // 8) * ((r1 - r0 + 7) // 8)
            roofline = {
                "bound": "valu",
                "achieved": achieved,
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS,
                "frac_of_no_fma_peak": achieved / PEAK_FP32_NO_FMA_TFLOPS,
                "traffic": int(measured_traffic(W, H) * band_share) if measured_traffic(W, H) else None,
                "traffic_source": "rocprofv3 PMC passes committed under profiles/ (not collected live)",
                # (a launch of 65536 tiles or more of a small scene runs the build of the kernel for 8 waves per SIMD)
                "kernel": "render_dense_kernel" if tiles >= 65536 and not args.ssaa and not args.refraction else "render_kernel",
                "kernel_ms": roof_kernel_ms,
                # a launch of fewer than 65536 tiles with one frame in flight is planned (slow tiles first and in
                # quarters): the events then also span the plan's 5 us kernel
                "kernel_ms_spans_the_launch_plan": bool(tiles < 65536 and (in_flight == 1 or gathering) and not args.ssaa and not args.refraction),
                "algorithmic_flops_per_launch": int(flops * band_share),
                "note": "FP32 vector ALU bound (no MFMA, HBM traffic is 4 B per ray); strict IEEE parity forbids FMA contraction; the flops are "
                "those of the reference's algorithm as the oracle counts them: work the kernel proves away (shadow rays whose way to the light "
                "is provably open, spheres out of a step's reach, the first step of a primary march) counts as done",
            }
            if per_frame_kernel:
                srt = sorted(per_frame_kernel)
                roofline["kernel_ms_min"] = srt[0]
                roofline["kernel_ms_median"] = srt[len(srt) // 2]
                roofline["kernel_ms_max"] = srt[-1]
                k = max(1, min(5, len(per_frame_kernel) // 2))
                roofline["kernel_ms_first_frames"] = sum(per_frame_kernel[:k]) / k
                roofline["kernel_ms_last_frames"] = sum(per_frame_kernel[-k:]) / k
                roofline["first_frames_slower_than_last"] = bool(roofline["kernel_ms_first_frames"] > 1.03 * roofline["kernel_ms_last_frames"])
            if gathering:
                roofline["kernel_ms_source"] = "calibration after the timed region: one frame at a time, nothing else on the GPU"
            clock = extras.get("shader_clock_mhz", {}).get("under_load") if extras else None
            if clock:
                roofline["shader_clock_mhz"] = clock
                roofline["frac_of_peak_at_measured_clock"] = achieved / (PEAK_FP32_TFLOPS * clock / 2400.0)
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
            "data": ("synthetic: the reference's default scene (main.rs:154-193), first-frame pose, lightmaps from 4 radiosity bounces"
                     if args.workload == "render" else
                     "synthetic: seed-fixed stress scene (raymarching_amd.scene.stress_scene: 167 spheres and 83 boxes in the reference's room, "
                     "15 of the spheres reflective), first-frame pose, lightmaps from 4 radiosity bounces over its 11072 patches"),
            "config": {
                "workload": (f"default scene {W}x{H}, march + shading + texture + lightmap + reflections" if args.workload == "render"
                             else f"stress scene: 250 spheres and boxes + room, smooth union k=0.08, {W}x{H}, march cap 1024")
                + (" + FXAA" if args.fxaa else "") + (", 4 rays per pixel" if args.ssaa else "") + (", refraction on" if args.refraction else ""),
                "width": W,
                "height": H,
                "bands": world,
                "sharding": "interleaved 8-row strips" if gathering else "none",
                "exchange": (("one all-gather per frame inside the library (rm_group_frame), " if args.exchange == "all_gather" else
                              "strips gathered to a rotating presenter (frame f lands on rank f mod N) inside the library (rm_group_frame), ")
                             + ("RCCL" if backend == "nccl" else ("gloo through host memory (rehearsal)" if backend == "gloo" else backend))) if gathering else None,
                "fxaa_after_gather": bool(args.fxaa and gathering),
                "fxaa": bool(args.fxaa),
                "ssaa": 4 if args.ssaa else 1,
                "refraction": bool(args.refraction),
                "map_size": 8,
                "overlap_gather": gathering and not args.no_overlap,
                "bytes_per_pixel_exchanged": (wire if args.exchange == "all_gather" else 4) if gathering else None,
                "frames_in_flight": in_flight,
                "preroll_frames": preroll_frames,
                "precision": "fast" if args.fast else "strict",
            },
            "roofline": roofline if args.workload == "render" else stress_roofline(W, H, kernel_ms),
            "cap_hits": stats,
        }
        if gathering:
            result["gathered_frame_equals_single_gpu_render"] = bool(checks) and all(checks.values())
            result["frame_checks"] = checks
            result["kernel_ms_overlapped"] = kernel_ms
            result["exchange_ms_overlapped"] = exchange_ms_overlapped
        for k, v in extras.items():
            result[k] = v
        if not args.no_cpu_baseline and not gathering:  # the CPU leg runs at N=1 only
            lightmaps = [eng.get_lightmap(i) for i in range(eng.n_objects)]
            base, cpu_rows = cpu_baseline(scene, lightmaps, W, H, args.cpu_seconds, args.ssaa, args.refraction)
            result["cpu_baseline"] = base
            if not args.fxaa:
                # the same frame from the GPU: a parity check at the benchmarked size
                gpu_rows = final[last_buf].cpu().numpy()
                diff = np.abs(gpu_rows.astype(np.int16) - cpu_rows.astype(np.int16))
                result["parity_vs_cpu_frame"] = {
                    "max_abs_diff": int(diff.max()),
                    "exact_pixel_fraction": float((diff.reshape(-1, 4).max(axis=1) == 0).mean()),
                    "tier": "fast" if args.fast else "strict",
                }
                if extras.get("fast_tier") and fast_frame is not None:
                    fdiff = np.abs(fast_frame.astype(np.int16) - cpu_rows.astype(np.int16))
                    result["fast_tier"]["parity_vs_cpu_frame"] = {
                        "max_abs_diff": int(fdiff.max()),
                        "exact_pixel_fraction": float((fdiff.reshape(-1, 4).max(axis=1) == 0).mean()),
                        "contract": "max |delta| <= 2/255 per channel, >= 99.9 % of pixels identical",
                    }
        print(json.dumps(result), flush=True)

    if group is not None:
        group.close()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
