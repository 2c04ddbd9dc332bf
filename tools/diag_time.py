"""Wave start/end times of a -DRM_DIAG_TIME build: python tools/diag_time.py tools/librm_diagtime.so"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from raymarching_amd import _lib, default_scene

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from raymarching_amd.engine import Engine

W, H = int(os.environ.get("RM_W", 3840)), int(os.environ.get("RM_H", 2160))
if os.environ.get("RM_STRESS"):  # the smooth-union stress scene (bench.py --workload stress) instead
    from raymarching_amd.scene import stress_scene

    e = Engine(stress_scene(250, boxes=True, smooth_union=0.08), W, H)
    e.set_march_cap(1024)
else:
    e = Engine(default_scene(), W, H)
e.compute_lightmaps(4)
only = os.environ.get("RM_ONLY_STRIP")  # render just this 8-row strip: every wave alone on its SIMD
if only is not None:
    r0 = int(only) * 8
    buf = torch.zeros((8, W, 4), dtype=torch.uint8, device="cuda")
    e.input(0, 0, (0, 0, 0), 0)
    for _ in range(2):
        e.render_rows_device(buf.data_ptr(), r0, r0 + 8)
    e.synchronize()
    img = buf.cpu().numpy()
    H = 8
else:
    # (one launch for the whole frame, as the benchmark renders it: rm_engine_render would send a
    # large frame in row bands)
    buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    e.input(0, 0, (0, 0, 0), 0)
    for _ in range(int(os.environ.get("RM_FRAMES", 2))):  # (a launch may be planned from the last one's timings)
        e.render_rows_device(buf.data_ptr(), 0, H)
    e.synchronize()
    print(f"tiles listed by the last launch's plan: {e.split_tiles()}")
    img = buf.cpu().numpy()
px = img.view(np.uint32).reshape(H, W)
t0 = px[0::8, 0::8].astype(np.int64)
t1 = px[0::8, 1::8].astype(np.int64)
hw = px[0::8, 2::8]
dur = ((t1 - t0) & 0xFFFFFFFF) / 100.0  # us
start = ((t0 - t0.min()) & 0xFFFFFFFF) / 100.0
end = start + dur
print(f"waves {dur.size}: mean {dur.mean():.2f} us, median {np.median(dur):.2f}, p90 {np.percentile(dur, 90):.2f}, p99 {np.percentile(dur, 99):.2f}, max {dur.max():.2f}")
print(f"grid span {end.max():.1f} us; sum of wave lifetimes {dur.sum() / 1e3:.1f} ms -> mean resident waves {dur.sum() / end.max():.0f}")
for q in (50, 75, 90, 95, 99, 100):
    print(f"  {q:3d} % of waves had ended by {np.percentile(end, q):.1f} us")
# resident waves over the span, in 20 bins, and how evenly the compute units were used
edges = np.linspace(0.0, end.max(), 21)
res = [int(((start.ravel() < b) & (end.ravel() > a)).sum() * 1.0) for a, b in zip(edges[:-1], edges[1:])]
print("  waves alive in each 1/20 of the span (a wave counts in every bin it touches): " + " ".join(str(r) for r in res))
# how long a wave slot (die, shader engine, array, compute unit, SIMD, wave id) stands empty between
# the end of one wave and the first instruction of the next
slot = (hw.ravel().astype(np.int64) & 0xF000FF3F)  # die | SE, SH, CU | SIMD | wave id
order_s = np.lexsort((start.ravel(), slot))
s_sorted, st_sorted, en_sorted = slot[order_s], start.ravel()[order_s], end.ravel()[order_s]
same = s_sorted[1:] == s_sorted[:-1]
gaps = (st_sorted[1:] - en_sorted[:-1])[same]
print(f"  {len(np.unique(slot))} wave slots seen; gap between consecutive waves of a slot: median {np.median(gaps):.2f} us, mean {gaps.mean():.2f}, "
      f"p10 {np.percentile(gaps, 10):.2f}, p90 {np.percentile(gaps, 90):.2f}, p99 {np.percentile(gaps, 99):.2f}; negative {int((gaps < 0).sum())}")
cu_key = (hw.ravel() >> 8) & 0xFFFF  # HW_ID above the wave and SIMD fields: CU, SH, SE
busy = {}
for k, d in zip(cu_key.tolist(), dur.ravel().tolist()):
    busy[k] = busy.get(k, 0.0) + d
b = np.array(sorted(busy.values()))
print(f"  {len(b)} distinct HW_ID compute-unit keys; summed wave lifetime per key: min {b.min():.0f} us, median {np.median(b):.0f}, max {b.max():.0f}")
if not os.environ.get("RM_DIAG_BOTH"):
    cyc = px[0::8, 3::8].astype(np.int64)
    long_enough = dur > 5.0
    mhz = cyc[long_enough] / dur[long_enough]
    print(f"  shader clock over waves longer than 5 us: median {np.median(mhz):.0f} MHz, p5 {np.percentile(mhz, 5):.0f}, p95 {np.percentile(mhz, 95):.0f}")
order = np.argsort(end.ravel())[-8:]
for i in order:
    ty, tx = divmod(int(i), W // 8)
    print(f"  late wave: tile ({tx},{ty}) start {start.ravel()[i]:.1f} us, lifetime {dur.ravel()[i]:.1f} us")
if os.environ.get("RM_DIAG_BOTH"):
    it = px[0::8, 3::8].astype(np.int64).ravel()
    busiest = px[0::8, 4::8].astype(np.int64).ravel()
    lanes = px[0::8, 5::8].astype(np.int64).ravel()
    marches = px[0::8, 6::8].astype(np.int64).ravel()
    for i in np.argsort(dur.ravel())[-8:]:
        ty, tx = divmod(int(i), W // 8)
        print(f"  long wave: tile ({tx},{ty}) {dur.ravel()[i]:.1f} us: {it[i]} loop iterations, busiest lane {busiest[i]}, "
              f"mean lane {lanes[i] / 64:.0f}, {marches[i]} marches")
    names = ["setup", "march loops", "closest", "shade", "shadow", "reflect bookkeeping", "pack"]
    for i in np.argsort(dur.ravel())[-3:]:
        ty, tx = divmod(int(i), W // 8)
        laps = [int(px[ty * 8 + 1, tx * 8 + k]) for k in range(7)]  # lanes 8..14 are row 1 of the tile
        kinds = [int(px[ty * 8 + 2, tx * 8 + k]) for k in range(8)]  # lanes 16..23 are row 2 of the tile
        print(f"  tile ({tx},{ty}): cycles " + ", ".join(f"{n} {c}" for n, c in zip(names, laps)) + f"; total {sum(laps)}")
        print(f"     march iterations {kinds[1]}, shadow iterations {kinds[2]}, sphere evaluations {kinds[3]}, penumbra divisions {kinds[4]}, marches {kinds[5]}")
    heavy = dur.ravel() > np.percentile(dur, 99)
    print(f"  slowest 1 % of waves: iterations {it[heavy].mean():.0f}, busiest lane {busiest[heavy].mean():.0f}, mean lane {lanes[heavy].mean() / 64:.0f}")
np.save("gpurun_out/diag_time_dur.npy", dur.astype(np.float32))
np.save("gpurun_out/diag_time_start.npy", start.astype(np.float32))
np.save("gpurun_out/diag_time_hw.npy", hw.astype(np.uint32))
