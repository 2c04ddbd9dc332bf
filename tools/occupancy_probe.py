"""Frame time at 4K with the render kernel's residency capped by a dummy LDS allocation per workgroup
(RM_EXP_LDS bytes, read once per process by a build made with `make EXTRA=-DRM_EXP_LDS OUT=...`: one
child process per setting).  usage: occupancy_probe.py lib.so bytes [bytes ...]"""
import os
import subprocess
import sys

lib = sys.argv[1]
for n in sys.argv[2:]:
    env = dict(os.environ, RM_EXP_LDS=n, RM_ROUNDS="7", RM_FRAMES="20")
    out = subprocess.run([sys.executable, "tools/quick_bench.py", lib], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if "median" in l]
    waves = (160 * 1024 // int(n)) if int(n) else 0
    print(f"LDS per wave {n:>6} B (at most {waves} waves per CU): {line[0] if line else out.stderr[-300:]}", flush=True)
    if out.returncode != 0:
        sys.exit(out.returncode)
