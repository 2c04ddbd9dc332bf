"""Regenerates the fixtures in this directory from the CPU oracle.

The reference has no tests or golden data and cannot be built here, so these
vectors are outputs of oracle/rm_oracle.cpp (g++ -O2 -ffp-contract=off, glibc
libm): they pin the oracle against accidental change and record the workload's
event counts.  Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from raymarching_amd.scene import default_scene  # noqa: E402

POSES = {  # name -> (width, height, mouse_x, mouse_y, rel_move, scroll)
    "pose0_800x600": (800, 600, 0, 0, (0.0, 0.0, 0.0), 0),
    "pose1_400x300": (400, 300, 150, -60, (0.0, 0.0, 0.0), 0),
    "pose2_400x300": (400, 300, -420, 90, (0.05, 0.0, 0.15), 2),
}


def save_bgra_png(path, frame):
    assert (frame[:, :, 3] == 0).all()
    Image.fromarray(np.ascontiguousarray(frame[:, :, 2::-1])).save(path, optimize=True)


def main():
    scene = default_scene()
    e = oracle.OracleEngine(scene, counting=True)
    lm, em = e.compute_lightmaps(4, stages=True)
    np.savez_compressed(os.path.join(HERE, "lightmaps_default.npz"), lightmaps=lm, emissive=em)
    counters = {"radiosity_map8_bounces4": e.counters()}
    e.close()

    for name, (w, h, mx, my, mv, sc) in POSES.items():
        e = oracle.OracleEngine(scene)
        for i in range(e.n_objects):
            e.set_lightmap(i, lm[-1][i])
        frame = e.render(w, h, mx, my, mv, sc)
        save_bgra_png(os.path.join(HERE, f"frame_{name}.png"), frame)
        if name == "pose1_400x300":
            save_bgra_png(os.path.join(HERE, "fxaa_pose1_400x300.png"), oracle.fxaa(frame))
        e.close()

    e = oracle.OracleEngine(scene)
    hit, steps, pos = e.trace_primary(800, 600)
    import hashlib

    # end positions are incompressible: keep their digest only
    digest = hashlib.sha256(np.ascontiguousarray(pos).tobytes()).hexdigest()
    np.savez_compressed(os.path.join(HERE, "trace_pose0_800x600.npz"), hit=hit.astype(np.int8), steps=steps.astype(np.uint16), pos_sha256=np.array(digest))
    e.close()

    for w, h in [(800, 600), (1920, 1080), (3840, 2160)]:
        e = oracle.OracleEngine(scene, counting=True)
        for i in range(e.n_objects):
            e.set_lightmap(i, lm[-1][i])
        e.reset_counters()
        e.render(w, h)
        counters[f"frame_pose0_{w}x{h}"] = e.counters()
        e.close()
        print(w, h, "done", flush=True)
    with open(os.path.join(HERE, "counters.json"), "w") as f:
        json.dump(counters, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
