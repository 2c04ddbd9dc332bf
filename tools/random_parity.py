"""Random-pose and random-scene parity sweep against the oracle (run on the GPU box):
python tools/random_parity.py [poses] [scenes]   (RM_REFRACTION=1: with the refraction switch on)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from conftest import frame_stats, make_oracle
from raymarching_amd import Engine
from raymarching_amd.scene import (SOFT_GRAY, SOFT_GREEN, SOFT_RED, SOFT_YELLOW, WHITE, Material, PointLight, Scene, Sphere,
                                   XPlane, YPlane, ZPlane, default_scene)

W, H = 320, 240

def run(n_poses=40, n_scenes=12, seed=2024, verbose=True):
    """Returns (frames compared, worst max|d|, worst inexact pixel fraction)."""
    rng = np.random.default_rng(seed)
    state = {"worst": 0, "inexact": 0.0, "frames": 0}


    def check(scene, poses, bounces, tag):
        eng = Engine(scene, W, H)
        ora = make_oracle(scene)
        if os.environ.get("RM_REFRACTION") == "1":  # the same sweep with the refraction switch on
            eng.set_refraction(True)
            ora.set_refraction(True)
        eng.compute_lightmaps(bounces)
        ora.compute_lightmaps(bounces)
        for i in range(eng.n_objects):
            assert np.array_equal(eng.get_lightmap(i).view(np.uint32), ora.get_lightmap(i).view(np.uint32)), (tag, "lightmap", i)
        for pose in poses:
            got, ref = eng.render(None, *pose), ora.render(W, H, *pose)
            mx, exact = frame_stats(got, ref)
            state["worst"], state["inexact"], state["frames"] = max(state["worst"], mx), max(state["inexact"], 1.0 - exact), state["frames"] + 1
            if mx > 2 or exact < 0.999:
                print(f"MISMATCH {tag} pose {pose}: max|d|={mx} exact={exact:.5%}")
        st = eng.stats()
        eng.close()
        return st


    def random_poses(n):
        out = []
        for _ in range(n):
            move = tuple(float(v) for v in rng.uniform(-0.6, 0.6, 3)) if rng.random() < 0.5 else (0.0, 0.0, 0.0)
            out.append((int(rng.integers(-900, 900)), int(rng.integers(-400, 400)), move, int(rng.integers(-6, 10))))
        return out


    st = check(default_scene(), random_poses(n_poses), 4, "default")
    if verbose:
        print("default scene:", st)
    tex = default_scene().objects[2].texture
    for s in range(n_scenes):
        n_sph = int(rng.integers(1, 5))
        objs, placed = [], []
        while len(objs) < n_sph:
            c = tuple(float(v) for v in rng.uniform([-2.0, -1.2, -0.8], [2.0, 2.5, 1.5]))
            r = float(rng.uniform(0.3, 0.9))
            if any(np.linalg.norm(np.subtract(c, pc)) < r + pr + 0.05 for pc, pr in placed):
                continue
            placed.append((c, r))
            mat = Material(float(rng.uniform(0.05, 0.3)), float(rng.uniform(0.1, 1.0)), float(rng.choice([0.0, 0.5, 0.9])),
                           float(rng.choice([4.0, 16.0, 32.0])), float(rng.choice([0.0, 0.0, 0.25, 1.0])), 0.0)
            objs.append(Sphere(c, r, mat, [WHITE, SOFT_YELLOW, SOFT_RED, SOFT_GREEN][len(objs) % 4]))
        basic = Material.basic()
        objs += [YPlane(-2.0, 1.0, basic, SOFT_GRAY, tex), YPlane(4.0, -1.0, basic, SOFT_GRAY, tex), XPlane(-3.0, 1.0, basic, SOFT_RED),
                 XPlane(3.0, -1.0, basic, SOFT_GREEN), ZPlane(2.0, -1.0, basic, SOFT_GRAY), ZPlane(-4.0, 1.0, basic, SOFT_GRAY)]
        if rng.random() < 0.3:
            objs.append(XPlane(float(rng.uniform(-1.0, 1.0)), 1.0 if rng.random() < 0.5 else -1.0, basic, SOFT_YELLOW))  # a partition wall
        light = PointLight(tuple(float(v) for v in rng.uniform([-2.0, 0.0, -2.0], [2.0, 3.0, 1.5])), float(rng.uniform(2.0, 5.0)))
        scene = Scene(objects=objs, camera_position=(0.0, 0.5, -3.5), light=light)
        st = check(scene, random_poses(4), 2, f"scene {s} ({n_sph} spheres, {len(objs)} objects)")
        if verbose:
            print(f"scene {s}: {n_sph} spheres, {len(objs)} objects, caps {st}")
    return state["frames"], state["worst"], state["inexact"]


if __name__ == "__main__":
    frames, worst, worst_inexact = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 12)
    print(f"{frames} frames: worst max|d| = {worst}, worst inexact fraction = {worst_inexact:.2e}")
    sys.exit(0 if worst <= 2 and worst_inexact <= 1e-3 else 1)
