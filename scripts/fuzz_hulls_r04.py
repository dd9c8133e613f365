"""Round 4 campaign for the hull-against-map collider (np_convex_static_tile: a conservative filter in front of the exact test, its slack
from an error analysis): random scenes of tilted, spinning teapot / cube hulls dropped on a static floor box or the ground plane strewn
with turned blocks, the whole scene shifted up to kilometres from the origin (where f32 rounds positions to 0.1-0.5 mm and the filter's
slack has to grow with them), both precisions, hull-hull collisions off -- state after 120-240 ticks bit-identical to the oracle.
usage: python scripts/fuzz_hulls_r04.py [n_seeds=200] [first_seed=0]"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
from oracle.orc_ctypes import Oracle  # noqa: E402

H = 1.0 / 60.0


def cube_hull(half):
    pts = np.array(list(itertools.product((-half, half), repeat=3)), float)
    planes = np.array([[(sg if a == k else 0.0) for k in range(3)] + [half] for a in range(3) for sg in (-1.0, 1.0)])
    return pts, planes


def one(seed, teapot):
    rng = np.random.default_rng(1000 + seed)
    dtype = "float32" if seed % 2 else "float64"
    plane = bool(rng.integers(0, 2))
    nx, nz = int(rng.integers(2, 10)), int(rng.integers(2, 9))
    tilt = float(rng.uniform(0.0, 1.2))
    off = np.array([rng.choice([0.0, 37.5, 911.0, 4096.0, -2500.0]), 0.0, rng.choice([0.0, -63.0, 1500.0, 7000.0])])
    if rng.integers(0, 3) == 0:
        scene = pkg.scenes.hull_grid(teapot, nx, nz, seed=seed + 1, y_range=(0.7, 1.8), spin=True, tilt=tilt, floor_box=not plane, plane=plane)
        nh = scene.n
    else:
        scene = pkg.scenes.hull_grid(teapot, nx, nz, seed=seed + 1, y_range=(0.7, 1.8), spin=True, tilt=tilt, floor_box=not plane, plane=plane)
        if rng.integers(0, 2) == 0:           # cube hulls instead of teapots: 8 points, every one of them a likely contact
            half = float(rng.uniform(0.2, 0.6))
            pts, planes = cube_hull(half)
            scene.hull_points, scene.hull_planes = pts, planes
            scene.sides[:, 0] = half * np.sqrt(3.0)
            scene.mass[:] = (2 * half) ** 3
            scene.inertia[:] = scene.mass[0, 0] * (2 * half) ** 2 / 6.0
            scene.pos[:, 1] = rng.uniform(half + 0.05, half + 1.0, scene.n)
        nh = scene.n
    scene.pos[:, 0] *= rng.uniform(0.6, 1.0); scene.pos[:, 2] *= rng.uniform(0.6, 1.0)
    statics = list(scene.static_boxes or [])
    for k in range(scene.n):
        if rng.uniform() < 0.5:
            yaw = rng.uniform(0, np.pi)
            c, s = np.cos(yaw), np.sin(yaw)
            R12 = np.array([c, 0.0, s, 0.0, 0.0, 1.0, 0.0, 0.0, -s, 0.0, c, 0.0])
            sz = rng.uniform(0.1, 0.6, 3)
            statics.append((tuple(sz), (scene.pos[k, 0] + rng.uniform(-0.6, 0.6), sz[1] / 2, scene.pos[k, 2] + rng.uniform(-0.6, 0.6)), R12))
    if len(statics) > 60:
        statics = statics[:60]
    # the whole scene, far from the origin (the plane stays y = 0)
    scene.pos = scene.pos + off
    statics = [(sd, (p[0] + off[0], p[1], p[2] + off[2]), R) for sd, p, R in statics]
    scene.static_boxes = statics
    scene = scene.astype(dtype)
    steps = int(rng.integers(120, 241))
    ow = Oracle(dtype).world()
    if scene.plane is not None:
        ow.add_plane(*scene.plane)
    ow.set_hull(scene.hull_points)
    ow.set_hull_faces(scene.hull_planes)
    for sides, pos, R12 in statics:
        ow.add_static_box(sides, pos, R12)
    ow.add_convex(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia)
    first = len(statics) + (1 if scene.plane is not None else 0)
    for g in range(first, first + scene.n):
        ow.lib.orc_geom_set_collide_bits(ow.w, g, 1)
    most = 0
    for _ in range(steps):
        ow.tick(H)
        most = max(most, ow.n_contacts())
    w = pkg.BatchWorld(scene.n, dtype=dtype)
    w.load_scene(scene)
    w.set_class_pairs(pkg.scenes.GEOM_CONVEX, pkg.scenes.GEOM_CONVEX, False)
    w.step(H, steps)
    ok = True
    for name, a, b in zip(("pos", "quat", "lvel", "avel"), w.state(), ow.state()):
        if not (np.all(np.isfinite(a)) and np.array_equal(a, b)):
            ok = False
            print(f"FAIL seed {seed} {dtype} {name}: max abs diff {np.max(np.abs(a - b))} bodies {scene.n} off {off} plane {plane}", flush=True)
            break
    w.close()
    return ok, most, scene.n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
    gold = np.load(os.path.join(ROOT, "tests", "golden", "teapot_hull.npz"))
    teapot = pkg.hull.build(gold["points"], 0.01)
    bad, contacts, bodies = 0, 0, 0
    t0 = time.time()
    for seed in (only if only else range(first, first + n)):
        ok, most, nb = one(seed, teapot)
        bad += 0 if ok else 1
        contacts += most; bodies += nb
        if seed % 25 == 0:
            print("seed", seed, f"{time.time() - t0:.0f}s", flush=True)
    print(f"hull campaign: {n} scenes ({bodies} hulls, {contacts} contacts at their fullest ticks), {bad} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
