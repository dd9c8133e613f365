"""Round 4 campaign far from the origin: boxes dropped on a static floor box (or the ground plane) strewn with turned blocks and planks,
piling into one another, the whole scene shifted up to 8 km from the origin -- where f32 rounds positions to 0.1-0.5 mm and every place
in the pipeline that answers one geometric question two ways (an AABB here, a collider there) may answer it differently from the oracle.
Fused static path, exact ticks with body pairs, islands: state after 150-300 ticks bit-identical to the oracle, both precisions.
usage: python scripts/fuzz_far_r04.py [n_seeds=200] [first_seed=0] [seed,seed,...]"""
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


def one(seed):
    rng = np.random.default_rng(7000 + seed)
    dtype = "float32" if seed % 2 else "float64"
    plane = bool(rng.integers(0, 2))
    nx, nz = int(rng.integers(2, 9)), int(rng.integers(2, 9))
    off = np.array([rng.choice([0.0, 911.0, 4096.0, -2500.0, 8000.0]), 0.0, rng.choice([0.0, -63.0, 1500.0, 7000.0])])
    scene = pkg.scenes.box_grid(nx, nz, seed=seed + 1, y_range=(0.6, 4.0), spin=True, box_mass=bool(rng.integers(0, 2)), plane=plane)
    scene.pos[:, 0] *= rng.uniform(0.25, 1.0); scene.pos[:, 2] *= rng.uniform(0.25, 1.0)        # tight grids: boxes land on one another
    statics = []
    if not plane:
        span = max(nx, nz) * 3.0 + 20.0
        statics.append(((span, 1.0, span), (0.0, -0.5, 0.0), np.array([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0])))
    for k in range(scene.n):
        if rng.uniform() < 0.4 and len(statics) < 60:
            yaw = rng.uniform(0, np.pi)
            c, s = np.cos(yaw), np.sin(yaw)
            R12 = np.array([c, 0.0, s, 0.0, 0.0, 1.0, 0.0, 0.0, -s, 0.0, c, 0.0])
            sz = rng.uniform(0.1, 1.2, 3)
            statics.append((tuple(sz), (scene.pos[k, 0] + rng.uniform(-0.6, 0.6), sz[1] / 2, scene.pos[k, 2] + rng.uniform(-0.6, 0.6)), R12))
    scene.pos = scene.pos + off
    statics = [(sd, (p[0] + off[0], p[1], p[2] + off[2]), R) for sd, p, R in statics]
    scene = scene.astype(dtype)
    steps = int(rng.integers(150, 301))
    ow = Oracle(dtype).world()
    if scene.plane is not None:
        ow.add_plane(*scene.plane)
    for sides, pos, R12 in statics:
        ow.add_static_box(sides, pos, R12)
    ow.add_boxes(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia, scene.sides)
    most = 0
    for _ in range(steps):
        ow.tick(H)
        most = max(most, ow.n_contacts())
    w = pkg.BatchWorld(scene.n, dtype=dtype)
    w.load_scene(scene)
    if statics:
        w.set_static_boxes(statics)
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
    bad, contacts, bodies = 0, 0, 0
    t0 = time.time()
    for seed in (only if only else range(first, first + n)):
        ok, most, nb = one(seed)
        bad += 0 if ok else 1
        contacts += most; bodies += nb
        if seed % 50 == 0:
            print("seed", seed, f"{time.time() - t0:.0f}s", flush=True)
    print(f"far-from-the-origin campaign: {len(only) if only else n} scenes ({bodies} boxes, {contacts} contacts at their fullest ticks), {bad} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
