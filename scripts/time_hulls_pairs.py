"""16 384 teapot hulls on a static box floor, hull-hull pairs ON (every class collides with every class): ms per tick at a pitch.
usage: time_hulls_pairs.py [pitch=4.5]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package, ROOT
pkg = load_package()
H = 1 / 60
pitch = float(sys.argv[1]) if len(sys.argv) > 1 else 4.5
gold = np.load(os.path.join(ROOT, "tests", "golden", "teapot_hull.npz"))
hull = pkg.hull.build(gold["points"], 0.01)
scene = pkg.scenes.hull_grid(hull, 128, 128, seed=1, y_range=(0.6, 1.6), spin=False, tilt=0.2, floor_box=True, pitch=pitch).astype("float32")
w = pkg.BatchWorld(scene.n, dtype="float32"); w.load_scene(scene)
w.step(H, 120); w.synchronize()
t0 = time.perf_counter(); w.step(H, 240); w.synchronize(); dt = time.perf_counter() - t0
print(f"{scene.n} teapot hulls, {pitch} m apart, static box floor, hull pairs on: {dt/240*1e3:.3f} ms/tick  {w.collision_stats()}", flush=True)
