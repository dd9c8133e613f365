import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
H = 1 / 60
side = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = pkg.scenes.box_grid(3 * side, 3 * side, seed=5, y_range=(0.6, 6.0), spin=True, box_mass=True).astype("float32")
ix = (np.arange(scene.n) % (3 * side)); iz = (np.arange(scene.n) // (3 * side))
scene.pos[:, 0] = (ix // 3) * 7.5 + (ix % 3) * 0.6
scene.pos[:, 2] = (iz // 3) * 7.5 + (iz % 3) * 0.6
w = pkg.BatchWorld(scene.n, dtype="float32")
w.load_scene(scene)
w.step(H, 120); w.synchronize()
t0 = time.perf_counter(); w.step(H, 100); w.synchronize(); dt = time.perf_counter() - t0
print(f"{scene.n} bodies: {dt/100*1e3:.3f} ms/tick", w.collision_stats())
w.close()
