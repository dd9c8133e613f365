"""Batch path on scenes with persistent body-body contacts (every tick is an exact tick): ms per tick by body count.
Many small piles: a grid of 3 x 3 clusters, each cluster 9 boxes dropped on top of one another."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
H = 1 / 60
for side in (16, 32, 64, 128):
    scene = pkg.scenes.box_grid(3 * side, 3 * side, seed=5, y_range=(0.6, 6.0), spin=True, box_mass=True).astype("float32")
    # squeeze every 3 x 3 block of the grid into a tight cluster (0.6 m pitch inside, clusters 7.5 m apart)
    ix = (np.arange(scene.n) % (3 * side)); iz = (np.arange(scene.n) // (3 * side))
    scene.pos[:, 0] = (ix // 3) * 7.5 + (ix % 3) * 0.6
    scene.pos[:, 2] = (iz // 3) * 7.5 + (iz % 3) * 0.6
    w = pkg.BatchWorld(scene.n, dtype="float32")
    w.load_scene(scene)
    w.step(H, 120); w.synchronize()
    t0 = time.perf_counter(); w.step(H, 60); w.synchronize(); dt = time.perf_counter() - t0
    st = w.collision_stats()
    print(f"{scene.n:7d} bodies in {side*side} piles: {dt/60*1e3:8.3f} ms/tick  contacts {w.last_contact_count()}  pairs {st['last_pairs']}  {st}", flush=True)
    w.close()
