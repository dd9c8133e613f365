"""Batch path on static-box scenes: ms per tick.  `python scripts/time_floor.py [floor] [hulls_plane] [hulls_floor]` (default: all)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package, ROOT
pkg = load_package()
H = 1 / 60
which = set(sys.argv[1:]) or {"floor", "hulls_plane", "hulls_floor"}
if "floor" in which:
    for side in (100, 320):
        scene = pkg.scenes.box_grid(side, side, seed=4, y_range=(1.2, 2.0), spin=False, plane=False).astype("float32")
        w = pkg.BatchWorld(scene.n, dtype="float32"); w.load_scene(scene)
        w.set_static_boxes([((1000.0, 1.0, 1000.0), (0.0, 0.0, 0.0), pkg.scenes._rot_z(0.0))])
        w.step(H, 60); w.synchronize()
        t0 = time.perf_counter(); w.step(H, 240); w.synchronize(); dt = time.perf_counter() - t0
        print(f"{scene.n:7d} boxes resting on a static box floor: {dt/240*1e3:8.3f} ms/tick  contacts {w.last_contact_count()}  {w.collision_stats()}", flush=True)
        w.close()
gold = np.load(os.path.join(ROOT, "tests", "golden", "teapot_hull.npz"))
hull = pkg.hull.build(gold["points"], 0.01)
for floor_box, pitch in ((False, 4.5), (True, 4.5), (True, 3.0)):
    if ("hulls_floor" if floor_box else "hulls_plane") not in which:
        continue
    scene = pkg.scenes.hull_grid(hull, 128, 128, seed=1, y_range=(0.6, 1.6), spin=False, tilt=0.2, floor_box=floor_box, pitch=pitch).astype("float32")
    for map_only in (True, False):
        # map_only: BASELINE configs[4] as worded (box-trimesh contacts with the floor only: hull-hull pairs switched off, bench.py's leg)
        w = pkg.BatchWorld(scene.n, dtype="float32"); w.load_scene(scene)
        if map_only:
            w.set_class_pairs(pkg.scenes.GEOM_CONVEX, pkg.scenes.GEOM_CONVEX, False)
        w.step(H, 120); w.synchronize()
        t0 = time.perf_counter(); w.step(H, 240); w.synchronize(); dt = time.perf_counter() - t0
        print(f"{scene.n:7d} teapot hulls, {pitch} m apart, on {'a static box floor' if floor_box else 'the ground plane'}"
              f"{' (hull-hull pairs off)' if map_only else ''}: {dt/240*1e3:8.3f} ms/tick  contacts {w.last_contact_count()}  {w.collision_stats()}", flush=True)
        w.close()
