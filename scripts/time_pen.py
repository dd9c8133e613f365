"""the reference's own scene (floor + three walls as static boxes, boxes and spheres spawned above: main.c:115-121, 502-521) through the
batch path: us per tick and what kind of ticks they were.  usage: time_pen.py [bodies ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
H = 1 / 60
for n_spawn in [int(a) for a in sys.argv[1:]] or [96, 400, 1000]:
    spawn = pkg.scenes.reference_spawn(n_spawn, seed=7, y_range=(3.0, 12.0))
    spawn.sort(key=lambda s: -s[0])
    n = len(spawn)
    sc = pkg.scenes.Scene(np.array([s[2] for s in spawn], float), np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros((n, 3)), np.zeros((n, 3)),
                          np.ones((n, 1)), np.ones((n, 3)), np.array([s[1] for s in spawn], float),
                          np.array([s[0] for s in spawn], np.uint8), None).astype("float32")
    w = pkg.BatchWorld(n, dtype="float32"); w.load_scene(sc); w.set_static_boxes(pkg.scenes.reference_map())
    w.step(H, 120); w.synchronize()                  # the bodies are down
    t0 = time.perf_counter(); w.step(H, 480); w.synchronize(); dt = time.perf_counter() - t0
    print(f"pen, {n} bodies: {dt / 480 * 1e6:.1f} us/tick", w.collision_stats(), flush=True)
    w.close()
