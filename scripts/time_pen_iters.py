"""the pen's big island: how much of the tick is the SOR sweeps?  The settled pen timed at 0 / 10 / 20 / 40 iterations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
H = 1 / 60
n_spawn = int(sys.argv[1]) if len(sys.argv) > 1 else 400
spawn = pkg.scenes.reference_spawn(n_spawn, seed=7, y_range=(3.0, 12.0))
spawn.sort(key=lambda s: -s[0])
n = len(spawn)
sc = pkg.scenes.Scene(np.array([s[2] for s in spawn], float), np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros((n, 3)), np.zeros((n, 3)),
                      np.ones((n, 1)), np.ones((n, 3)), np.array([s[1] for s in spawn], float),
                      np.array([s[0] for s in spawn], np.uint8), None).astype("float32")
for iters in (20, 0, 10, 20, 40):
    w = pkg.BatchWorld(n, dtype="float32"); w.load_scene(sc); w.set_static_boxes(pkg.scenes.reference_map())
    w.step(H, 400); w.synchronize()
    w.set_quickstep(iters)
    w.step(H, 10); w.synchronize()
    t0 = time.perf_counter(); w.step(H, 100); w.synchronize(); dt = time.perf_counter() - t0
    print(f"iters {iters:2d}: {dt / 100 * 1e6:8.1f} us/tick  contacts {w.last_contact_count()}", flush=True)
    w.close()
