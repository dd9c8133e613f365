"""step_plane: how much of the tick is the 20 SOR sweeps?  configs[2]'s scene (262 144 boxes resting on the plane) timed at
0 / 10 / 20 / 40 QuickStep iterations: the slope is the sweeps, the intercept everything else (loads, narrowphase, row set-up,
integration, stores, launch)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
H = 1 / 60
scene = pkg.scenes.config3().astype("float32")
for dtype in ("float32",):
    for iters in (20, 0, 10, 20, 40):
        w = pkg.BatchWorld(scene.n, dtype=dtype); w.load_scene(scene)
        w.step(H, 150); w.synchronize()               # landed, with the default 20 iterations
        w.set_quickstep(iters)
        w.step(H, 20); w.synchronize()
        ms = w.step_timed(H, 200)
        print(f"{dtype} iters {iters:2d}: {ms / 200 * 1e3:7.2f} us/tick (HIP events)  contacts {w.last_contact_count()}", flush=True)
        w.close()
