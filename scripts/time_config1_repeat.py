import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
H = 1/60
for dtype in ("float32", "float64", "float32", "float32"):
    scene = pkg.scenes.config1().astype(dtype)
    w = pkg.BatchWorld(scene.n, dtype=dtype); w.load_scene(scene)
    w.step(H, 10); w.synchronize()
    t0 = time.perf_counter(); w.step(H, 600); w.synchronize(); dt = time.perf_counter() - t0
    print(dtype, "config1: %.1f us/tick" % (dt/600*1e6), flush=True)
    w.close()
