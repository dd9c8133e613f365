import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
H = 1/60
for dtype in ("float32", "float64"):
    scene = pkg.scenes.config1().astype(dtype)
    w = pkg.BatchWorld(scene.n, dtype=dtype); w.load_scene(scene)
    w.step(H, 10); w.synchronize()
    t0 = time.perf_counter(); w.step(H, 600); w.synchronize(); dt = time.perf_counter() - t0
    print(dtype, "config1 1024 boxes: %.1f us/tick, %.1f M body-steps/s" % (dt/600*1e6, 1024*600/dt/1e6), w.collision_stats())
from oracle.orc_ctypes import Oracle
orc = Oracle("float32"); ow = orc.world(); sc = pkg.scenes.config1().astype("float32")
ow.add_plane(*sc.plane); ow.add_boxes(sc.pos, sc.quat, sc.lvel, sc.avel, sc.mass[:,0], sc.inertia, sc.sides)
t = ow.run(H, 610); print("oracle f32: %.1f us/tick, %.2f M body-steps/s" % (t/610*1e6, 1024*610/t/1e6))
