import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
H = 1.0 / 60.0
for nx, nz, coll in ((64, 64, True), (64, 64, False), (1024, 128, True), (1024, 128, False)):
    scene = pkg.scenes.box_grid(nx, nz, seed=1, spin=True, plane=False).astype("float32")
    w = pkg.BatchWorld(scene.n, dtype="float32"); w.load_scene(scene)
    if not coll: w.set_body_collisions(False)
    w.step(H, 600); w.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); w.step(H, 2048); t1 = time.perf_counter(); w.synchronize(); t2 = time.perf_counter()
        best = min(best, (t2 - t0) / 2048 * 1e6); enq = (t1 - t0) / 2048 * 1e6
    print(f"{scene.n:8d} bodies, collision proof {'on ' if coll else 'off'}: {best:6.2f} us/tick (host enqueue {enq:5.2f} us/tick)", flush=True)
    w.close()
