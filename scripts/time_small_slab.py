"""The strong-scaled slab of configs[3] at 8 GPUs (131 072 bodies) and its neighbours through dmxBatchStep with the collision proof
on (lazy ballistic chunks): us per tick.  DMX_TICK_GRAPH=0/1 in the environment selects launches or captured graphs."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
H = 1.0 / 60.0
for nx, nz in ((1024, 128), (1024, 256), (1024, 1024)):
    scene = pkg.scenes.box_grid(nx, nz, seed=1, spin=True, plane=False).astype("float32")
    w = pkg.BatchWorld(scene.n, dtype="float32"); w.load_scene(scene)
    w.step(H, 600); w.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); w.step(H, 2048); w.synchronize(); best = min(best, (time.perf_counter() - t0) / 2048 * 1e6)
    print(f"{scene.n:8d} bodies: {best:6.2f} us/tick  {w.collision_stats()}", flush=True)
    w.close()
