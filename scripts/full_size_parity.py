"""One-off evidence run: the HIP path against the oracle at BASELINE.json's FULL sizes, bit for bit (the test suite
checks full sizes through properties and parity at sizes the oracle finishes in seconds; this takes minutes of host
CPU).  Output goes to profiles/r01_full_size_parity.txt."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
from oracle.orc_ctypes import Oracle
H = 1 / 60


def run(name, scene, dtype, steps, setup=None):
    scene = scene.astype(dtype)
    w = pkg.BatchWorld(scene.n, dtype=dtype)
    if setup:
        setup(w)
    w.load_scene(scene)
    t0 = time.perf_counter(); w.step(H, steps); w.synchronize(); tg = time.perf_counter() - t0
    orc = Oracle(dtype); ow = orc.world()
    if scene.plane is not None:
        ow.add_plane(*scene.plane)
    if scene.hull_points is not None:
        ow.set_hull(scene.hull_points)
        ow.add_convex(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia)
    else:
        ow.add_boxes(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia, scene.sides)
    t0 = time.perf_counter(); ow.run(H, steps); tc = time.perf_counter() - t0
    same = all(np.array_equal(a, r) for a, r in zip(w.state(), ow.state()))
    worst = max(float(np.max(np.abs(a - r))) for a, r in zip(w.state(), ow.state()))
    print(f"{name:44s} {dtype:8s} {scene.n:8d} bodies x {steps:4d} ticks: bit-identical={same} (max abs diff {worst:.3g}); "
          f"contacts gpu/oracle {w.last_contact_count()}/{ow.n_contacts()}; gpu {tg:.2f} s, oracle {tc:.1f} s", flush=True)
    w.close()
    return same


ok = True
ok &= run("configs[1] 1 048 576 free boxes", pkg.scenes.config2(), "float32", 200)
ok &= run("configs[1] 1 048 576 free boxes", pkg.scenes.config2(), "float64", 100)
ok &= run("configs[1], 32 ticks per launch", pkg.scenes.config2(), "float32", 200, setup=lambda w: w.set_ticks_per_launch(32))
ok &= run("configs[2] 262 144 boxes on the plane", pkg.scenes.config3(), "float32", 240)
gold = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden", "teapot_hull.npz"))
hull = pkg.hull.build(gold["points"], 0.01)
ok &= run("configs[4] 16 384 teapot hulls on the plane", pkg.scenes.hull_grid(hull, 128, 128, seed=1, y_range=(0.6, 1.6), tilt=0.2), "float32", 200)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
