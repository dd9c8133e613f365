"""A slab past 2^32 elements (150 994 944 free bodies: 4.5e9 reals per slab, 36 GB for the two): the state after 40 ticks
against the oracle on a strided sample of bodies (free bodies never interact: a body's trajectory does not depend on the
others), and the size-independent properties for all of them.  One-off: 60 GB of host arrays."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
from oracle.orc_ctypes import Oracle
side = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
H, steps = 1 / 60, 40
t0 = time.time()
scene = pkg.scenes.box_grid(side, side, seed=1, spin=True, plane=False).astype("float32")
print(f"scene of {scene.n} bodies built in {time.time()-t0:.0f} s", flush=True)
w = pkg.BatchWorld(scene.n, dtype="float32"); w.load_scene(scene)
t0 = time.time(); w.step(H, steps); w.synchronize(); dt = time.time() - t0
print(f"{steps} ticks: {dt/steps*1e3:.2f} ms/tick  {w.collision_stats()}", flush=True)
pos, quat, lvel, avel = w.state()
assert np.array_equal(pos[:, 0], scene.pos[:, 0]) and np.array_equal(pos[:, 2], scene.pos[:, 2])
assert np.allclose(lvel[:, 1], steps * H * -9.8, rtol=1e-5)
assert np.all(np.abs(np.linalg.norm(quat[::97].astype(np.float64), axis=1) - 1.0) < 2e-6)
sel = np.concatenate([np.arange(0, scene.n, 65521), np.arange(scene.n - 4096, scene.n)])       # a stride across the slab and its very end
sub = pkg.scenes.Scene(scene.pos[sel], scene.quat[sel], scene.lvel[sel], scene.avel[sel], scene.mass[sel],
                       scene.inertia[sel], scene.sides[sel], scene.gtype[sel], None)
orc = Oracle("float32"); ow = orc.world()
ow.add_boxes(sub.pos, sub.quat, sub.lvel, sub.avel, sub.mass[:, 0], sub.inertia, sub.sides)
ow.run(H, steps)
for name, a, b in zip(("pos", "quat", "lvel", "avel"), (pos, quat, lvel, avel), ow.state()):
    assert np.array_equal(a[sel], b), name
print(f"{len(sel)} sampled bodies bit-identical to the oracle; x/z untouched and v_y = n h g for all {scene.n}")
