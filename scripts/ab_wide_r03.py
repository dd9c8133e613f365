"""integrate_free against its two experimental forms at 1 Mi and 16 Mi bodies: DMX_WIDE=1 (the tile moved in 16-byte pieces
through LDS) and DMX_WIDE=2 (reads by LDS-DMA into a double buffer, persistent grid, DMX_WIDE_BLOCKS workgroups per CU), in
place and with the slabs alternating every tick (DMX_OOP).  Plain loop (collision proof off) so the knobs act on every tick;
also checks that every form gives the same bits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
H = 1.0 / 60.0
dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
base = pkg.scenes.box_grid(1024, 1024, seed=1, spin=True, plane=False).astype(dtype)
rs = np.dtype(dtype).itemsize
for side in (1024, 4096):
    reps = max(1, side * side // base.n)
    cat = lambda a: np.concatenate([a] * reps, axis=0)
    pos = cat(base.pos); pos[:, 0] += np.repeat(np.arange(reps) * 2600.0, base.n).astype(pos.dtype)
    scene = type(base)(pos, cat(base.quat), cat(base.lvel), cat(base.avel), cat(base.mass), cat(base.inertia), cat(base.sides),
                       np.concatenate([base.gtype] * reps), None, None)
    ref = None
    for wide, blocks, oop in ((0, 4, 0), (1, 4, 0), (2, 2, 0), (2, 3, 0), (2, 4, 0), (2, 6, 0), (2, 8, 0), (0, 4, 1), (2, 4, 1), (2, 8, 1)):
        os.environ["DMX_WIDE"] = str(wide); os.environ["DMX_OOP"] = str(oop); os.environ["DMX_WIDE_BLOCKS"] = str(blocks)
        w = pkg.BatchWorld(scene.n, dtype=dtype)
        w.load_scene(scene)
        w.set_body_collisions(False)
        w.step(H, 10); w.synchronize()
        state = np.concatenate([w.download(pkg.batch.STATE, 0, 4096), w.download(pkg.batch.STATE, scene.n - 4096, 4096)])
        if ref is None:
            ref = state
        same = np.array_equal(state, ref)
        best = 1e9
        for _ in range(3):
            ms = w.step_timed(H, 100)
            best = min(best, ms / 100 * 1e3)
        w.close()
        gbs = 30 * rs * scene.n / (best * 1e-6) / 1e9
        print(f"{scene.n:9d} bodies {dtype} wide={wide} blocks/CU={blocks} oop={oop}: {best:8.2f} us/tick  {gbs/1e3:5.2f} TB/s  frac {gbs/8000:.3f}  same bits: {same}", flush=True)
