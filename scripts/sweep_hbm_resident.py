"""A/B table for integrate_free at an HBM-resident size (16 Mi f32 bodies: 1.1 GB touched per tick, Infinity Cache 256 MB):
in place vs out of place (alternating slabs), non-temporal stores / loads, bodies per lane.  The plain loop (no collision
proof) so that the knobs act on every tick.  Prints one line per variant: us per tick, algorithmic TB/s, fraction of 8 TB/s."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package
pkg = load_package()
H = 1.0 / 60.0
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "float32"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
base = pkg.scenes.box_grid(1024, 1024, seed=1, spin=True, plane=False).astype(dtype)
reps = max(1, side * side // base.n)
cat = lambda a: np.concatenate([a] * reps, axis=0)
pos = cat(base.pos); pos[:, 0] += np.repeat(np.arange(reps) * 2600.0, base.n).astype(pos.dtype)
scene = type(base)(pos, cat(base.quat), cat(base.lvel), cat(base.avel), cat(base.mass), cat(base.inertia), cat(base.sides),
                   np.concatenate([base.gtype] * reps), None, None)
rs = np.dtype(dtype).itemsize
print(f"# {scene.n} bodies {dtype}, {steps} ticks per timing, plain loop (collision proof off)", flush=True)
variants = []
for oop, nt, vec in itertools.product((0, 1), (0, 1, 2, 3), (1, 4 if rs == 4 else 2)):
    variants.append({"DMX_OOP": oop, "DMX_NT": nt, "DMX_VEC": vec})
for mw in (6, 8):
    variants.append({"DMX_OOP": 1, "DMX_NT": 1, "DMX_VEC": 1, "DMX_MIN_WAVES": mw})
for v in variants:
    for k, val in v.items():
        os.environ[k] = str(val)
    os.environ.setdefault("DMX_MIN_WAVES", "0")
    if "DMX_MIN_WAVES" not in v:
        os.environ["DMX_MIN_WAVES"] = "0"
    w = pkg.BatchWorld(scene.n, dtype=dtype)
    w.load_scene(scene)
    w.set_body_collisions(False)
    w.step(H, 10); w.synchronize()
    best = 1e9
    for _ in range(3):
        ms = w.step_timed(H, steps)
        best = min(best, ms / steps * 1e3)
    w.close()
    gbs = 30 * rs * scene.n / (best * 1e-6) / 1e9
    print(f"oop={v['DMX_OOP']} nt={v['DMX_NT']} vec={v['DMX_VEC']} minw={os.environ['DMX_MIN_WAVES']}: {best:8.2f} us/tick  {gbs/1e3:5.2f} TB/s  frac {gbs/8000:.3f}", flush=True)
