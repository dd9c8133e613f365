"""Why does the collision-checked loop run slower than the plain loop at 16 Mi bodies?  Same scene, same kernel:
plain loop / checked loop, the batch's own stream / a torch stream, lazy chunks on / off, ping-pong / copy snapshot."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package
pkg = load_package()
H = 1.0 / 60.0
base = pkg.scenes.box_grid(1024, 1024, seed=1, spin=True, plane=False).astype("float32")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cat = lambda a: np.concatenate([a] * reps, axis=0)
pos = cat(base.pos); pos[:, 0] += np.repeat(np.arange(reps) * 2600.0, base.n).astype(pos.dtype)
scene = type(base)(pos, cat(base.quat), cat(base.lvel), cat(base.avel), cat(base.mass), cat(base.inertia), cat(base.sides),
                   np.concatenate([base.gtype] * reps), None, None)
def run(label, collide, torch_stream, env=None, snapshot=None, steps=200):
    for k, v in (env or {}).items():
        os.environ[k] = v
    w = pkg.BatchWorld(scene.n, dtype="float32")
    for k in (env or {}):
        del os.environ[k]
    w.load_scene(scene)
    w.set_body_collisions(collide)
    if snapshot is not None:
        w.set_snapshot_mode(snapshot)
    stream = torch.cuda.Stream()
    if torch_stream:
        w.set_stream(stream.cuda_stream)
    w.step(H, 40); w.synchronize()
    best = []
    for _ in range(3):
        best.append(w.step_timed(H, steps) / steps * 1e3)
    st = w.collision_stats()
    w.close()
    print(f"{label:58s} {min(best):8.2f} us/tick (runs {', '.join(f'{b:.1f}' for b in best)})  frac {30*4*scene.n/(min(best)*1e-6)/8e12:.3f}  {st}", flush=True)
print(f"# {scene.n} bodies f32")
run("plain loop, own stream", False, False)
run("plain loop, torch stream", False, True)
run("checked loop (lazy chunks, ping-pong), own stream", True, False)
run("checked loop (lazy chunks, ping-pong), torch stream", True, True)
run("checked loop, DMX_LAZY_CHUNKS=0", True, False, env={"DMX_LAZY_CHUNKS": "0"})
run("checked loop, copy snapshot", True, False, snapshot=1)
run("plain loop again", False, False)
