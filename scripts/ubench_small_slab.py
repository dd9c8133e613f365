"""Launch-bound regime of the strong-scaled slab (configs[3] at 8 GPUs: 131 072 bodies per GPU): us per tick of the plain loop
issued eagerly, replayed from a captured HIP graph of 32 ticks, and with 2 / 4 ticks per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package
pkg = load_package()
H = 1.0 / 60.0
for nx, nz in ((1024, 128), (1024, 256), (1024, 512)):
    scene = pkg.scenes.box_grid(nx, nz, seed=1, spin=True, plane=False).astype("float32")
    w = pkg.BatchWorld(scene.n, dtype="float32")
    w.load_scene(scene)
    w.set_body_collisions(False)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        w.set_stream(stream.cuda_stream)
        w.step(H, 64); torch.cuda.synchronize()
        def timed(fn, ticks):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / ticks)
            return best
        eager = timed(lambda: w.step(H, 512), 512)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=stream):
                w.step(H, 32)
            graph = timed(lambda: [g.replay() for _ in range(16)], 512)
        except Exception as e:      # noqa: BLE001
            graph = float("nan"); print("graph capture failed:", e)
        res = {}
        for tpl in (2, 4, 8):
            w.set_ticks_per_launch(tpl)
            res[tpl] = timed(lambda: w.step(H, 512), 512)
        w.set_ticks_per_launch(1)
    print(f"{scene.n:8d} bodies: eager {eager:6.2f} us/tick, graph(32) {graph:6.2f}, " + ", ".join(f"{t} ticks/launch {v:5.2f}" for t, v in res.items()), flush=True)
    w.close()
