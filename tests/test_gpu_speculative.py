"""The small-scene exact tick enqueues the island solve and the fused step BEHIND its bookkeeping kernels, gated by the record
those leave on the device (dmx_general.cpp: careful_tick; ExactCounts::spec_ok).  Same ticks with the speculation off
(DMX_SPECULATE=0: the host waits for the counts, then launches) must give the same bits -- and so must ticks whose speculation
is refused on the device (singles at static boxes: not solve_island_wg's kind)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUNNER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from __graft_entry__ import load_package
pkg = load_package()
kind, out = sys.argv[2], sys.argv[3]
H = 1.0 / 60.0
if kind == "plane":
    scene = pkg.scenes.box_grid(16, 16, seed=5, y_range=(0.8, 3.0), spin=True, box_mass=True).astype("float32")
    scene.pos[:, [0, 2]] *= 0.55           # close enough to topple into one another
    w = pkg.BatchWorld(scene.n, dtype="float32"); w.load_scene(scene)
else:
    spawn = pkg.scenes.reference_spawn(64, seed=3, y_range=(3.0, 9.0))
    spawn.sort(key=lambda s: -s[0])
    n = len(spawn)
    scene = pkg.scenes.Scene(np.array([s[2] for s in spawn], float), np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros((n, 3)), np.zeros((n, 3)),
                             np.ones((n, 1)), np.ones((n, 3)), np.array([s[1] for s in spawn], float),
                             np.array([s[0] for s in spawn], np.uint8), None).astype("float32")
    w = pkg.BatchWorld(n, dtype="float32"); w.load_scene(scene); w.set_static_boxes(pkg.scenes.reference_map())
w.step(H, 300)
st = w.collision_stats()
pos, quat, lvel, avel = w.state()
np.savez(out, pos=pos, quat=quat, lvel=lvel, avel=avel, careful=st["careful_ticks"], spec=st["speculated_ticks"], pairs=st["pair_ticks"])
"""


def _run(tmp_path, kind, speculate):
    out = str(tmp_path / f"{kind}_{speculate}.npz")
    env = dict(os.environ, DMX_SPECULATE=str(speculate))
    p = subprocess.run([sys.executable, "-c", RUNNER, ROOT, kind, out], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    return np.load(out)


@pytest.mark.parametrize("kind", ["plane", "pen"])
def test_speculative_exact_ticks_give_the_waiting_ticks_bits(tmp_path, kind):
    a, b = _run(tmp_path, kind, 1), _run(tmp_path, kind, 0)
    for k in ("pos", "quat", "lvel", "avel"):
        assert np.array_equal(a[k], b[k]), k
    assert int(a["careful"]) == int(b["careful"]) > 0 and int(a["pairs"]) == int(b["pairs"])
    assert int(b["spec"]) == 0
    if kind == "plane":
        assert int(a["spec"]) > 0 and int(a["pairs"]) > 0        # islands of toppled boxes, solved without the host round trip
    else:
        # bodies at the pen's floor and walls are one-body islands (solve_singles' kind): the device refuses those ticks'
        # speculation and the host launches them as it always did
        assert int(a["spec"]) < int(a["careful"])
