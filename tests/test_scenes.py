"""Host logic: vectorised scene PRNG vs the rand.c golden values and the C
restatement; scene generator invariants."""
import json
import os

import numpy as np

from __graft_entry__ import load_package

pkg = load_package()
Rand = pkg.rand.Rand


def test_numpy_rand_matches_golden():
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rand_golden.json")))
    for case in g["next"]:
        r = Rand(case["seed"])
        assert [r.next() for _ in case["values"]] == case["values"]
        r = Rand(case["seed"])
        assert list(r.next(len(case["values"]))) == case["values"]
    r = Rand(g["mixed"]["seed"])
    for call in g["mixed"]["calls"]:
        got = r.double(call["min"], call["max"]) if call["fn"] == "double" else r.int(call["min"], call["max"])
        assert got == call["value"]


def test_numpy_rand_block_matches_c_restatement(orc64):
    lib = orc64.lib
    lib.orc_ref_rand_seed(987654321)
    c = [lib.orc_ref_rand_next() for _ in range(5000)]
    r = Rand(987654321)
    a = r.next(3000)
    b = r.next(2000)                      # state carries across blocks
    assert list(a) + list(b) == c
    lib.orc_ref_rand_seed(7)
    r = Rand(7)
    d = r.double(20.0, 50.0, 100)
    assert [lib.orc_ref_rand_double(20.0, 50.0) for _ in range(100)] == list(d)


def test_scene_follows_reference_spawn_distribution():
    s = pkg.scenes.config1()
    assert s.n == 1024 and s.plane == (0.0, 1.0, 0.0, 0.0)
    assert s.sides.min() >= 0.2 and s.sides.max() <= 1.0          # main.c:508
    assert s.pos[:, 1].min() >= 20.0 and s.pos[:, 1].max() <= 50.0   # main.c:504
    assert np.all(s.mass == 1.0) and np.all(s.inertia == 1.0)     # dBodySetMass never called
    # first body's draws are the first outputs of Rand(1)
    r = Rand(1)
    assert s.sides[0, 0] == r.double(0.2, 1.0)
    # grid pitch keeps boxes apart: min centre distance 2.5 > max box diagonal sqrt(3)
    assert abs(s.pos[1, 0] - s.pos[0, 0]) == 2.5 and abs(s.pos[32, 2] - s.pos[0, 2]) == 2.5


def test_config4_slabs_are_disjoint():
    s = pkg.scenes.box_grid(16, 16, spin=True, plane=False, slabs=4, slab_gap=10.0)
    z = s.pos[:, 2].reshape(16, 16)[:, 0]
    gaps = np.diff(z)
    assert np.allclose(gaps[[3, 7, 11]], 12.5) and np.allclose(np.delete(gaps, [3, 7, 11]), 2.5)


def test_box_mass_variant():
    s = pkg.scenes.config1(box_mass=True)
    m = s.sides.prod(axis=1)
    assert np.allclose(s.mass[:, 0], m)
    assert np.allclose(s.inertia[:, 0], m / 12 * (s.sides[:, 1] ** 2 + s.sides[:, 2] ** 2))
    assert np.abs(s.avel).max() <= 1.0 and np.abs(s.avel).max() > 0.5


def test_reference_pen_is_the_map_and_the_spawner():
    """scenes.reference_pen: the reference's floor + walls (main.c:115-121) and n spawned bodies (main.c:502-521), boxes ahead of
    the spheres, unit mass and identity inertia as AddBody leaves them (main.c:695-733), nobody above MAX_BODIES' pile height"""
    scenes = pkg.scenes
    sc, boxes, nb = scenes.reference_pen(64, seed=3)
    assert sc.n == 64 and len(boxes) == len(scenes.reference_map()) == 4
    assert (sc.gtype[:nb] == scenes.GEOM_BOX).all() and (sc.gtype[nb:] == scenes.GEOM_SPHERE).all()
    assert np.all(sc.mass == 1.0) and np.all(sc.inertia == 1.0)
    assert np.all(np.abs(sc.pos[:, [0, 2]]) <= 4.0) and sc.pos[:, 1].min() >= 3.0 and sc.pos[:, 1].max() <= 12.0
    drawn = sorted(scenes.reference_spawn(64, seed=3, y_range=(3.0, 12.0)), key=lambda s: -s[0])
    assert np.allclose(sc.pos, np.array([s[2] for s in drawn]))
