"""GPU parity of the convex-hull colliders added in round 3 (hull against hull, sphere against hull: this repository's own,
defined in the oracle and mirrored one wavefront per pair in csrc/dmx_collide_wave.hpp): bit-identical to the oracle's
sequential walk, in scenes where the hulls really meet -- cube hulls stacked on one another, spheres dropped on teapots,
teapots tipped so far that they skid into their neighbours ("configs[4] with tilt 0.6")."""
import itertools
import os

import numpy as np
import pytest

from __graft_entry__ import ROOT, load_package

pkg = load_package()
pytestmark = pytest.mark.gpu
H = 1.0 / 60.0


def _orc(dtype):
    from oracle.orc_ctypes import Oracle
    return Oracle(dtype)


def _teapot():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "teapot_hull.npz"))
    return pkg.hull.build(gold["points"], 0.01)


def _oracle_for(dtype, scene, n_hulls):
    """plane, hull shape, statics, then the bodies in slot order: convex bodies first, spheres behind them"""
    ow = _orc(dtype).world()
    if scene.plane is not None:
        ow.add_plane(*scene.plane)
    ow.set_hull(scene.hull_points)
    ow.set_hull_faces(scene.hull_planes)
    for sides, pos, R12 in (scene.static_boxes or []):
        ow.add_static_box(sides, pos, R12)
    a = n_hulls
    ow.add_convex(scene.pos[:a], scene.quat[:a], scene.lvel[:a], scene.avel[:a], scene.mass[:a, 0], scene.inertia[:a])
    if a < scene.n:
        ow.add_spheres(scene.pos[a:], scene.quat[a:], scene.lvel[a:], scene.avel[a:], scene.mass[a:, 0], scene.inertia[a:], scene.sides[a:, 0])
    return ow


def _same(w, ow):
    for name, a, b in zip(("pos", "quat", "lvel", "avel"), w.state(), ow.state()):
        assert np.all(np.isfinite(a)), name
        assert np.array_equal(a, b), f"{name}: max abs diff {np.max(np.abs(a - b))}"


def _pair_kinds(ow, n_hulls):
    hh = hs = 0
    for j in ow.joints():
        if j[0] >= 0 and j[1] >= 0:
            if j[0] < n_hulls and j[1] < n_hulls:
                hh += 1
            elif (j[0] < n_hulls) != (j[1] < n_hulls):
                hs += 1
    return hh, hs


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_cube_hulls_stacked_and_struck_by_spheres(dtype):
    """piles of two cube hulls (the upper one shifted and turned a little, so vertices of each are inside the other) on the
    ground plane, a sphere dropped on every pile: hull-hull and sphere-hull contacts in both slot orders"""
    half = 0.4
    pts = np.array(list(itertools.product((-half, half), repeat=3)), float)
    planes = np.array([[(sg if a == k else 0.0) for k in range(3)] + [half] for a in range(3) for sg in (-1.0, 1.0)])
    rng = np.random.default_rng(8)
    piles = 12
    pos, quat = [], []
    for p in range(piles):
        x, z = 4.0 * (p % 4), 4.0 * (p // 4)
        pos.append((x, half - 0.001, z)); quat.append((1.0, 0.0, 0.0, 0.0))
    for p in range(piles):
        x, z = 4.0 * (p % 4), 4.0 * (p // 4)
        yaw = rng.uniform(-0.3, 0.3)
        pos.append((x + rng.uniform(-0.15, 0.15), 3 * half - 0.003, z + rng.uniform(-0.15, 0.15)))
        quat.append((np.cos(yaw / 2), 0.0, np.sin(yaw / 2), 0.0))
    nh = 2 * piles
    for p in range(piles):
        x, z = 4.0 * (p % 4), 4.0 * (p // 4)
        pos.append((x + rng.uniform(-0.2, 0.2), 4 * half + 0.6 + 0.1 * p, z + rng.uniform(-0.2, 0.2))); quat.append((1.0, 0.0, 0.0, 0.0))
    n = nh + piles
    sides = np.zeros((n, 3)); sides[:nh, 0] = half * np.sqrt(3.0); sides[nh:, 0] = 0.25
    mass = np.ones((n, 1)); mass[:nh] = (2 * half) ** 3
    inertia = np.ones((n, 3)); inertia[:nh] = mass[0, 0] * (2 * half) ** 2 / 6.0; inertia[nh:] = 0.4 * 0.25 ** 2
    gtype = np.full(n, pkg.scenes.GEOM_CONVEX, np.uint8); gtype[nh:] = pkg.scenes.GEOM_SPHERE
    scene = pkg.scenes.Scene(np.array(pos), np.array(quat), np.zeros((n, 3)), np.zeros((n, 3)), mass, inertia, sides, gtype,
                             (0.0, 1.0, 0.0, 0.0), pts, planes).astype(dtype)
    steps = 150
    ow = _oracle_for(dtype, scene, nh)
    hh = hs = 0
    for _ in range(steps):
        ow.tick(H)
        a, b = _pair_kinds(ow, nh)
        hh += a; hs += b
    assert hh > 100 and hs > 20, (hh, hs)
    w = pkg.BatchWorld(n, dtype=dtype)
    w.load_scene(scene)
    w.step(H, steps)
    _same(w, ow)
    st = w.collision_stats()
    assert st["pair_ticks"] > 0 and st["unsupported_pairs"] == 0
    w.close()


def test_teapots_tipped_until_they_skid_into_their_neighbours():
    """configs[4]'s scene at a reduced size with tilt 0.6 and spin: tipped teapots roll and skid across the static floor into one
    another -- hull-hull pairs with the full 1 265-point, 2 526-face hull, in islands that also hold floor contacts"""
    hull = _teapot()
    scene = pkg.scenes.hull_grid(hull, 6, 6, seed=11, y_range=(0.7, 1.4), spin=True, tilt=0.6, floor_box=True).astype("float64")
    scene.pos[:, 0] *= 0.62; scene.pos[:, 2] *= 0.62                  # a tighter grid: 1.86 m pitch against a 2.13 m bounding sphere
    scene.lvel[:, 0] = np.random.default_rng(4).uniform(-1.0, 1.0, scene.n)
    steps = 200
    ow = _oracle_for("float64", scene, scene.n)
    hh = 0
    for _ in range(steps):
        ow.tick(H)
        hh += _pair_kinds(ow, scene.n)[0]
    assert hh > 30, hh
    w = pkg.BatchWorld(scene.n, dtype="float64")
    w.load_scene(scene)
    w.step(H, steps)
    _same(w, ow)
    assert w.collision_stats()["unsupported_pairs"] == 0
    w.close()


def test_spheres_dropped_on_teapots():
    hull = _teapot()
    hs = pkg.scenes.hull_grid(hull, 4, 4, seed=6, y_range=(0.45, 0.55), spin=False, tilt=0.0)
    n_h = hs.n
    rng = np.random.default_rng(2)
    sp_pos = hs.pos + np.array([0.0, 1.2, 0.0]) + rng.uniform(-0.25, 0.25, hs.pos.shape) * np.array([1.0, 0.0, 1.0])
    n = 2 * n_h
    cat = lambda a, b: np.concatenate([a, b])
    sides = cat(hs.sides, np.tile([0.2, 0.0, 0.0], (n_h, 1)))
    scene = pkg.scenes.Scene(cat(hs.pos, sp_pos), cat(hs.quat, np.tile([1.0, 0, 0, 0], (n_h, 1))), np.zeros((n, 3)), np.zeros((n, 3)),
                             cat(hs.mass, np.full((n_h, 1), 0.5)), cat(hs.inertia, np.full((n_h, 3), 0.5 * 0.4 * 0.04)), sides,
                             cat(hs.gtype, np.full(n_h, pkg.scenes.GEOM_SPHERE, np.uint8)), hs.plane, hs.hull_points, hs.hull_planes).astype("float32")
    steps = 120
    ow = _oracle_for("float32", scene, n_h)
    hs_contacts = 0
    for _ in range(steps):
        ow.tick(H)
        hs_contacts += _pair_kinds(ow, n_h)[1]
    assert hs_contacts > 50, hs_contacts
    w = pkg.BatchWorld(n, dtype="float32")
    w.load_scene(scene)
    w.step(H, steps)
    _same(w, ow)
    w.close()


def test_hulls_whose_collide_bits_name_the_map_only():
    """BASELINE configs[4] names box-trimesh contacts: teapots that collide with the floor and not with one another -- in ODE,
    geoms whose collide bits do not name each other's category (dGeomSetCollideBits, main.c:725, sets them per geom).  The batch's
    form: dmxBatchSetClassPairs(CONVEX, CONVEX, off).  Same crowded scene as above: the oracle with collide bits = CMASK_MAP on
    every hull geom, the batch with the class pair off -- identical, and no tick needs the pair search."""
    hull = _teapot()
    scene = pkg.scenes.hull_grid(hull, 6, 6, seed=11, y_range=(0.7, 1.4), spin=True, tilt=0.6, floor_box=True).astype("float64")
    scene.pos[:, 0] *= 0.62; scene.pos[:, 2] *= 0.62
    steps = 150
    ow = _oracle_for("float64", scene, scene.n)
    n_static = len(scene.static_boxes)
    for g in range(n_static, n_static + scene.n):           # geoms in creation order: the static boxes, then one per body
        ow.lib.orc_geom_set_collide_bits(ow.w, g, 1)         # CMASK_MAP (inc/body.h:8-12)
    ow.run(H, steps)
    assert all(j[0] < 0 or j[1] < 0 for j in ow.joints()) and ow.n_contacts() > scene.n
    w = pkg.BatchWorld(scene.n, dtype="float64")
    w.load_scene(scene)
    w.set_class_pairs(pkg.scenes.GEOM_CONVEX, pkg.scenes.GEOM_CONVEX, False)
    w.step(H, steps)
    _same(w, ow)
    st = w.collision_stats()
    assert st["careful_ticks"] == 0 and st["fast_ticks"] == steps, st
    w.close()


@pytest.mark.parametrize("dtype,plane", [("float64", False), ("float32", False), ("float32", True)])
def test_teapots_on_a_floor_strewn_with_small_blocks(dtype, plane):
    """The hull-against-map collider with a 64-body tile per workgroup (np_convex_static_tile): 9 x 8 = 72 tilted, spinning teapots (two
    wavefronts, the second one part empty) fall on a floor -- a static box, or the ground plane -- strewn with turned blocks
    0.3 m across: the hull's points inside the floor and inside the blocks (array order, the first max_contacts kept), and block
    corners inside the hull (the wavefront's walk over the hull's faces, entered from single lanes).  Collisions between the
    teapots are off so that every contact is the map's.  Bit-identical to the oracle."""
    hull = _teapot()
    scene = pkg.scenes.hull_grid(hull, 9, 8, seed=23, y_range=(0.9, 1.6), spin=True, tilt=0.5, floor_box=not plane, plane=plane).astype(dtype)
    rng = np.random.default_rng(12)
    statics = list(scene.static_boxes or [])
    for k in range(scene.n):
        if k % 3 == 2:
            continue
        yaw = rng.uniform(0, np.pi)
        c, s = np.cos(yaw), np.sin(yaw)
        R12 = np.array([c, 0.0, s, 0.0, 0.0, 1.0, 0.0, 0.0, -s, 0.0, c, 0.0])
        statics.append(((0.3, 0.3, 0.3), (scene.pos[k, 0] + rng.uniform(-0.5, 0.5), 0.15, scene.pos[k, 2] + rng.uniform(-0.5, 0.5)), R12))
    scene.static_boxes = statics
    steps = 240
    ow = _oracle_for(dtype, scene, scene.n)
    n_static = len(statics)
    first = n_static + (1 if plane else 0)                  # geoms in creation order: the plane, the statics, one per body
    for g in range(first, first + scene.n):
        ow.lib.orc_geom_set_collide_bits(ow.w, g, 1)
    most = 0
    for _ in range(steps):
        ow.tick(H)
        most = max(most, ow.n_contacts())
    assert most > 3 * scene.n, most
    w = pkg.BatchWorld(scene.n, dtype=dtype)
    w.load_scene(scene)
    w.set_class_pairs(pkg.scenes.GEOM_CONVEX, pkg.scenes.GEOM_CONVEX, False)
    w.step(H, steps)
    _same(w, ow)
    w.close()


@pytest.mark.parametrize("seed", [1179, 1235, 1527, 1581, 2091])
def test_hulls_on_blocks_kilometres_from_the_origin(seed):
    """Scenes of scripts/fuzz_hulls_r04.py that failed before the fused hull path confirmed its pairs by dSpaceCollide's test on
    the hull's EXACT box: f32, the whole scene 2.5-8 km from the origin, where positions round to 0.2-0.5 mm -- a hull vertex within
    rounding of a block's face is inside the block by the collider's arithmetic while the two AABBs, rounded their own way, miss
    each other by an ulp; the oracle (ODE's order: AABBs first) then makes no contact, and neither may the device."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_hulls_r04", os.path.join(ROOT, "scripts", "fuzz_hulls_r04.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    ok, most, n = fz.one(seed, _teapot())
    assert ok and most > 0
