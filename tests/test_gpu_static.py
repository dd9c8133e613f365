"""GPU parity of the fused path for bodies at static geometry (np_static -> step_contacts): the reference's floor IS a
static box (/root/reference/src/main.c:115, AddBodyMap main.c:735-761), so a body resting on it is a one-body dynamics
island whose contacts are dCollide(static geom, body geom).  Everything here is held against the CPU oracle bit for bit,
and the fused path against the exact tick it replaces (dmxBatchSetStaticPath)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from __graft_entry__ import ROOT, load_package

pkg = load_package()
pytestmark = pytest.mark.gpu

H = 1.0 / 60.0


def _orc(dtype):
    from oracle.orc_ctypes import Oracle
    return Oracle(dtype)


def _oracle(dtype, scene, statics):
    ow = _orc(dtype).world()
    if scene.plane is not None:
        ow.add_plane(*scene.plane)
    if scene.hull_points is not None:
        ow.set_hull(scene.hull_points)
        ow.set_hull_faces(scene.hull_planes)
    for sides, pos, R12 in statics:
        ow.add_static_box(sides, pos, R12)
    if (scene.gtype == pkg.scenes.GEOM_CONVEX).all():
        ow.add_convex(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia)
    else:
        ow.add_boxes(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia, scene.sides)
    return ow


def _world(dtype, scene, statics, fused=True):
    w = pkg.BatchWorld(scene.n, dtype=dtype)
    w.load_scene(scene)
    w.set_static_boxes(statics)
    w.set_static_path(fused)
    return w


def _same(got, ref):
    for name, a, b in zip(("pos", "quat", "lvel", "avel"), got, ref):
        assert np.all(np.isfinite(a)), name
        assert np.array_equal(a, b), f"{name}: max abs diff {np.max(np.abs(a - b))}"


def _contacts_per_body(ow, n):
    c = np.zeros(n, int)
    for j in ow.joints():
        for b in j[:2]:
            if b >= 0:
                c[b] += 1
    return c


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_and_exact_static_paths_agree_with_the_oracle(dtype):
    """boxes dropped on a tilted plank over a ground plane: plane contacts first, then the plank's, in one island per
    body -- the fused path, the exact tick and the oracle give the same bits; the fused path takes fast ticks only"""
    scene = pkg.scenes.box_grid(14, 10, seed=21, y_range=(2.0, 3.0), spin=False, box_mass=True, plane=True).astype(dtype)
    plank = [((40.0, 0.5, 5.0), (0.0, 0.45, 0.0), pkg.scenes._rot_z(0.02))]      # under the two middle rows; the others land on the plane
    steps = 110
    ow = _oracle(dtype, scene, plank)
    seen = 0
    for _ in range(steps):
        ow.tick(H)
        seen = max(seen, int(_contacts_per_body(ow, scene.n).max()))
        assert all(j[0] < 0 or j[1] < 0 for j in ow.joints())           # static contacts only: every island is one body
    assert seen == 4 and ow.state()[0][:, 1].max() > 0.7 and ow.state()[0][:, 1].min() < 0.6      # some rest on the plank, some on the plane
    states = {}
    for fused in (True, False):
        w = _world(dtype, scene, plank, fused)
        w.step(H, steps)
        states[fused] = w.state()
        st = w.collision_stats()
        if fused:
            assert st["careful_ticks"] == 0 and st["fast_ticks"] == steps, st
        else:
            assert st["careful_ticks"] > 0, st
        assert w.last_contact_count() == ow.n_contacts()
        w.close()
    _same(states[True], ow.state())
    _same(states[False], ow.state())


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_boxes_across_a_narrow_plank_have_five_to_eight_contacts(dtype):
    """a box lying diagonally across a plank narrower than itself: the clipped face is a hexagon / octagon -- more than
    four contacts from ONE geom pair, the second step_contacts launch (rows of 5..8 contacts in registers) steps it"""
    n = 24
    rng = np.random.default_rng(5)
    pos = np.stack([np.arange(n) * 3.0 - 1.5 * n, np.full(n, 0.62), np.zeros(n)], axis=1)
    yaw = rng.uniform(0.5, 1.0, n)
    quat = np.stack([np.cos(yaw / 2), np.zeros(n), np.sin(yaw / 2), np.zeros(n)], axis=1)
    sides = np.tile([1.6, 0.4, 1.6], (n, 1))
    mass = np.full((n, 1), 1.6 * 0.4 * 1.6)
    inertia = np.stack([mass[:, 0] / 12 * (0.4 ** 2 + 1.6 ** 2), mass[:, 0] / 12 * (1.6 ** 2 + 1.6 ** 2), mass[:, 0] / 12 * (0.4 ** 2 + 1.6 ** 2)], axis=1)
    scene = pkg.scenes.Scene(pos, quat, np.zeros((n, 3)), np.zeros((n, 3)), mass, inertia, sides, np.full(n, pkg.scenes.GEOM_BOX, np.uint8), None).astype(dtype)
    plank = [((200.0, 0.8, 0.9), (0.0, 0.0, 0.0), pkg.scenes._rot_z(0.0))]
    steps = 90
    ow = _oracle(dtype, scene, plank)
    most = 0
    for _ in range(steps):
        ow.tick(H)
        most = max(most, int(_contacts_per_body(ow, n).max()))
    assert 5 <= most <= 8, most
    w = _world(dtype, scene, plank)
    w.step(H, steps)
    _same(w.state(), ow.state())
    st = w.collision_stats()
    assert st["careful_ticks"] == 0 and st["fast_ticks"] == steps, st          # the chunk that met such a body first ran again, fused
    assert w.last_contact_count() == ow.n_contacts()
    w.close()


def test_box_wedged_in_a_corner_overflows_the_fused_buffer():
    """floor + two walls: a box pushed into the corner touches three static boxes -- twelve contacts, more than the fused
    path's buffer of eight: its chunks go the exact way (and back off), results stay the oracle's"""
    statics = [((20.0, 1.0, 20.0), (0.0, -0.5, 0.0), pkg.scenes._rot_z(0.0)),
               ((1.0, 6.0, 20.0), (-3.0, 3.0, 0.0), pkg.scenes._rot_z(0.0)),
               ((20.0, 6.0, 1.0), (0.0, 3.0, -3.0), pkg.scenes._rot_z(0.0))]
    scene = pkg.scenes.box_grid(3, 3, seed=2, y_range=(0.6, 0.9), spin=False, plane=False).astype("float64")
    scene.sides[:] = 0.8
    scene.pos[0] = (-2.1 - 0.005, 0.4 - 0.002, -2.1 - 0.005)          # flush into the corner, a hair inside all three
    scene.pos[1:, 0] += 4.0
    scene.pos[1:, 2] += 4.0
    steps = 150
    ow = _oracle("float64", scene, statics)
    most = 0
    for _ in range(steps):
        ow.tick(H)
        most = max(most, int(_contacts_per_body(ow, scene.n).max()))
    assert most > 8, most
    w = _world("float64", scene, statics)
    w.step(H, steps)
    _same(w.state(), ow.state())
    st = w.collision_stats()
    assert st["careful_ticks"] > 0 and st["careful_ticks"] + st["fast_ticks"] == steps, st
    w.close()


def test_pen_with_piles_mixes_fused_static_bodies_and_islands():
    """the reference's pen with bodies piling up in the middle: in an exact tick the bodies in pairs (or over two static
    boxes) go through the islands, everyone else at the floor through the fused path with those masked out"""
    spawn = pkg.scenes.reference_spawn(140, seed=17, y_range=(1.0, 9.0))
    boxes_only = [s for s in spawn if s[0] == pkg.scenes.GEOM_BOX]
    n = len(boxes_only)
    sc = pkg.scenes.Scene(np.array([s[2] for s in boxes_only], float), np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros((n, 3)), np.zeros((n, 3)),
                          np.ones((n, 1)), np.ones((n, 3)), np.array([s[1] for s in boxes_only], float),
                          np.full(n, pkg.scenes.GEOM_BOX, np.uint8), None).astype("float32")
    statics = pkg.scenes.reference_map()
    steps = 200
    ow = _oracle("float32", sc, statics)
    pair_ticks = 0
    for _ in range(steps):
        ow.tick(H)
        pair_ticks += any(j[0] >= 0 and j[1] >= 0 for j in ow.joints())
    assert pair_ticks > 20
    for fused in (True, False):
        w = _world("float32", sc, statics, fused)
        w.step(H, steps)
        _same(w.state(), ow.state())
        assert w.collision_stats()["pair_ticks"] > 0
        w.close()


def test_static_path_with_the_collision_proof_switched_off():
    """dmxBatchSetBodyCollisions(b, 0): no zones, no chunks -- the fused static path still makes the floor's contacts"""
    scene = pkg.scenes.box_grid(16, 16, seed=3, y_range=(1.1, 1.8), spin=False, box_mass=True, plane=False).astype("float32")
    floor = [((100.0, 1.0, 100.0), (0.0, 0.0, 0.0), pkg.scenes._rot_z(0.0))]
    ow = _oracle("float32", scene, floor)
    ow.run(H, 120)
    assert ow.n_contacts() > scene.n
    w = pkg.BatchWorld(scene.n, dtype="float32")
    w.load_scene(scene)
    w.set_static_boxes(floor)
    w.set_body_collisions(False)
    w.step(H, 120)
    _same(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts()
    w.close()


def test_batches_created_stepped_and_destroyed_in_a_loop_never_read_a_stale_record():
    """The small-scene exact tick's counts come back through a pinned host record the device writes; a recycled pinned page
    must never be taken for a fresh record (the record is zeroed at allocation and its sequence numbers are
    process-wide).  Forty short-lived batches, each taking exact ticks with body pairs, each the oracle's bits."""
    scene = pkg.scenes.box_grid(6, 6, seed=12, y_range=(0.5, 0.7), spin=False, plane=True).astype("float64")
    scene.pos[:, 0] *= 0.3
    scene.pos[:, 2] *= 0.3                                   # 0.75 m pitch: neighbours' AABBs overlap from the start
    ow = _oracle("float64", scene, [])
    ow.run(H, 3)
    assert ow.n_body_pairs() > 0
    ref = ow.state()
    for k in range(40):
        w = pkg.BatchWorld(scene.n, dtype="float64")
        w.set_exact_pipeline(2)
        w.load_scene(scene)
        w.step(H, 1 + k % 3)
        if k % 3 == 2:
            _same(w.state(), ref)
        w.close()


def test_host_record_wait_through_the_stream():
    """DMX_RECORD_SPIN=0: the small-scene exact tick waits on the stream instead of watching the record's sequence number.
    The switch is read once per process, so this runs one child process (configs[0]'s shape: boxes meeting on the plane)."""
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from __graft_entry__ import load_package\n"
        "from oracle.orc_ctypes import Oracle\n"
        "pkg = load_package()\n"
        "scene = pkg.scenes.box_grid(8, 8, seed=5, y_range=(0.5, 1.5), spin=True, box_mass=True, plane=True).astype('float32')\n"
        "scene.pos[:, 0] *= 0.35; scene.pos[:, 2] *= 0.35\n"
        "w = pkg.BatchWorld(scene.n, dtype='float32'); w.load_scene(scene); w.step(1 / 60, 120)\n"
        "ow = Oracle('float32').world(); ow.add_plane(*scene.plane)\n"
        "ow.add_boxes(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia, scene.sides)\n"
        "pairs = 0\n"
        "for _ in range(120):\n"
        "    ow.tick(1 / 60); pairs += ow.n_body_pairs()\n"
        "assert pairs > 0 and w.collision_stats()['careful_ticks'] > 0\n"
        "for a, b in zip(w.state(), ow.state()):\n"
        "    assert np.array_equal(a, b)\n"
        "print('record-wait-ok')\n")
    env = dict(os.environ, DMX_RECORD_SPIN="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "record-wait-ok" in out.stdout, out.stderr[-2000:]
