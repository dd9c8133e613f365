"""Analytic known-answer tests that pin the CPU oracle.

The reference ships no tests and ODE is not vendored (parity unpinned), so the
oracle is pinned by closed forms that follow from the integrator / contact
definitions, plus the golden values of the one compilable reference slice
(src/rand.c, tests/golden/rand_golden.json).
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from oracle import orc_ctypes as oc

H = 1.0 / 60.0
G = -9.8


def _one_box(orc, pos=(0, 10, 0), quat=(1, 0, 0, 0), lvel=(0, 0, 0), avel=(0, 0, 0),
             mass=None, idiag=None, sides=(1, 1, 1), gravity=(0, G, 0)):
    w = orc.world(gravity=gravity)
    w.add_boxes([pos], [quat], [lvel], [avel],
                None if mass is None else [mass],
                None if idiag is None else [idiag], [sides])
    return w


# ---------------------------------------------------------------- golden: rand.c
def test_rand_golden(orc64):
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rand_golden.json")))
    lib = orc64.lib
    for case in g["next"]:
        lib.orc_ref_rand_seed(case["seed"])
        assert [lib.orc_ref_rand_next() for _ in case["values"]] == case["values"]
    lib.orc_ref_rand_seed(g["mixed"]["seed"])
    for call in g["mixed"]["calls"]:
        if call["fn"] == "double":
            assert lib.orc_ref_rand_double(call["min"], call["max"]) == call["value"]
        else:
            assert lib.orc_ref_rand_int(call["min"], call["max"]) == call["value"]


# ---------------------------------------------------------------- KAT-1 free fall
@pytest.mark.parametrize("prec", ["float64", "float32"])
def test_kat1_free_fall_symplectic_euler(prec):
    orc = oc.Oracle(prec)
    w = _one_box(orc, pos=(1.0, 100.0, -2.0))
    n = 60
    w.run(H, n)
    pos, quat, lvel, avel = w.state()
    # velocity first, then position with the NEW velocity
    v = n * H * G
    dy = G * H * H * n * (n + 1) / 2.0
    assert abs(dy - (-4.981666666666667)) < 1e-12
    tol = 1e-12 if prec == "float64" else 2e-6
    assert abs(lvel[0, 1] - v) <= tol * abs(v)
    assert abs(pos[0, 1] - (100.0 + dy)) <= tol * 100.0
    assert pos[0, 0] == 1.0 and pos[0, 2] == -2.0
    assert np.all(quat[0] == np.array([1, 0, 0, 0], quat.dtype))


def test_kat1_initial_velocity(orc64):
    w = _one_box(orc64, pos=(0, 0, 0), lvel=(3.0, 5.0, -1.0))
    n = 17
    w.run(H, n)
    pos, _, lvel, _ = w.state()
    assert abs(lvel[0, 1] - (5.0 + n * H * G)) < 1e-12
    assert abs(pos[0, 1] - (n * H * 5.0 + G * H * H * n * (n + 1) / 2.0)) < 1e-12
    assert abs(pos[0, 0] - 3.0 * n * H) < 1e-12
    assert abs(pos[0, 2] + 1.0 * n * H) < 1e-12


# ---------------------------------------------------------------- KAT-2 quaternion update
@pytest.mark.parametrize("axis", [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 2, -2)])
def test_kat2_spin_angle_is_2atan(orc64, axis):
    a = np.array(axis, float)
    a /= np.linalg.norm(a)
    wmag = 3.0
    w = _one_box(orc64, avel=tuple(wmag * a), gravity=(0, 0, 0))
    n = 25
    w.run(H, n)
    _, quat, _, avel = w.state()
    theta = 2.0 * n * math.atan(wmag * H / 2.0)     # NOT n*|w|*h
    expect = np.concatenate([[math.cos(theta / 2)], math.sin(theta / 2) * a])
    assert np.allclose(quat[0], expect, atol=1e-13)
    assert abs(np.linalg.norm(quat[0]) - 1.0) < 1e-15
    # isotropic default inertia: the gyroscopic torque vanishes to rounding
    assert np.allclose(avel[0], wmag * a, rtol=0, atol=1e-13)


def test_kat2_quaternion_order_wxyz(orc64):
    # 90 deg about z given as (w,x,y,z) -> R maps x to y
    s = math.sqrt(0.5)
    w = _one_box(orc64, quat=(s, 0, 0, s), gravity=(0, 0, 0))
    Rm = np.ctypeslib.as_array(orc64.lib.orc_body_get_rotation(w.w, 0), shape=(12,)).reshape(3, 4)
    assert np.allclose(Rm[:, :3] @ np.array([1.0, 0, 0]), [0, 1, 0], atol=1e-15)
    assert np.all(Rm[:, 3] == 0)


# ---------------------------------------------------------------- KAT-3 gyroscopic term
def _cross_mat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def test_kat3_gyro_explicit(orc64):
    I = np.array([1.0, 2.0, 3.5])
    w0 = np.array([0.7, -1.1, 0.4])
    w = _one_box(orc64, avel=tuple(w0), mass=2.0, idiag=tuple(I), gravity=(0, 0, 0))
    orc64.lib.orc_world_set_gyro_mode(w.w, oc.GYRO_EXPLICIT)
    w.run(H, 1)
    _, _, _, avel = w.state()
    expect = w0 + H * (-np.cross(w0, I * w0)) / I
    assert np.allclose(avel[0], expect, rtol=0, atol=1e-14)


def test_kat3_gyro_implicit_lacoursiere(orc64):
    I = np.array([1.0, 2.0, 3.5])
    w0 = np.array([0.7, -1.1, 0.4])
    w = _one_box(orc64, avel=tuple(w0), mass=2.0, idiag=tuple(I), gravity=(0, 0, 0))
    w.run(H, 1)     # default mode = implicit
    _, _, _, avel = w.state()
    L = I * w0
    expect = np.linalg.solve(np.diag(I) - H * _cross_mat(L), L)   # (I - h[L]x) w' = L
    assert np.allclose(avel[0], expect, rtol=0, atol=1e-13)
    # and it differs from the explicit form at O(h^2)
    assert np.linalg.norm(expect - (w0 + H * (-np.cross(w0, I * w0)) / I)) > 1e-6


def test_kat3_rotated_body_uses_world_inertia(orc64):
    # rotate the body 90 deg about z: world inertia swaps Ixx and Iyy
    s = math.sqrt(0.5)
    I = np.array([1.0, 2.0, 3.5])
    w0 = np.array([0.7, -1.1, 0.4])
    w = _one_box(orc64, quat=(s, 0, 0, s), avel=tuple(w0), mass=1.0, idiag=tuple(I),
                 gravity=(0, 0, 0))
    orc64.lib.orc_world_set_gyro_mode(w.w, oc.GYRO_EXPLICIT)
    w.run(H, 1)
    _, _, _, avel = w.state()
    Iw = np.array([2.0, 1.0, 3.5])
    expect = w0 + H * (-np.cross(w0, Iw * w0)) / Iw
    assert np.allclose(avel[0], expect, rtol=0, atol=1e-13)


# ---------------------------------------------------------------- KAT-4 contact row rhs / bounce
def _sphere_on_plane(orc, depth, vy, radius=0.5, gravity=(0, 0, 0)):
    w = orc.world(gravity=gravity)
    w.add_plane(0, 1, 0, 0)
    w.add_spheres([(0.0, radius - depth, 0.0)], None, [(0.0, vy, 0.0)], None, None, None, [radius])
    return w


def test_kat4_resting_penetration_pushout(orc64):
    d = 0.01
    w = _sphere_on_plane(orc64, d, 0.0)
    w.tick(H)
    assert w.n_contacts() == 1
    _, _, lvel, avel = w.state()
    # bias dominates: post-step normal velocity = erp*d/h (CFM 1e-10 negligible)
    assert abs(lvel[0, 1] - 0.2 * d / H) < 1e-8
    assert np.allclose(lvel[0, [0, 2]], 0, atol=1e-14) and np.allclose(avel[0], 0, atol=1e-14)
    assert w.sor_residual() < 1e-8        # |1-1.3|^19 * 1.3 * lambda: geometric contraction on a 1-row problem


def test_kat4_bounce_rule(orc64):
    d = 0.001
    # incoming 2 m/s > bounce_vel 0.1: outgoing = max(erp*d/h, 0.2*2) = 0.4   (main.c:684-686)
    w = _sphere_on_plane(orc64, d, -2.0)
    w.tick(H)
    _, _, lvel, _ = w.state()
    # exact row equation: (1 + cfm/h) * lambda = (c + 2)/h with c = 0.2*2, v' = -2 + h*lambda
    assert abs(lvel[0, 1] - (-2.0 + 2.4 / (1.0 + 1e-10 / H))) < 1e-10
    assert abs(lvel[0, 1] - 0.4) < 1e-7
    # incoming 0.05 m/s < bounce_vel: no bounce, only the ERP push-out
    w = _sphere_on_plane(orc64, d, -0.05)
    w.tick(H)
    _, _, lvel, _ = w.state()
    assert abs(lvel[0, 1] - 0.2 * d / H) < 1e-8


def test_kat4_gravity_is_cancelled_by_contact(orc64):
    d = 0.0
    w = _sphere_on_plane(orc64, d, 0.0, gravity=(0, G, 0))
    w.tick(H)
    _, _, lvel, _ = w.state()
    assert abs(lvel[0, 1]) < 1e-8          # v' = erp*0/h = 0, gravity absorbed by lambda


def test_kat4_separated_no_contact(orc64):
    w = _sphere_on_plane(orc64, -0.25, 0.0, gravity=(0, G, 0))
    w.tick(H)
    assert w.n_contacts() == 0
    _, _, lvel, _ = w.state()
    assert abs(lvel[0, 1] - H * G) < 1e-15


# ---------------------------------------------------------------- KAT-5 box-plane contact set
def _box_plane_contacts(orc, pos, quat, sides, maxc=8):
    w = orc.world()
    pl = w.add_plane(0, 1, 0, 0)
    w.add_boxes([pos], [quat], None, None, None, None, [sides])
    cg = (orc.ContactGeom * 16)()
    n = orc.lib.orc_collide(w.w, 1, pl, maxc, cg)
    return [(np.array(cg[i].pos[:]), np.array(cg[i].normal[:]), cg[i].depth) for i in range(n)]


def test_kat5_axis_aligned_box_four_corners(orc64):
    c = _box_plane_contacts(orc64, (0.0, 0.45, 0.0), (1, 0, 0, 0), (1.0, 1.0, 2.0))
    assert len(c) == 4
    for p, n, d in c:
        assert np.all(n == [0, 1, 0])
        assert abs(d - 0.05) < 1e-15
        assert abs(p[1] + 0.05) < 1e-15          # the box corner, below the plane
        assert abs(abs(p[0]) - 0.5) < 1e-15 and abs(abs(p[2]) - 1.0) < 1e-15
    assert len({(round(p[0], 6), round(p[2], 6)) for p, _, _ in c}) == 4


def test_kat5_tilted_box_edge_contact(orc64):
    # 45 deg about z: the lowest feature is an edge along z -> two equally deep contacts
    a = math.pi / 4
    q = (math.cos(a / 2), 0, 0, math.sin(a / 2))
    half_diag = math.sqrt(0.5)
    c = _box_plane_contacts(orc64, (0.0, half_diag - 0.02, 0.0), q, (1.0, 1.0, 1.0))
    assert len(c) == 2
    assert abs(c[0][2] - 0.02) < 1e-12 and abs(c[1][2] - 0.02) < 1e-12
    assert abs(c[0][0][2] - c[1][0][2]) == pytest.approx(1.0, abs=1e-12)   # one side apart along z
    assert c[0][2] >= c[1][2]                                               # deepest first


def test_kat5_corner_contact_and_limits(orc64):
    # generic tilt: one corner deepest, depths non-increasing from the first
    q = np.array([0.9, 0.3, 0.2, 0.1])
    q /= np.linalg.norm(q)
    c = _box_plane_contacts(orc64, (0.0, 0.6, 0.0), tuple(q), (1.0, 0.8, 0.6))
    assert 1 <= len(c) <= 4
    assert all(c[0][2] >= ci[2] for ci in c[1:])
    # depth of each contact = -(n . p) for the plane y = 0
    for p, n, d in c:
        assert abs(d + p[1]) < 1e-12
    # maxc = 1 keeps only the deepest
    c1 = _box_plane_contacts(orc64, (0.0, 0.6, 0.0), tuple(q), (1.0, 0.8, 0.6), maxc=1)
    assert len(c1) == 1 and c1[0][2] == c[0][2]
    # above the plane: nothing
    assert _box_plane_contacts(orc64, (0.0, 5.0, 0.0), tuple(q), (1.0, 0.8, 0.6)) == []


def test_reversed_pair_flips_normal(orc64):
    w = orc64.world()
    pl = w.add_plane(0, 1, 0, 0)
    w.add_spheres([(0.0, 0.4, 0.0)], None, None, None, None, None, [0.5])
    cg = (orc64.ContactGeom * 4)()
    assert orc64.lib.orc_collide(w.w, 1, pl, 8, cg) == 1
    assert cg[0].normal[1] == 1.0 and (cg[0].g1, cg[0].g2) == (1, pl)
    assert orc64.lib.orc_collide(w.w, pl, 1, 8, cg) == 1
    assert cg[0].normal[1] == -1.0 and (cg[0].g1, cg[0].g2) == (pl, 1)


# ---------------------------------------------------------------- resting box, settles
def test_box_settles_on_plane(orc64):
    w = orc64.world()
    w.add_plane(0, 1, 0, 0)
    w.add_boxes([(0.0, 1.0, 0.0)], None, None, None, None, None, [(1.0, 0.5, 0.8)])
    w.run(H, 600)
    pos, quat, lvel, avel = w.state()
    assert w.n_contacts() == 4
    assert abs(pos[0, 1] - 0.25) < 2e-3            # resting height = half side (small ERP sag)
    assert np.linalg.norm(lvel[0]) < 1e-3 and np.linalg.norm(avel[0]) < 1e-3
    assert abs(np.linalg.norm(quat[0]) - 1.0) < 1e-14


# ---------------------------------------------------------------- pose packing
def test_pack_transform_matches_reference_layout(orc64):
    # GetTransformMat, main.c:602-622
    R = np.arange(12, dtype=float) + 1.0
    p = np.array([7.0, 8.0, 9.0])
    out = np.zeros(16)
    RP = C.POINTER(C.c_double)
    orc64.lib.orc_pack_transform(out.ctypes.data_as(RP), p.ctypes.data_as(RP), R.ctypes.data_as(RP))
    expect = [R[0], R[4], R[8], 0, R[1], R[5], R[9], 0, R[2], R[6], R[10], 0, 7, 8, 9, 1]
    assert list(out) == expect


# ---------------------------------------------------------------- row-order modes agree on 1-body islands to solver tolerance
def test_ode_row_order_mode_close_to_fixed(orc64):
    def run(mode):
        w = orc64.world()
        orc64.lib.orc_world_set_row_order(w.w, mode)
        orc64.lib.orc_rand_seed(0)
        w.add_plane(0, 1, 0, 0)
        w.add_boxes([(0.0, 0.6, 0.0)], [(0.99, 0.1, 0.05, 0.02)], None, None, None, None,
                    [(1.0, 0.5, 0.8)])
        w.run(H, 240)
        return w.state()
    a, b = run(oc.ORDER_FIXED), run(oc.ORDER_ODE)
    assert np.allclose(a[0], b[0], atol=5e-3)


# ---------------------------------------------------------------- oracle broadphase variants enumerate the same pairs
def test_grid_and_sweep_broadphase_agree(orc64):
    from __graft_entry__ import load_package
    pkg = load_package()
    s = pkg.scenes.box_grid(24, 24, seed=11, y_range=(0.8, 4.0), spin=True, box_mass=True).astype("float64")
    s.avel *= 3.0

    def run(mode):
        w = orc64.world()
        orc64.lib.orc_world_set_broadphase(w.w, mode)
        w.add_plane(*s.plane)
        w.add_boxes(s.pos, s.quat, s.lvel, s.avel, s.mass[:, 0], s.inertia, s.sides)
        pairs = 0
        for _ in range(300):
            w.tick(H)
            pairs += w.n_body_pairs()
        return w.state(), pairs
    (a, pa), (b, pb) = run(1), run(2)
    assert pa == pb > 0
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


# ---------------------------------------------------------------- body-body colliders (closed forms)
def _pair_contacts(orc, w, g1, g2, maxc=8):
    cg = (orc.ContactGeom * 16)()
    n = orc.lib.orc_collide(w.w, g1, g2, maxc, cg)
    return [(np.array(cg[i].pos[:]), np.array(cg[i].normal[:]), cg[i].depth) for i in range(n)]


def test_kat6_sphere_sphere(orc64):
    # centres 0.7 apart along a slanted axis, radii 0.5 and 0.3: one contact, depth 0.1, normal from sphere 2 to sphere 1
    axis = np.array([2.0, 1.0, -2.0]) / 3.0
    p1, p2 = np.array([1.0, 2.0, 3.0]), np.array([1.0, 2.0, 3.0]) - 0.7 * axis
    w = orc64.world()
    w.add_spheres([p1, p2], None, None, None, None, None, [0.5, 0.3])
    c = _pair_contacts(orc64, w, 0, 1)
    assert len(c) == 1
    p, n, d = c[0]
    assert np.allclose(n, axis, atol=1e-15) and abs(d - 0.1) < 1e-15
    # the contact point lies on the line of centres, inside the overlap lens
    t = np.dot(p - p2, axis)
    assert np.allclose(p, p2 + t * axis, atol=1e-15) and 0.2 - 1e-12 <= t <= 0.3 + 1e-12
    # apart: nothing; swapped: the normal flips
    w2 = orc64.world()
    w2.add_spheres([p1, p1 - 0.81 * axis], None, None, None, None, None, [0.5, 0.3])
    assert _pair_contacts(orc64, w2, 0, 1) == []
    assert np.allclose(_pair_contacts(orc64, w, 1, 0)[0][1], -axis, atol=1e-15)


def test_kat6_sphere_on_a_box_face(orc64):
    # sphere r = 0.4 whose centre sits 0.35 above the top face of a 2 x 1 x 2 box: depth 0.05, normal +y (into the sphere)
    w = orc64.world()
    w.add_spheres([(0.3, 0.85, -0.2)], None, None, None, None, None, [0.4])
    w.add_boxes([(0.0, 0.0, 0.0)], None, None, None, None, None, [(2.0, 1.0, 2.0)])
    c = _pair_contacts(orc64, w, 0, 1)
    assert len(c) == 1
    p, n, d = c[0]
    assert np.allclose(n, [0, 1, 0], atol=1e-15) and abs(d - 0.05) < 1e-15
    assert np.allclose(p[[0, 2]], [0.3, -0.2], atol=1e-15) and 0.45 - 1e-12 <= p[1] <= 0.5 + 1e-12
    # beyond reach: nothing
    w2 = orc64.world()
    w2.add_spheres([(0.3, 0.95, -0.2)], None, None, None, None, None, [0.4])
    w2.add_boxes([(0.0, 0.0, 0.0)], None, None, None, None, None, [(2.0, 1.0, 2.0)])
    assert _pair_contacts(orc64, w2, 0, 1) == []


def test_kat6_box_box_face_contact(orc64):
    # two unit cubes, the second shifted 0.9 along x and a little in y, z: the faces overlap by 0.1 -> four contacts at
    # the corners of the common face region, each 0.1 deep, normal along x pointing into box 1
    w = orc64.world()
    w.add_boxes([(0.0, 0.0, 0.0), (0.9, 0.2, -0.1)], None, None, None, None, None, [(1.0, 1.0, 1.0), (1.0, 1.0, 1.0)])
    c = _pair_contacts(orc64, w, 0, 1)
    assert len(c) == 4
    for p, n, d in c:
        assert np.allclose(n, [-1, 0, 0], atol=1e-15) and abs(d - 0.1) < 1e-12
    yz = sorted((round(float(p[1]), 9), round(float(p[2]), 9)) for p, _, _ in c)
    assert yz == [(-0.3, -0.5), (-0.3, 0.4), (0.5, -0.5), (0.5, 0.4)]       # y in [-0.3, 0.5], z in [-0.5, 0.4]
    # fewer contacts asked than found: the deepest comes first and max_contacts is honoured (ODE's cullPoints)
    c2 = _pair_contacts(orc64, w, 0, 1, maxc=2)
    assert len(c2) == 2 and all(abs(d - 0.1) < 1e-12 for _, _, d in c2)
    # separated by a hair: nothing
    w2 = orc64.world()
    w2.add_boxes([(0.0, 0.0, 0.0), (1.0001, 0.2, -0.1)], None, None, None, None, None, [(1.0, 1.0, 1.0), (1.0, 1.0, 1.0)])
    assert _pair_contacts(orc64, w2, 0, 1) == []


def test_kat6_box_box_edge_edge(orc64):
    # cube 1 axis-aligned; cube 2 turned 45 degrees about x and about z-offset so one of its edges crosses an edge of
    # cube 1: a single edge-edge contact whose normal is perpendicular to both edges
    a = math.pi / 4
    qx = np.array([math.cos(a / 2), math.sin(a / 2), 0, 0])           # cube 2's edge along x sticks down like a roof ridge
    w = orc64.world()
    # cube 1 turned 45 degrees about z: its top is a ridge along z.  Ridge heights: sqrt(0.5) each.
    qz = np.array([math.cos(a / 2), 0, 0, math.sin(a / 2)])
    gap = 2 * math.sqrt(0.5) - 0.05                                  # ridges overlap by 0.05
    w.add_boxes([(0.0, 0.0, 0.0), (0.0, gap, 0.0)], [tuple(qz), tuple(qx)], None, None, None, None,
                [(1.0, 1.0, 1.0), (1.0, 1.0, 1.0)])
    c = _pair_contacts(orc64, w, 0, 1)
    assert len(c) == 1
    p, n, d = c[0]
    # depth: the ridges' overlap, plus dBoxBox's guard against parallel edges [ODE-recall "fudge2"]: 1e-5 on every |R| entry, i.e.
    # 1e-5 x the four half-sides that enter the edge-pair axis (4 x 0.5) / |u x v| (= 1 here)
    assert np.allclose(np.abs(n), [0, 1, 0], atol=1e-12) and abs(d - (0.05 + 2.0e-5)) < 1e-12
    assert abs(p[0]) < 1e-12 and abs(p[2]) < 1e-12                   # where the two ridges cross


# ---------------------------------------------------------------- friction rows (closed forms)
def _sliding_sphere(orc, mu, vx=1.0, radius=0.5):
    w = orc.world(gravity=(0, 0, 0))
    orc.lib.orc_world_set_surface(w.w, 0, mu, 0.0, 0.0)          # no bounce
    w.add_plane(0, 1, 0, 0)
    w.add_spheres([(0.0, radius, 0.0)], None, [(vx, 0.0, 0.0)], None, None, None, [radius])   # touching, depth 0
    return w


def test_kat7_unbounded_friction_brings_the_contact_point_to_rest(orc64):
    """mu = dInfinity (the reference's surface, main.c:687): the two friction rows are equality constraints on the
    contact point's tangential velocity.  Sphere m = I = 1 (dBodyCreate's default), R = 0.5, sliding at 1 m/s: the
    impulse J = -v / (1/m + R^2/I) = -0.8 leaves v = 0.2, w_z = -0.4 (rolling without slipping), rows decoupled."""
    w = _sliding_sphere(orc64, float("inf"))
    w.tick(H)
    _, _, lvel, avel = w.state()
    assert abs(lvel[0, 0] - 0.2) < 1e-8 and abs(avel[0, 2] + 0.4) < 1e-8
    assert abs(lvel[0, 0] + avel[0, 2] * 0.5) < 1e-8                 # contact point at rest: v_x + w_z R = 0
    assert abs(lvel[0, 1]) < 1e-9 and abs(lvel[0, 2]) < 1e-12 and abs(avel[0, 0]) < 1e-12


def test_kat7_bounded_friction_clamps_at_mu(orc64):
    """finite mu without dContactApprox1: |lambda_friction| <= mu as a FORCE bound [ODE-recall contact getInfo2], so one
    tick removes at most mu*h of momentum: v = 1 - mu*h/m, w_z = -R*mu*h/I."""
    mu = 0.3
    w = _sliding_sphere(orc64, mu)
    w.tick(H)
    _, _, lvel, avel = w.state()
    assert abs(lvel[0, 0] - (1.0 - mu * H)) < 1e-12 and abs(avel[0, 2] + 0.5 * mu * H) < 1e-12
    # mu = 0: a single (normal) row, the sphere keeps sliding
    w0 = _sliding_sphere(orc64, 0.0)
    w0.tick(H)
    _, _, lvel, avel = w0.state()
    assert lvel[0, 0] == 1.0 and np.all(avel[0] == 0)


# ---------------------------------------------------------------- two dynamic bodies: equal and opposite impulses
def test_kat8_head_on_spheres_conserve_momentum_and_separate_at_the_row_velocity(orc64):
    """Two unit-mass spheres (r = 0.5) overlapping by 0.1, closing head-on at 1 m/s each, no gravity, mu = 0, bounce 0.5:
    the row's target separation speed is max(erp * depth / h, bounce * closing speed) = max(1.2, 1.0) = 1.2, split evenly;
    linear momentum is conserved to rounding; a glancing hit with friction also conserves it."""
    w = orc64.world(gravity=(0, 0, 0))
    orc64.lib.orc_world_set_surface(w.w, oc.CONTACT_BOUNCE, 0.0, 0.5, 0.1)
    w.add_spheres([(-0.45, 0.0, 0.0), (0.45, 0.0, 0.0)], None, [(1.0, 0.0, 0.0), (-1.0, 0.0, 0.0)], None, None, None, [0.5, 0.5])
    w.tick(H)
    assert w.n_contacts() == 1
    _, _, lvel, avel = w.state()
    assert abs(lvel[0, 0] + 0.6) < 1e-7 and abs(lvel[1, 0] - 0.6) < 1e-7
    assert abs(lvel[0, 0] + lvel[1, 0]) < 1e-14 and np.all(avel == 0)
    # glancing, unequal masses, infinite friction: total momentum m1 v1 + m2 v2 is what it was
    w = orc64.world(gravity=(0, 0, 0))
    m = np.array([2.0, 0.5])
    v = np.array([[0.8, 0.1, 0.0], [-1.0, 0.3, 0.2]])
    w.add_spheres([(-0.4, 0.1, 0.0), (0.4, -0.2, 0.1)], None, v, None, m, [(0.2, 0.2, 0.2), (0.05, 0.05, 0.05)], [0.5, 0.5])
    w.tick(H)
    assert w.n_contacts() == 1
    _, _, lvel, _ = w.state()
    assert np.allclose((m[:, None] * lvel).sum(axis=0), (m[:, None] * v).sum(axis=0), rtol=0, atol=1e-13)


# ----------------------------------------------------------------------------------------------------------------
# dWorldStep's exact solve (ORC_STEPPER_EXACT): known answers that 20 SOR sweeps do not reach
def _stack(orc, n, exact, erp=0.0, iters=None):
    ow = orc.world()
    ow.add_plane(0, 1, 0, 0)
    pos = np.array([[0.0, 0.5 + k, 0.0] for k in range(n)])
    quat = np.tile([1.0, 0, 0, 0], (n, 1))
    z = np.zeros((n, 3))
    ow.add_boxes(pos, quat, z, z, np.ones(n), np.ones((n, 3)), np.ones((n, 3)))
    orc.lib.orc_world_set_erp(ow.w, erp)
    if iters is not None:
        orc.lib.orc_world_set_quickstep(ow.w, iters, 1.3)
    ow.set_stepper(exact)
    return ow


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-6), ("float32", 0.05)])
def test_exact_step_carries_a_stack_with_the_analytic_normal_forces(dtype, tol):
    """three unit boxes (m = 1) stacked on the plane at rest, ERP 0: the interfaces carry 3 m g, 2 m g and m g (summed over
    each interface's contacts) and the stack does not move -- the LCP's solution, not an iteration's approximation"""
    from oracle.orc_ctypes import Oracle
    orc = Oracle(dtype)
    ow = _stack(orc, 3, exact=True)
    ow.tick(1.0 / 60.0)
    force = {}
    for b1, b2, _p, _n, _d, lam in ow.joints():
        force[(b1, b2)] = force.get((b1, b2), 0.0) + lam
    assert abs(force[(0, -1)] - 3 * 9.8) < tol and abs(force[(0, 1)] - 2 * 9.8) < tol and abs(force[(1, 2)] - 9.8) < tol
    assert np.max(np.abs(ow.state()[2])) < (1e-8 if dtype == "float64" else 1e-3)     # v stays 0 (cfm lets ~1e-9 through)
    # twenty SOR sweeps leave the stack moving by millimetres per second
    sor = _stack(orc, 3, exact=False)
    sor.tick(1.0 / 60.0)
    assert np.max(np.abs(sor.state()[2])) > 1e-3


def test_sor_converges_to_the_exact_solution():
    """the SOR's fixed point is the LCP's solution: with thousands of sweeps QuickStep's velocities approach the exact ones"""
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    ex = _stack(orc, 3, exact=True, erp=0.2)
    sor = _stack(orc, 3, exact=False, erp=0.2, iters=20000)
    # start from a slightly interpenetrating, moving state so the right-hand side is not trivial
    for w in (ex, sor):
        for k in range(3):
            orc.lib.orc_body_set_position(w.w, k, 0.0, 0.5 + k * 0.999, 0.0)
            orc.lib.orc_body_set_linear_vel(w.w, k, 0.01 * k, -0.2, 0.0)
        w.tick(1.0 / 60.0)
    for a, b in zip(ex.state(), sor.state()):
        assert np.max(np.abs(a - b)) < 1e-6


def test_exact_step_on_a_frictionless_slope():
    """mu = 0, plane tilted by theta: the box accelerates along the slope with g sin(theta), nothing moves along the normal,
    and the contacts carry m g cos(theta) between them"""
    import math
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    th, h = 0.3, 1.0 / 60.0
    n = np.array([math.sin(th), math.cos(th), 0.0])
    t = np.array([math.cos(th), -math.sin(th), 0.0])
    ow = orc.world()
    ow.add_plane(n[0], n[1], 0.0, 0.0)
    orc.lib.orc_world_set_surface(ow.w, 0, 0.0, 0.0, 0.0)
    orc.lib.orc_world_set_erp(ow.w, 0.0)
    q = np.array([[math.cos(th / 2), 0.0, 0.0, -math.sin(th / 2)]])           # the box's y axis along the plane's normal
    ow.add_boxes((0.499 * n)[None, :], q, np.zeros((1, 3)), np.zeros((1, 3)), np.ones(1), np.ones((1, 3)), np.ones((1, 3)))
    ow.set_stepper(True)
    ow.tick(h)
    v = ow.state()[2][0]
    joints = ow.joints()
    assert len(joints) == 4
    assert abs(v @ n) < 1e-8
    assert abs(v @ t - h * 9.8 * math.sin(th)) < 1e-12
    assert abs(sum(j[5] for j in joints) - 9.8 * math.cos(th)) < 1e-6
    assert ow.sor_residual() < 1e-9                                             # complementarity holds


def test_exact_step_reaches_complementarity_on_a_pile():
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    ow = orc.world()
    ow.add_plane(0, 1, 0, 0)
    rng = np.random.default_rng(3)
    nb = 12
    pos = np.stack([rng.uniform(-0.15, 0.15, nb), 0.45 + 0.85 * np.arange(nb), rng.uniform(-0.15, 0.15, nb)], axis=1)   # a column, no overlap
    quat = np.tile([1.0, 0, 0, 0], (nb, 1))
    z = np.zeros((nb, 3))
    ow.add_boxes(pos, quat, z, z, np.ones(nb), np.ones((nb, 3)), np.full((nb, 3), 0.8))
    ow.set_stepper(True)
    worst = 0.0
    most = 0
    for _ in range(150):
        ow.tick(1.0 / 120.0)
        worst = max(worst, ow.sor_residual())
        most = max(most, ow.n_contacts())
    assert most > 2 * nb and np.all(np.isfinite(ow.state()[0]))
    assert worst < 1e-5                                   # summed over the rows, against right-hand sides of order 1e2-1e3
    assert orc.lib.orc_world_last_lcp_rounds(ow.w) < 200


# ----------------------------------------------------------------------------------------------------------------
# box against convex hull: the repository's own collider (ODE's dCollideConvexBox is a stub) -- closed-form cases
def _cube_hull_world(orc, hull_half, hull_pos, box_sides, box_pos):
    """a static box and one convex body whose hull is a cube of half-extent hull_half (8 points, 6 x 2 triangles' planes)"""
    import itertools
    pts = np.array(list(itertools.product((-hull_half, hull_half), repeat=3)), float)
    planes = []
    for a in range(3):
        for sg in (-1.0, 1.0):
            n = [0.0, 0.0, 0.0]; n[a] = sg
            planes.append(n + [hull_half])
    ow = orc.world()
    ow.set_hull(pts); ow.set_hull_faces(np.array(planes))
    g_box = ow.add_static_box(box_sides, box_pos, [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0])
    ow.add_convex(np.array([hull_pos], float), np.array([[1.0, 0, 0, 0]]), np.zeros((1, 3)), np.zeros((1, 3)), np.ones(1), np.ones((1, 3)))
    return ow, g_box, g_box + 1


def _collide(orc, ow, g1, g2, maxc=8):
    out = (orc.ContactGeom * 16)()
    n = orc.lib.orc_collide(ow.w, g1, g2, maxc, out)
    return [(list(out[i].pos), list(out[i].normal), out[i].depth) for i in range(n)]


def test_box_convex_hull_vertices_inside_the_box():
    """a cube hull (half 0.5) sunk 0.01 into the top face of a big static box: its four bottom vertices are the contacts,
    each 0.01 deep along the box's top face; the normal points into the box (o1) and the flipped order flips it"""
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    ow, gb, gh = _cube_hull_world(orc, 0.5, (0.3, 0.49, -0.2), (10.0, 1.0, 10.0), (0.0, -0.5, 0.0))
    cs = _collide(orc, ow, gb, gh)
    assert len(cs) == 4
    for pos, nrm, dep in cs:
        assert abs(pos[1] + 0.01) < 1e-12 and abs(abs(pos[0] - 0.3) - 0.5) < 1e-12 and abs(abs(pos[2] + 0.2) - 0.5) < 1e-12
        assert nrm == [0.0, -1.0, 0.0] or np.allclose(nrm, [0, -1, 0])
        assert abs(dep - 0.01) < 1e-12
    flipped = _collide(orc, ow, gh, gb)
    assert len(flipped) == 4 and all(np.allclose(c[1], [0, 1, 0]) for c in flipped)
    assert len(_collide(orc, ow, gb, gh, maxc=3)) == 3                         # the first three in the hull's array order
    assert [c[0] for c in _collide(orc, ow, gb, gh, maxc=3)] == [c[0] for c in cs[:3]]


def test_box_convex_box_corner_inside_the_hull():
    """a small static box poking its top into the bottom face of a big cube hull: no hull vertex is inside the box, the box's
    four top corners are inside the hull; the contacts sit at those corners (in corner order), along the hull face they
    are nearest to"""
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    # hull: cube of half 1 centred at (0, 1.05, 0) -> bottom face at y = 0.05.  Box: sides 0.2, centre (0.5, 0, 0.5): top at y = 0.1
    ow, gb, gh = _cube_hull_world(orc, 1.0, (0.0, 1.05, 0.0), (0.2, 0.2, 0.2), (0.5, 0.0, 0.5))
    cs = _collide(orc, ow, gb, gh)
    assert [c[0] for c in cs] == [[x, 0.1, z] for z in (0.4, 0.6) for x in (0.4, 0.6)] or \
        np.allclose([c[0] for c in cs], [[x, 0.1, z] for z in (0.4, 0.6) for x in (0.4, 0.6)])      # corners 2, 3, 6, 7 (bit 1 = +y)
    for pos, nrm, dep in cs:
        assert np.allclose(nrm, [0.0, -1.0, 0.0])        # the hull's bottom face (outward normal -y) points into the box
        assert abs(dep - 0.05) < 1e-12
    # without the faces only the vertex half of the collider runs: nothing found
    ow.set_hull_faces(np.zeros((0, 4)))
    assert _collide(orc, ow, gb, gh) == []


def test_box_convex_supports_a_resting_hull():
    """one exact step of a unit-cube hull resting on a static box: the four contacts carry m g between them"""
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    ow, gb, gh = _cube_hull_world(orc, 0.5, (0.0, 0.4999, 0.0), (10.0, 1.0, 10.0), (0.0, -0.5, 0.0))
    orc.lib.orc_world_set_erp(ow.w, 0.0)
    ow.set_stepper(True)
    ow.tick(1.0 / 60.0)
    j = ow.joints()
    assert len(j) == 4 and abs(sum(x[5] for x in j) - 9.8) < 1e-6
    assert all(np.allclose(x[3], [0, 1, 0]) for x in j)                      # after the joint's reversal: into the hull
    assert np.max(np.abs(ow.state()[2])) < 1e-8


# ---------------------------------------------------------------- more closed forms for the colliders (round 2)
def test_kat6_box_box_fewer_contacts_than_found_keeps_the_deepest(orc64):
    """a cube tipped slightly onto another's top face: the clipped face region has four points of DIFFERENT depths; asking for
    1, 2 or 3 contacts keeps the deepest one first and only points of the full set (ODE's cullPoints picks around the
    centroid starting from the deepest)"""
    tilt = 0.05
    q = np.array([math.cos(tilt / 2), 0, 0, math.sin(tilt / 2)])              # a small turn about z: one bottom edge dips
    w = orc64.world()
    w.add_boxes([(0.0, 0.0, 0.0), (0.0, 0.95, 0.0)], [(1, 0, 0, 0), tuple(q)], None, None, None, None,
                [(2.0, 1.0, 2.0), (1.0, 1.0, 1.0)])
    full = _pair_contacts(orc64, w, 0, 1)
    assert len(full) == 4
    depths = sorted(d for _, _, d in full)
    assert depths[-1] - depths[0] > 0.02                                      # the tilt makes them differ: 2 deep, 2 shallow
    for _, n, _ in full:
        assert np.allclose(np.abs(n), [0, 1, 0], atol=1e-12)                   # the reference face is box 1's top
    pts = [tuple(np.round(p, 9)) for p, _, _ in full]
    for maxc in (1, 2, 3):
        c = _pair_contacts(orc64, w, 0, 1, maxc=maxc)
        assert len(c) == maxc
        assert abs(c[0][2] - depths[-1]) < 1e-12                               # deepest first
        assert all(tuple(np.round(p, 9)) in pts for p, _, _ in c)
    # closed form of the deepest point's depth: the dipping bottom corner sits at y = 0.95 - (0.5 cos t + 0.5 sin t)
    expect = 0.5 - (0.95 - 0.5 * (math.cos(tilt) + math.sin(tilt)))
    assert abs(depths[-1] - expect) < 1e-12


def test_kat6_box_box_parallel_edges_take_the_degenerate_branch(orc64):
    """edge-edge with (nearly) parallel edges: dLineClosestApproach's determinant d = 1 - (ua.ub)^2 is <= 1e-4 and ODE sets
    alpha = beta = 0, i.e. the contact is the midpoint of the two edges' reference points, unshifted [ODE-recall].  Two
    cubes turned 45 degrees about the SAME axis, ridge over ridge: the separating axis is an edge cross product only when the
    edges are not parallel, so with parallel ridges the face axes win and the result is a face contact -- what is pinned here
    is that nothing blows up and the contact is where the ridges meet."""
    a = math.pi / 4
    qz = np.array([math.cos(a / 2), 0, 0, math.sin(a / 2)])
    gap = 2 * math.sqrt(0.5) - 0.02
    w = orc64.world()
    w.add_boxes([(0.0, 0.0, 0.0), (0.0, gap, 0.0)], [tuple(qz), tuple(qz)], None, None, None, None,
                [(1.0, 1.0, 1.0), (1.0, 1.0, 1.0)])
    c = _pair_contacts(orc64, w, 0, 1)
    assert 1 <= len(c) <= 8
    for p, n, d in c:
        assert np.all(np.isfinite(p)) and np.all(np.isfinite(n)) and abs(np.linalg.norm(n) - 1) < 1e-12
        assert -1e-12 <= d <= 0.02 + 1e-12
        assert abs(p[0]) < 0.02 and abs(p[1] - gap / 2) < 0.02                 # on the line where the two ridges overlap
    assert max(d for _, _, d in c) > 0.0


def test_kat6_sphere_centre_inside_a_box(orc64):
    """dCollideSphereBox with the sphere's centre INSIDE the box: the contact sits at the centre, the normal is the box face
    nearest to it, the depth is the distance to that face plus the radius [ODE-recall]"""
    w = orc64.world()
    w.add_spheres([(0.3, 0.1, -0.05)], None, None, None, None, None, [0.1])
    w.add_boxes([(0.0, 0.0, 0.0)], None, None, None, None, None, [(1.0, 1.0, 1.0)])
    c = _pair_contacts(orc64, w, 0, 1)
    assert len(c) == 1
    p, n, d = c[0]
    assert np.allclose(p, [0.3, 0.1, -0.05], atol=1e-15)
    assert np.allclose(n, [1, 0, 0], atol=1e-15)                                # out through the +x face, 0.2 away
    assert abs(d - (0.2 + 0.1)) < 1e-15


def test_kat5_convex_plane_needs_points_on_both_sides():
    """dCollideConvexPlane returns contacts only if the hull has points on both sides of the plane (or on it): a hull wholly
    below the plane yields NONE [ODE-recall]; one that straddles it yields its penetrating points in array order"""
    from oracle.orc_ctypes import Oracle
    import itertools
    orc = Oracle("float64")
    pts = np.array(list(itertools.product((-0.5, 0.5), repeat=3)), float)
    for y, expect in ((-2.0, 0), (0.4, 4), (0.0, 4), (0.5, 4), (0.6, 0)):
        ow = orc.world()
        ow.set_hull(pts)
        g_plane = ow.add_plane(0, 1, 0, 0)
        ow.add_convex(np.array([[0.0, y, 0.0]]), np.array([[1.0, 0, 0, 0]]), np.zeros((1, 3)), np.zeros((1, 3)), np.ones(1), np.ones((1, 3)))
        out = (orc.ContactGeom * 16)()
        n = orc.lib.orc_collide(ow.w, g_plane + 1, g_plane, 8, out)
        assert n == expect, (y, n)
        for i in range(n):
            assert abs(out[i].depth - (0.5 - y)) < 1e-15 and list(out[i].normal) == [0.0, 1.0, 0.0]
            assert out[i].pos[1] == y - 0.5


# ----------------------------------------------------------------------------------------------------------------
# sphere against hull, hull against hull (round 3): the repository's own colliders -- closed-form cases
def _hull_world(orc, half, hull_poses, spheres=()):
    """convex bodies sharing one cube hull of half-extent `half` (8 points, 6 face planes), plus spheres (pos, radius);
    geoms are numbered in creation order: hulls first, then spheres"""
    import itertools
    pts = np.array(list(itertools.product((-half, half), repeat=3)), float)
    planes = []
    for a in range(3):
        for sg in (-1.0, 1.0):
            n = [0.0, 0.0, 0.0]; n[a] = sg
            planes.append(n + [half])
    ow = orc.world()
    ow.set_hull(pts); ow.set_hull_faces(np.array(planes))
    k = len(hull_poses)
    ow.add_convex(np.array([p for p, _ in hull_poses], float), np.array([q for _, q in hull_poses], float), np.zeros((k, 3)), np.zeros((k, 3)),
                  np.ones(k), np.ones((k, 3)))
    if spheres:
        m = len(spheres)
        ow.add_spheres(np.array([p for p, _ in spheres], float), None, None, None, np.ones(m), np.ones((m, 3)), np.array([r for _, r in spheres], float))
    return ow


IDQ = (1.0, 0.0, 0.0, 0.0)


def test_sphere_on_a_hull_face_and_beside_an_edge():
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    # sphere r = 0.3, centre 0.28 above the top face of a unit-cube hull: one contact, 0.02 deep, along +y (into the sphere, o1)
    ow = _hull_world(orc, 0.5, [((0.0, 0.0, 0.0), IDQ)], spheres=[((0.1, 0.78, -0.05), 0.3)])
    (pos, nrm, dep), = _collide(orc, ow, 1, 0)                    # dCollide(sphere, hull)
    assert np.allclose(nrm, [0, 1, 0]) and abs(dep - 0.02) < 1e-12 and np.allclose(pos, [0.1, 0.48, -0.05])
    (pos2, nrm2, dep2), = _collide(orc, ow, 0, 1)                 # (hull, sphere): swapped and flipped
    assert np.allclose(nrm2, [0, -1, 0]) and abs(dep2 - 0.02) < 1e-12 and np.allclose(pos2, pos)
    # out of reach of every face plane: nothing
    ow = _hull_world(orc, 0.5, [((0.0, 0.0, 0.0), IDQ)], spheres=[((0.1, 0.81, -0.05), 0.3)])
    assert _collide(orc, ow, 1, 0) == []
    # the stated over-estimate: diagonally off an edge the true distance is 0.2 sqrt(2) = 0.283 > r = 0.25, but the face planes'
    # maximum is 0.2: met early, 0.05 deep, along the FIRST of the two faces at that distance (+x comes before +y)
    ow = _hull_world(orc, 0.5, [((0.0, 0.0, 0.0), IDQ)], spheres=[((0.7, 0.7, 0.0), 0.25)])
    (pos, nrm, dep), = _collide(orc, ow, 1, 0)
    assert np.allclose(nrm, [1, 0, 0]) and abs(dep - 0.05) < 1e-12
    # a rotated, displaced hull: the same contact in the hull's frame
    a = 0.4
    q = (np.cos(a / 2), 0.0, 0.0, np.sin(a / 2))                  # about z
    up = np.array([-np.sin(a), np.cos(a), 0.0])                   # the hull's +y face normal in the world
    c = np.array([2.0, 1.0, -3.0]) + 0.78 * up
    ow = _hull_world(orc, 0.5, [((2.0, 1.0, -3.0), q)], spheres=[(tuple(c), 0.3)])
    (pos, nrm, dep), = _collide(orc, ow, 1, 0)
    assert np.allclose(nrm, up, atol=1e-12) and abs(dep - 0.02) < 1e-12 and np.allclose(pos, c - 0.3 * up, atol=1e-12)


def test_hull_against_hull_vertices_in_the_other():
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    # B shifted 0.9 along x (and a little in y, z): ONE vertex of B is inside A and ONE of A inside B (the other two corners of
    # the common face region are edge-edge crossings, which this collider does not see)
    ow = _hull_world(orc, 0.5, [((0.0, 0.0, 0.0), IDQ), ((0.9, 0.2, -0.15), IDQ)])
    cs = _collide(orc, ow, 0, 1)
    assert len(cs) == 2
    (p0, n0, d0), (p1, n1, d1) = cs
    assert np.allclose(p0, [0.4, -0.3, 0.35]) and np.allclose(n0, [-1, 0, 0]) and abs(d0 - 0.1) < 1e-12      # B's vertex in A: against A's +x face, into A
    assert np.allclose(p1, [0.5, 0.5, -0.5]) and np.allclose(n1, [-1, 0, 0]) and abs(d1 - 0.1) < 1e-12       # A's vertex in B: B's -x face normal
    flipped = _collide(orc, ow, 1, 0)
    assert len(flipped) == 2 and np.allclose(flipped[0][1], [1, 0, 0]) and np.allclose(flipped[0][0], p1)    # B is o1 now: A's vertex in B comes first
    assert len(_collide(orc, ow, 0, 1, maxc=1)) == 1
    # apart by a hair: nothing
    ow = _hull_world(orc, 0.5, [((0.0, 0.0, 0.0), IDQ), ((1.0001, 0.2, -0.15), IDQ)])
    assert _collide(orc, ow, 0, 1) == []


def test_a_hull_balanced_on_another_shares_its_weight_by_the_lever_rule():
    """an upper cube hull shifted by (0.1, 0.1) on a lower one that rests on the ground plane: the collider sees the two diagonal
    corners of the common square -- one vertex of each hull in the other -- and the line between them passes under the upper
    hull's centre, so one exact step leaves it at rest with the two contacts sharing m g as a lever does: the corner 0.4 sqrt(2)
    from the centre (the lower hull's, at (0.5, 0.5)) carries 5/9, the one 0.5 sqrt(2) away (the upper hull's own) 4/9"""
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    ow = _hull_world(orc, 0.5, [((0.0, 0.4999, 0.0), IDQ), ((0.1, 1.4997, 0.1), IDQ)])
    ow.add_plane(0.0, 1.0, 0.0, 0.0)
    orc.lib.orc_world_set_erp(ow.w, 0.0)
    ow.set_stepper(True)
    ow.tick(1.0 / 60.0)
    between = [j for j in ow.joints() if j[0] >= 0 and j[1] >= 0]
    assert len(between) == 2
    lam = {tuple(np.round(j[2], 6)[[0, 2]]): j[5] for j in between}
    assert abs(sum(lam.values()) - 9.8) < 1e-6
    near, far = lam[(-0.4, -0.4)], lam[(0.5, 0.5)]                  # the upper hull's own corner; the lower hull's corner
    assert abs(near - 9.8 * 4 / 9) < 1e-6 and abs(far - 9.8 * 5 / 9) < 1e-6
    assert np.max(np.abs(ow.state()[2][1])) < 1e-8                  # the upper hull stays put
