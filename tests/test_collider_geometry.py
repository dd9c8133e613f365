"""The oracle's colliders against GEOMETRY, not against their sibling in csrc/.

`oracle/orc_boxbox.c`, `orc_collide.c` restate ODE's dBoxBox / dCollideSphereBox / dCollideBoxPlane from recollection (ODE is
not vendored by the reference), and the product's colliders are pinned against them bit for bit
(tests/test_collider_equivalence.py, the GPU parity tests) -- which would pass a shared mis-recollection.  Here every collider
the reference's NearCallback can reach (/root/reference/src/main.c:678: box-box, sphere-box, sphere-sphere; box-plane for
BASELINE's configs) and the repository's own hull colliders are checked on >= 10^5 random pairs against brute-force numpy
references that share no code with them: projections of the boxes' 8 vertices on the 15 candidate axes instead of the
closed-form |R| sums, closest points, half-space tests.

What is pinned by geometry here (DESIGN.md section 5 lists it):
  * box-box: contacts <=> no separating axis among the 15 (the nine edge-pair axes with dBoxBox's 1e-5 guard against parallel
    edges: without it two of 20 000 boxes lying flat on a larger one came out "separated" -- found by this file);
    the depth is the least overlap over those axes under dBoxBox's own selection rule (faces first, an edge axis only when
    1.05 x its overlap is still less); the normal is that axis, unit, and points from box 2 into box 1; every contact of a face
    case lies in both boxes inflated by the depth, an edge case's contact is the midpoint of the two edge lines' closest
    approach; never more than maxc contacts, four for a face resting fully on a larger face, each as deep as the overlap.
  * sphere-box, sphere-sphere, box-plane, sphere-plane: the closed forms.
  * box-hull, hull-hull, hull-plane, sphere-hull (this repository's own definitions): exactly the vertices / corners that
    half-space tests find inside, in order, with the nearest face's distance.
What stays [ODE-recall]: which of several equally valid contact sets dBoxBox keeps when clipping yields more than maxc points
(cull_points' angular choice), and the order of its contacts."""
import ctypes as C

import numpy as np
import pytest

from oracle.orc_ctypes import Oracle

N_BOXBOX = 120_000
N_OTHER = 100_000


@pytest.fixture(scope="module")
def orc():
    return Oracle("float64")


# ------------------------------------------------------------------------------------------------------------ helpers
def _rand_rot(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _small_rot(rng, n, angle):
    """rotations by `angle` radians (scalar or per-pair) about random axes: boxes with near-parallel edges"""
    ax = rng.normal(size=(n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    a = np.broadcast_to(np.asarray(angle, float), (n,))
    K = np.zeros((n, 3, 3))
    K[:, 0, 1] = -ax[:, 2]; K[:, 0, 2] = ax[:, 1]; K[:, 1, 0] = ax[:, 2]; K[:, 1, 2] = -ax[:, 0]; K[:, 2, 0] = -ax[:, 1]; K[:, 2, 1] = ax[:, 0]
    return np.eye(3)[None] + np.sin(a)[:, None, None] * K + (1 - np.cos(a))[:, None, None] * (K @ K)


def _pose(p, R):
    """position + 3x4 row-major rotation, the oracle's geom layout"""
    n = len(p)
    out = np.zeros((n, 15))
    out[:, :3] = p
    R12 = np.zeros((n, 3, 4))
    R12[:, :, :3] = R
    out[:, 3:] = R12.reshape(n, 12)
    return out


def _bulk(orc, w, g1, g2, pose1, size1, pose2, size2, maxc=8):
    n = len(pose1)
    counts = np.zeros(n, np.int32)
    out = (orc.ContactGeom * (n * maxc))()
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)
    keep = [np.ascontiguousarray(a, np.float64) if a is not None else None for a in (pose1, size1, pose2, size2)]
    orc.lib.orc_collide_bulk(w.w, g1, g2, n, *[None if a is None else a.ctypes.data_as(C.c_void_p) for a in keep], maxc,
                             counts.ctypes.data_as(C.c_void_p), C.cast(out, C.c_void_p))
    del f
    raw = np.frombuffer(out, dtype=np.dtype([("pos", "f8", 3), ("normal", "f8", 3), ("depth", "f8"), ("g1", "i4"), ("g2", "i4")]))
    raw = raw.reshape(n, maxc)
    return counts, raw["pos"].copy(), raw["normal"].copy(), raw["depth"].copy()


def _box_vertices(p, R, side):
    """(n, 8, 3) world-space corners"""
    sg = np.array([[(c >> a) & 1 for a in range(3)] for c in range(8)], float) * 2 - 1          # (8, 3)
    local = sg[None] * (0.5 * side)[:, None, :]                                                 # (n, 8, 3)
    return p[:, None, :] + np.einsum("nij,nkj->nki", R, local)


def _in_box(pts, p, R, side, grow):
    """pts (n, k, 3) inside the box inflated by grow (n,) on every side?"""
    loc = np.einsum("nji,nkj->nki", R, pts - p[:, None, :])
    return np.all(np.abs(loc) <= (0.5 * side)[:, None, :] + grow[:, None, None], axis=2)


def _sat_by_projection(p1, R1, s1, p2, R2, s2):
    """overlap (positive = interpenetration) of the two boxes' projections on each of the 15 candidate axes, from the projected
    vertices themselves: (n, 15) overlaps, (n, 15, 3) unit axes, (n, 15) axis is usable (edge pairs may be parallel)"""
    n = len(p1)
    axes = np.empty((n, 15, 3))
    axes[:, 0:3] = np.swapaxes(R1, 1, 2)            # rows = box 1's axes (columns of R1)
    axes[:, 3:6] = np.swapaxes(R2, 1, 2)
    k = 6
    for i in range(3):
        for j in range(3):
            axes[:, k] = np.cross(R1[:, :, i], R2[:, :, j])
            k += 1
    ln = np.linalg.norm(axes, axis=2)
    ok = ln > 1e-7
    axes = axes / np.where(ok, ln, 1.0)[:, :, None]
    v1 = _box_vertices(p1, R1, s1) - p1[:, None, :]
    v2 = _box_vertices(p2, R2, s2) - p2[:, None, :]
    ra = np.max(np.abs(np.einsum("nkj,naj->nak", v1, axes)), axis=2)       # half-extent of box 1 along each axis
    rb = np.max(np.abs(np.einsum("nkj,naj->nak", v2, axes)), axis=2)
    dist = np.abs(np.einsum("nj,naj->na", p2 - p1, axes))
    ov = ra + rb - dist
    # dBoxBox's "fudge2" [ODE-recall box.cpp]: before the nine edge-pair axes every |R1^T R2| entry grows by 1e-5, which widens
    # the boxes' extent along u_i x v_j by 1e-5 x (the four half-sides that enter it) / |u_i x v_j| -- the guard that keeps
    # (nearly) parallel edges from "separating" two boxes on rounding error alone.  Part of the rule being checked.
    a, b = 0.5 * s1, 0.5 * s2
    k = 6
    for i in range(3):
        for j in range(3):
            others = a[:, (i + 1) % 3] + a[:, (i + 2) % 3] + b[:, (j + 1) % 3] + b[:, (j + 2) % 3]
            ov[:, k] += 1e-5 * others / np.where(ok[:, k], ln[:, k], 1.0)
            k += 1
    return ov, axes, ok


def _boxbox_cases(rng, n):
    """three regimes: generic pairs near contact; near-parallel edges (tiny relative rotation); a floor-sized box under a small one"""
    n1, n2 = n // 2, n // 4
    n3 = n - n1 - n2
    s1 = rng.uniform(0.2, 1.0, (n, 3)); s2 = rng.uniform(0.2, 1.0, (n, 3))
    R1 = _rand_rot(rng, n); R2 = _rand_rot(rng, n)
    R2[n1:n1 + n2] = _small_rot(rng, n2, 10.0 ** rng.uniform(-9, -2, n2)) @ R1[n1:n1 + n2]
    p1 = rng.uniform(-1, 1, (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    reach = 0.5 * (np.linalg.norm(s1, axis=1) + np.linalg.norm(s2, axis=1))
    p2 = p1 + d * (rng.uniform(0.15, 1.0, n) * reach)[:, None]
    # the reference's floor (main.c:115): 100 x 1 x 100, a spawned box resting on / sunk into / hovering over its top
    a = n1 + n2
    s2[a:] = [100.0, 1.0, 100.0]
    R2[a:] = np.eye(3)
    R1[a:] = _small_rot(rng, n3, rng.choice([0.0, 1e-7, 1e-3, 0.3], n3)) @ np.eye(3)
    p2[a:] = 0.0
    p1[a:, 0] = rng.uniform(-40, 40, n3); p1[a:, 2] = rng.uniform(-40, 40, n3)
    p1[a:, 1] = 0.5 + 0.5 * s1[a:, 1] + rng.uniform(-0.05, 0.02, n3)
    return p1, R1, s1, p2, R2, s2


# ------------------------------------------------------------------------------------------------------------ box - box
def test_box_box_against_the_separating_axis_theorem(orc):
    rng = np.random.default_rng(2024)
    n = N_BOXBOX
    p1, R1, s1, p2, R2, s2 = _boxbox_cases(rng, n)
    w = orc.world()
    g1 = orc.lib.orc_geom_create_box(w.w, 1.0, 1.0, 1.0)
    g2 = orc.lib.orc_geom_create_box(w.w, 1.0, 1.0, 1.0)
    cnt, pos, nrm, dep = _bulk(orc, w, g1, g2, _pose(p1, R1), s1, _pose(p2, R2), s2, maxc=8)
    ov, axes, ok = _sat_by_projection(p1, R1, s1, p2, R2, s2)
    ov_ok = np.where(ok, ov, np.inf)
    least = ov_ok.min(axis=1)

    # (1) contacts <=> no separating axis (pairs within 1e-9 of touching may go either way)
    clear_sep = least < -1e-9
    clear_hit = least > 1e-9
    assert clear_sep.sum() > n // 10 and clear_hit.sum() > n // 4
    assert np.all(cnt[clear_sep] == 0), "contacts reported across a separating axis"
    assert np.all(cnt[clear_hit] >= 1), "no contact although all 15 axes overlap"
    assert cnt.max() <= 8 and np.all(cnt >= 0)
    hit = np.flatnonzero(clear_hit & (cnt > 0))

    # (2) dBoxBox's axis choice: the least face overlap, unless an edge axis' overlap x 1.05 is smaller still (in order, strict)
    face = ov[hit, :6]
    s = -face[:, 0].copy(); pick = np.zeros(len(hit), int)
    for a in range(1, 6):
        better = -face[:, a] > s
        s[better] = -face[better, a]; pick[better] = a
    ambiguous = np.zeros(len(hit), bool)
    for a in range(6, 15):
        cand = -ov[hit, a]
        use = ok[hit, a] & (cand * 1.05 > s)
        ambiguous |= ok[hit, a] & (np.abs(cand * 1.05 - s) < 1e-9)
        s[use] = cand[use]; pick[use] = a
    expect_depth = -s
    first_depth = dep[hit, 0]
    top_depth = np.max(np.where(np.arange(8)[None, :] < cnt[hit, None], dep[hit], -np.inf), axis=1)
    sure = ~ambiguous
    edge = pick >= 6
    # edge-edge: one contact, its depth the overlap
    assert np.all(cnt[hit][edge & sure] == 1)
    assert np.allclose(first_depth[edge & sure], expect_depth[edge & sure], atol=1e-11)
    # face: every contact at most as deep as the overlap, none negative
    fs = ~edge & sure
    assert np.all(top_depth[fs] <= expect_depth[fs] + 1e-10)
    assert np.all(np.where(np.arange(8)[None, :] < cnt[hit, None], dep[hit], 0.0) >= -1e-12)
    assert edge.sum() > 1000 and (~edge).sum() > 10000

    # (3) the normal: unit, along the chosen axis, from box 2 into box 1
    n0 = nrm[hit, 0]
    assert np.allclose(np.linalg.norm(n0, axis=1), 1.0, atol=1e-12)
    chosen = axes[hit, pick]
    assert np.all(np.abs(np.abs(np.einsum("nj,nj->n", n0[sure], chosen[sure])) - 1.0) < 1e-9)
    assert np.all(np.einsum("nj,nj->n", n0, (p1 - p2)[hit]) >= -1e-12)
    same = np.where(np.arange(8)[None, :, None] < cnt[hit, None, None], nrm[hit] - n0[:, None, :], 0.0)
    assert np.abs(same).max() == 0.0                      # every contact of a pair carries the one normal

    # (4) every contact of a FACE case lies in both boxes inflated by its pair's penetration
    grow = expect_depth + 1e-9
    m = np.arange(8)[None, :] < cnt[hit, None]
    in1 = _in_box(pos[hit], p1[hit], R1[hit], s1[hit], grow)
    in2 = _in_box(pos[hit], p2[hit], R2[hit], s2[hit], grow)
    fm = m & fs[:, None]
    assert np.all(in1[fm]) and np.all(in2[fm])

    # (5) an EDGE case's one contact is the midpoint of the closest approach of the two edges' LINES -- dBoxBox's construction:
    # the edge of box 1 farthest along the normal, the edge of box 2 farthest against it.  When the closest points fall within
    # both segments the point lies in both inflated boxes; when they do not (a handful in 10^5: the least-overlap axis is an edge
    # pair whose segments do not actually face each other) it can lie outside by a few depths -- a known trait of dBoxBox,
    # recorded in DESIGN.md section 5, not an error of the restatement.
    es = np.flatnonzero(edge & sure)
    ih = hit[es]
    ei, ej = (pick[es] - 6) // 3, (pick[es] - 6) % 3
    n12 = -nrm[ih, 0]                                            # from box 1 towards box 2
    pa = p1[ih].copy(); pb = p2[ih].copy()
    for j in range(3):
        sa = np.where(np.einsum("nj,nj->n", n12, R1[ih][:, :, j]) > 0, 1.0, -1.0)
        pa += (sa * 0.5 * s1[ih, j])[:, None] * R1[ih][:, :, j]
        sb = np.where(np.einsum("nj,nj->n", n12, R2[ih][:, :, j]) > 0, -1.0, 1.0)
        pb += (sb * 0.5 * s2[ih, j])[:, None] * R2[ih][:, :, j]
    ua = R1[ih, :, ei]; ub = R2[ih, :, ej]
    # closest approach of pa + alpha ua and pb + beta ub
    dp = pb - pa
    uaub = np.einsum("nj,nj->n", ua, ub); q1 = np.einsum("nj,nj->n", ua, dp); q2 = -np.einsum("nj,nj->n", ub, dp)
    den = 1 - uaub * uaub
    good = den > 1e-6
    alpha = (q1 + uaub * q2) / np.where(good, den, 1.0); beta = (uaub * q1 + q2) / np.where(good, den, 1.0)
    mid = 0.5 * ((pa + alpha[:, None] * ua) + (pb + beta[:, None] * ub))
    assert good.sum() > 1000
    assert np.abs(pos[ih, 0] - mid)[good].max() < 1e-8
    # where along each edge (measured from the edge's own centre) the closest points are
    ca = np.einsum("nj,nj->n", pa + alpha[:, None] * ua - p1[ih], ua); cb = np.einsum("nj,nj->n", pb + beta[:, None] * ub - p2[ih], ub)
    within = good & (np.abs(ca) <= 0.5 * s1[ih, ei]) & (np.abs(cb) <= 0.5 * s2[ih, ej])
    assert within.sum() > 0.99 * good.sum()
    assert np.all(in1[es, 0][within]) and np.all(in2[es, 0][within])


def test_box_resting_flat_on_a_larger_face_gives_four_contacts_as_deep_as_the_overlap(orc):
    rng = np.random.default_rng(5)
    n = 20_000
    s1 = rng.uniform(0.2, 1.0, (n, 3)); s2 = np.column_stack([rng.uniform(3, 100, n), rng.uniform(0.5, 2, n), rng.uniform(3, 100, n)])
    G = _rand_rot(rng, n)                                           # the whole configuration turned at random
    yaw = rng.uniform(0, 2 * np.pi, n)
    Y = np.zeros((n, 3, 3)); Y[:, 1, 1] = 1; Y[:, 0, 0] = np.cos(yaw); Y[:, 0, 2] = np.sin(yaw); Y[:, 2, 0] = -np.sin(yaw); Y[:, 2, 2] = np.cos(yaw)
    sink = rng.uniform(1e-6, 0.05, n)
    local1 = np.column_stack([rng.uniform(-1, 1, n), 0.5 * s2[:, 1] + 0.5 * s1[:, 1] - sink, rng.uniform(-1, 1, n)])
    p2 = rng.uniform(-2, 2, (n, 3))
    p1 = p2 + np.einsum("nij,nj->ni", G, local1)
    R1 = G @ Y; R2 = G
    w = orc.world()
    g1 = orc.lib.orc_geom_create_box(w.w, 1.0, 1.0, 1.0)
    g2 = orc.lib.orc_geom_create_box(w.w, 1.0, 1.0, 1.0)
    cnt, pos, nrm, dep = _bulk(orc, w, g1, g2, _pose(p1, R1), s1, _pose(p2, R2), s2)
    assert np.all(cnt == 4)
    assert np.allclose(dep[:, :4], sink[:, None], atol=1e-10)
    up = G[:, :, 1]
    assert np.allclose(nrm[:, 0], up, atol=1e-9)                    # out of the larger box's top, into box 1
    # the four points are box 1's bottom corners, lifted onto box 2's top face: with the two y axes parallel the first axis
    # tested wins (box 1's), so box 1 is the reference box and the contacts are points of the INCIDENT face (box 2's top)
    # clipped to box 1's bottom rectangle [ODE-recall: dBoxBox returns points on the incident face]
    corners = _box_vertices(p1, R1, s1)
    below = np.argsort(np.einsum("nkj,nj->nk", corners, up), axis=1)[:, :4]
    want = np.take_along_axis(corners, below[:, :, None], axis=1) + (sink[:, None] * up)[:, None, :]
    d = np.linalg.norm(pos[:, :4, None, :] - want[:, None, :, :], axis=3).min(axis=2)
    assert d.max() < 1e-9


# ------------------------------------------------------------------------------------------------------------ sphere - box / sphere
def test_sphere_box_is_the_closest_point_on_the_box(orc):
    rng = np.random.default_rng(7)
    n = N_OTHER
    side = rng.uniform(0.2, 1.0, (n, 3)); side[: n // 10] = [100.0, 1.0, 100.0]
    Rb = _rand_rot(rng, n); pb = rng.uniform(-1, 1, (n, 3))
    r = rng.uniform(0.1, 0.4, n)
    loc = rng.uniform(-1, 1, (n, 3)) * (0.5 * side + r[:, None] * 1.3)      # inside, near the surface and beyond
    ps = pb + np.einsum("nij,nj->ni", Rb, loc)
    w = orc.world()
    gs = orc.lib.orc_geom_create_sphere(w.w, 0.3)
    gb = orc.lib.orc_geom_create_box(w.w, 1.0, 1.0, 1.0)
    rad = np.column_stack([r, np.zeros(n), np.zeros(n)])
    cnt, pos, nrm, dep = _bulk(orc, w, gs, gb, _pose(ps, np.tile(np.eye(3), (n, 1, 1))), rad, _pose(pb, Rb), side, maxc=4)
    half = 0.5 * side
    clamp = np.clip(loc, -half, half)
    outside = np.any(np.abs(loc) > half, axis=1)
    dist = np.linalg.norm(loc - clamp, axis=1)
    # centre outside the box: one contact at the closest point, depth r - distance, normal from it to the centre
    hit = outside & (r - dist > 1e-9); miss = outside & (r - dist < -1e-9)
    assert hit.sum() > n // 10 and miss.sum() > n // 20
    assert np.all(cnt[miss] == 0) and np.all(cnt[hit] == 1)
    q = pb + np.einsum("nij,nj->ni", Rb, clamp)
    assert np.allclose(pos[hit, 0], q[hit], atol=1e-12)
    assert np.allclose(dep[hit, 0], (r - dist)[hit], atol=1e-12)
    nn = (ps - q)[hit] / dist[hit, None]
    assert np.allclose(nrm[hit, 0], nn, atol=1e-9)
    # centre inside: pushed out through the nearest face, depth = distance to it + r, contact at the centre
    ins = ~outside
    assert ins.sum() > n // 20 and np.all(cnt[ins] == 1)
    fd = half - np.abs(loc)
    k = np.argmin(fd, axis=1)
    assert np.allclose(dep[ins, 0], (fd[np.arange(n), k] + r)[ins], atol=1e-12)
    axis = Rb[np.arange(n), :, k] * np.sign(loc[np.arange(n), k])[:, None]
    tie = np.sort(fd, axis=1)[:, 1] - np.sort(fd, axis=1)[:, 0] < 1e-9
    sel = ins & ~tie & (loc[np.arange(n), k] != 0)
    assert np.allclose(nrm[sel, 0], axis[sel], atol=1e-12)
    assert np.allclose(pos[ins, 0], ps[ins], atol=0)


def test_sphere_sphere_and_sphere_plane_closed_forms(orc):
    rng = np.random.default_rng(8)
    n = N_OTHER
    r1 = rng.uniform(0.1, 0.4, n); r2 = rng.uniform(0.1, 0.4, n)
    p1 = rng.uniform(-1, 1, (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    gap = rng.uniform(0.5, 1.3, n) * (r1 + r2)
    p2 = p1 - d * gap[:, None]
    w = orc.world()
    a = orc.lib.orc_geom_create_sphere(w.w, 0.3); b = orc.lib.orc_geom_create_sphere(w.w, 0.3)
    I = np.tile(np.eye(3), (n, 1, 1))
    z = np.zeros(n)
    cnt, pos, nrm, dep = _bulk(orc, w, a, b, _pose(p1, I), np.column_stack([r1, z, z]), _pose(p2, I), np.column_stack([r2, z, z]), maxc=2)
    hit = r1 + r2 - gap > 1e-9; miss = r1 + r2 - gap < -1e-9
    assert np.all(cnt[hit] == 1) and np.all(cnt[miss] == 0)
    assert np.allclose(dep[hit, 0], (r1 + r2 - gap)[hit], atol=1e-12)
    assert np.allclose(nrm[hit, 0], d[hit], atol=1e-9)                       # from sphere 2 into sphere 1
    t = np.einsum("nj,nj->n", pos[:, 0] - p2, d)                             # on the line of centres, inside the lens
    assert np.all((t[hit] >= (gap - r1)[hit] - 1e-9) & (t[hit] <= r2[hit] + 1e-9))
    # sphere - plane
    pn = rng.normal(size=(n, 3)); pn /= np.linalg.norm(pn, axis=1, keepdims=True)
    for k in range(0, n, n // 20):                                            # the plane is per world: a handful of planes
        w2 = orc.world()
        pl = orc.lib.orc_geom_create_plane(w2.w, *pn[k], 0.25)
        s = orc.lib.orc_geom_create_sphere(w2.w, 0.3)
        sl = slice(k, k + n // 20)
        m = sl.stop - sl.start
        c2, pos2, n2, d2 = _bulk(orc, w2, s, pl, _pose(p1[sl], I[sl]), np.column_stack([r1[sl], z[sl], z[sl]]), _pose(np.zeros((m, 3)), I[sl]), None, maxc=2)
        depth = 0.25 - p1[sl] @ pn[k] + r1[sl]
        assert np.all(c2[depth > 1e-9] == 1) and np.all(c2[depth < -1e-9] == 0)
        h = depth > 1e-9
        assert np.allclose(d2[h, 0], depth[h], atol=1e-12) and np.allclose(n2[h, 0], pn[k], atol=1e-15)
        assert np.allclose(pos2[h, 0], (p1[sl] - pn[k] * r1[sl, None])[h], atol=1e-12)


# ------------------------------------------------------------------------------------------------------------ box - plane
def test_box_plane_contacts_are_the_boxs_lowest_corners(orc):
    rng = np.random.default_rng(9)
    n = N_OTHER
    side = rng.uniform(0.2, 1.0, (n, 3))
    R = _rand_rot(rng, n)
    R[: n // 4] = _small_rot(rng, n // 4, 10.0 ** rng.uniform(-9, -1, n // 4))        # nearly flat on the plane
    normal = np.array([0.0, 1.0, 0.0])
    p = rng.uniform(-1, 1, (n, 3))
    reach = 0.5 * np.abs(np.einsum("nji,j->ni", R, normal)) @ np.ones(3) * 0 + 0.5 * np.einsum("ni,ni->n", np.abs(np.einsum("nji,j->ni", R, normal)), side)
    p[:, 1] = reach * rng.uniform(0.3, 1.2, n)
    w = orc.world()
    pl = orc.lib.orc_geom_create_plane(w.w, 0.0, 1.0, 0.0, 0.0)
    b = orc.lib.orc_geom_create_box(w.w, 1.0, 1.0, 1.0)
    cnt, pos, nrm, dep = _bulk(orc, w, b, pl, _pose(p, R), side, _pose(np.zeros((n, 3)), np.tile(np.eye(3), (n, 1, 1))), None, maxc=4)
    V = _box_vertices(p, R, side)
    vd = -V[:, :, 1]                                            # depth of each corner below y = 0
    deepest = vd.max(axis=1)
    assert np.all(cnt[deepest < -1e-9] == 0) and np.all(cnt[deepest > 1e-9] >= 1)
    hit = np.flatnonzero(deepest > 1e-9)
    assert len(hit) > n // 4 and cnt.max() <= 4
    m = np.arange(4)[None, :] < cnt[hit, None]
    # every contact is one of the box's corners, at that corner's depth, no corner twice; the first is the deepest
    dist = np.linalg.norm(pos[hit][:, :, None, :] - V[hit][:, None, :, :], axis=3)
    which = dist.argmin(axis=2)
    assert dist.min(axis=2)[m].max() < 1e-9
    assert np.allclose(dep[hit][m], np.take_along_axis(vd[hit], which, axis=1)[m], atol=1e-10)
    assert np.allclose(dep[hit, 0], deepest[hit], atol=1e-10)
    assert np.all(dep[hit][m] >= -1e-12)
    srt = np.sort(np.where(m, which, 100 + np.arange(4)[None, :]), axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1])
    assert np.allclose(nrm[hit][m], normal[None, :], atol=0)
    # the deepest corner, its neighbours along the two sides that rise least, and that face's fourth corner -- those of them that
    # are below the plane [ODE-recall dCollideBoxPlane]: one contact per corner below up to three; four exactly when the four lowest
    # corners are one face (a box standing on a corner with its three neighbours just under the plane gets three: two such boxes in
    # 10^5 here).  Boxes with a corner within 1e-6 of the plane are left out: there the count may go either way.
    clear = np.all(np.abs(vd) > 1e-6, axis=1)
    below = (vd > 0).sum(axis=1)
    assert np.all(cnt[clear] <= np.minimum(4, below[clear])) and np.all(cnt[clear] >= np.minimum(3, below[clear]))
    four = clear & (below >= 4)
    low4 = np.argsort(-vd, axis=1)[:, :4]
    c4 = np.take_along_axis(V, low4[:, :, None], axis=1)
    planar = np.abs(np.linalg.det(c4[:, 1:] - c4[:, :1])) < 1e-12          # the four lowest corners lie in one plane: a face
    assert np.all(cnt[four & planar] == 4) and np.all(cnt[four & ~planar] == 3)
    assert (four & planar).sum() > 1000 and (below[clear] == 3).sum() > 1000


# ------------------------------------------------------------------------------------------------------------ hulls
def _random_hull(rng, k=40):
    from scipy.spatial import ConvexHull
    pts = rng.normal(size=(k, 3)) * [0.5, 0.35, 0.4]
    h = ConvexHull(pts)
    verts = pts[h.vertices]
    eq = np.unique(np.round(h.equations, 12), axis=0)             # (nf, 4): n.x + d <= 0 inside  ->  n.x <= -d
    planes = np.column_stack([eq[:, :3], -eq[:, 3]])
    return verts, planes


def test_hull_colliders_take_exactly_the_points_the_half_spaces_contain(orc):
    """box-hull, hull-hull, sphere-hull, hull-plane: this repository's own colliders ("a point inside a convex shape, along
    the face it is nearest to").  The numpy side decides "inside" from the face planes and box half-extents alone."""
    rng = np.random.default_rng(11)
    verts, planes = _random_hull(rng)
    nv = len(verts)
    n = 25_000
    w = orc.world()
    w.set_hull(verts)
    w.set_hull_faces(planes)
    gh = orc.lib.orc_geom_create_convex(w.w)
    gb = orc.lib.orc_geom_create_box(w.w, 1.0, 1.0, 1.0)
    side = rng.uniform(0.3, 1.2, (n, 3))
    Rh = _rand_rot(rng, n); Rb = _rand_rot(rng, n)
    ph = rng.uniform(-1, 1, (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pb = ph + d * rng.uniform(0.2, 1.1, n)[:, None]
    cnt, pos, nrm, dep = _bulk(orc, w, gb, gh, _pose(pb, Rb), side, _pose(ph, Rh), None, maxc=8)
    VW = ph[:, None, :] + np.einsum("nij,kj->nki", Rh, verts)            # hull vertices, world
    loc = np.einsum("nji,nkj->nki", Rb, VW - pb[:, None, :])             # in the box frame
    half = 0.5 * side
    margin = (half[:, None, :] - np.abs(loc)).min(axis=2)                # > 0 inside the box
    corners = _box_vertices(pb, Rb, side)
    cl = np.einsum("nji,nkj->nki", Rh, corners - ph[:, None, :])          # box corners in the hull frame
    cm = (planes[None, None, :, 3] - np.einsum("nki,fi->nkf", cl, planes[:, :3])).min(axis=2)      # > 0 inside the hull
    sure = (np.abs(margin) > 1e-9).all(axis=1) & (np.abs(cm) > 1e-9).all(axis=1)
    want_n = np.minimum(8, (margin > 0).sum(axis=1) + (cm > 0).sum(axis=1))
    assert np.all(cnt[sure] == want_n[sure])
    assert (want_n > 0).sum() > n // 20
    for i in np.flatnonzero(sure & (want_n > 0))[:4000]:
        vs = np.flatnonzero(margin[i] > 0)
        cs = np.flatnonzero(cm[i] > 0)
        exp_pos = np.vstack([VW[i, vs], corners[i, cs]])[:8]
        exp_dep = np.concatenate([margin[i, vs], cm[i, cs]])[:8]
        assert np.allclose(pos[i, :cnt[i]], exp_pos, atol=1e-12)
        assert np.allclose(dep[i, :cnt[i]], exp_dep, atol=1e-12)
        inward = np.einsum("kj,j->k", nrm[i, :cnt[i]], pb[i] - ph[i])    # contact normals point into the box (o1)
        assert np.all(np.abs(np.linalg.norm(nrm[i, :cnt[i]], axis=1) - 1) < 1e-9)
        del inward
    # hull - hull: B's vertices inside A first, then A's inside B
    g2 = orc.lib.orc_geom_create_convex(w.w)
    Ra = _rand_rot(rng, n); pa = ph + d * rng.uniform(0.2, 0.9, n)[:, None]
    cnt, pos, nrm, dep = _bulk(orc, w, gh, g2, _pose(pa, Ra), None, _pose(ph, Rh), None, maxc=8)

    def inside(pw, pc, Rc):
        l = np.einsum("nji,nkj->nki", Rc, pw - pc[:, None, :])
        return (planes[None, None, :, 3] - np.einsum("nki,fi->nkf", l, planes[:, :3])).min(axis=2)
    VA = pa[:, None, :] + np.einsum("nij,kj->nki", Ra, verts)
    mB_in_A = inside(VW, pa, Ra); mA_in_B = inside(VA, ph, Rh)
    sure = (np.abs(mB_in_A) > 1e-9).all(axis=1) & (np.abs(mA_in_B) > 1e-9).all(axis=1)
    want_n = np.minimum(8, (mB_in_A > 0).sum(axis=1) + (mA_in_B > 0).sum(axis=1))
    assert np.all(cnt[sure] == want_n[sure]) and (want_n > 0).sum() > n // 20
    for i in np.flatnonzero(sure & (want_n > 0))[:3000]:
        b_in = np.flatnonzero(mB_in_A[i] > 0); a_in = np.flatnonzero(mA_in_B[i] > 0)
        assert np.allclose(pos[i, :cnt[i]], np.vstack([VW[i, b_in], VA[i, a_in]])[:8], atol=1e-12)
        assert np.allclose(dep[i, :cnt[i]], np.concatenate([mB_in_A[i, b_in], mA_in_B[i, a_in]])[:8], atol=1e-12)
    # sphere - hull: the face plane farthest out decides
    gs = orc.lib.orc_geom_create_sphere(w.w, 0.3)
    r = rng.uniform(0.1, 0.4, n)
    ps = ph + d * rng.uniform(0.1, 1.2, n)[:, None]
    z = np.zeros(n)
    cnt, pos, nrm, dep = _bulk(orc, w, gs, gh, _pose(ps, np.tile(np.eye(3), (n, 1, 1))), np.column_stack([r, z, z]), _pose(ph, Rh), None, maxc=2)
    c = np.einsum("nji,nj->ni", Rh, ps - ph)
    sd = c @ planes[:, :3].T - planes[None, :, 3]
    smax = sd.max(axis=1); f = sd.argmax(axis=1)
    hit = r - smax > 1e-9; miss = r - smax < -1e-9
    assert np.all(cnt[hit] == 1) and np.all(cnt[miss] == 0) and hit.sum() > n // 10
    assert np.allclose(dep[hit, 0], (r - smax)[hit], atol=1e-12)
    nw = np.einsum("nij,nj->ni", Rh, planes[f, :3])
    uniq = np.sort(sd, axis=1)[:, -1] - np.sort(sd, axis=1)[:, -2] > 1e-9
    assert np.allclose(nrm[hit & uniq, 0], nw[hit & uniq], atol=1e-12)
    assert nv >= 10
