"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle on the same
seeded inputs, plus size-independent properties at BASELINE.json's full sizes.

Tolerance: north_star asks for 1e-5 relative on positions / quaternions after
N steps.  The library is built without FMA contraction and mirrors the step's
evaluation order, so on single-body islands the result is expected to be
bit-identical to the oracle in both precisions; the tests assert exact
equality where that holds and the 1e-5 bound everywhere.
"""
import numpy as np
import pytest

from __graft_entry__ import load_package

pkg = load_package()
pytestmark = pytest.mark.gpu

H = 1.0 / 60.0
REL_TOL = 1e-5          # BASELINE.json north_star


def _steps_without_body_pairs(orc, scene, steps, spheres=False):
    """Number of leading ticks in which no two bodies' AABBs overlap (every island is one body).
    The fused single-body-island path is compared over exactly that span; multi-body islands are the
    general path's job."""
    ow = _oracle_build(orc, scene, spheres=spheres)
    for k in range(steps):
        ow.tick(H)
        if ow.n_body_pairs() > 0:
            return k
    return steps


def _oracle_build(orc, scene, gyro=None, spheres=False, setup=None):
    ow = orc.world()
    if gyro is not None:
        orc.lib.orc_world_set_gyro_mode(ow.w, gyro)
    if setup:
        setup(orc, ow)
    if scene.plane is not None:
        ow.add_plane(*scene.plane)
    if scene.hull_points is not None:
        ow.set_hull(scene.hull_points)
        if scene.hull_planes is not None:
            ow.set_hull_faces(scene.hull_planes)
    for sides, pos, R12 in (scene.static_boxes or []):
        ow.add_static_box(sides, pos, R12)
    convex = scene.gtype == pkg.scenes.GEOM_CONVEX
    if spheres:
        ow.add_spheres(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia,
                       scene.sides[:, 0])
    elif convex.any():
        # boxes first, hulls behind them (the scenes that mix the two are laid out that way)
        nb = int(np.argmax(convex)) if not convex.all() else 0
        assert not convex[:nb].any() and convex[nb:].all()
        if nb:
            ow.add_boxes(scene.pos[:nb], scene.quat[:nb], scene.lvel[:nb], scene.avel[:nb], scene.mass[:nb, 0],
                         scene.inertia[:nb], scene.sides[:nb])
        ow.add_convex(scene.pos[nb:], scene.quat[nb:], scene.lvel[nb:], scene.avel[nb:], scene.mass[nb:, 0],
                      scene.inertia[nb:])
    else:
        ow.add_boxes(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia, scene.sides)
    return ow


def _oracle_run(orc, scene, steps, gyro=None, spheres=False, setup=None, allow_pairs=False):
    ow = _oracle_build(orc, scene, gyro=gyro, spheres=spheres, setup=setup)
    ow.run(H, steps)
    if not allow_pairs:
        assert ow.n_body_pairs() == 0, "scene left the single-body-island regime"
    return ow


def _gpu_run(scene, dtype, steps, gyro=None, setup=None):
    w = pkg.BatchWorld(scene.n, dtype=dtype)
    if gyro is not None:
        w.set_gyro_mode(gyro)
    if setup:
        setup(w)
    w.load_scene(scene)
    w.step(H, steps)
    w.synchronize()
    return w


def _rel_err(a, b):
    scale = np.maximum(np.abs(b), 1.0)
    return float(np.max(np.abs(a - b) / scale))


def _compare(got, ref, exact=True):
    for name, a, b in zip(("pos", "quat", "lvel", "avel"), got, ref):
        assert np.all(np.isfinite(a)), name
        assert _rel_err(a, b) <= REL_TOL, f"{name}: rel err {_rel_err(a, b)}"
        if exact:
            assert np.array_equal(a, b), f"{name}: not bit-identical, max abs diff {np.max(np.abs(a - b))}"


def _orc(dtype):
    from oracle.orc_ctypes import Oracle
    return Oracle(dtype)


# ----------------------------------------------------------------- free flight (configs[1] shape)
@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("box_mass", [False, True])
def test_free_flight_matches_oracle(dtype, box_mass):
    scene = pkg.scenes.box_grid(64, 64, seed=1, spin=True, box_mass=box_mass, plane=False).astype(dtype)
    w = _gpu_run(scene, dtype, 600)
    ow = _oracle_run(_orc(dtype), scene, 600)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == 0


@pytest.mark.parametrize("gyro", [0, 1, 2])
def test_gyro_modes_match_oracle(gyro):
    scene = pkg.scenes.box_grid(32, 16, seed=5, spin=True, box_mass=True, plane=False).astype("float64")
    scene.avel[:] *= 4.0
    w = _gpu_run(scene, "float64", 240, gyro=gyro)
    ow = _oracle_run(_orc("float64"), scene, 240, gyro=gyro)
    _compare(w.state(), ow.state())


@pytest.mark.parametrize("n", [1, 3, 255, 256, 257, 1000])
def test_ragged_sizes(n):
    full = pkg.scenes.box_grid(40, 25, seed=3, spin=True, box_mass=True, plane=False)
    scene = full.slice(0, n).astype("float32")
    w = _gpu_run(scene, "float32", 50)
    ow = _oracle_run(_orc("float32"), scene, 50)
    _compare(w.state(), ow.state())


# ----------------------------------------------------------------- contact path (configs[0] / configs[2])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_config1_boxes_on_plane_matches_oracle(dtype):
    """BASELINE configs[0]: 1 024 boxes over the ground plane, 600 QuickSteps at dt = 1/60."""
    scene = pkg.scenes.config1().astype(dtype)
    steps = 600
    # boxes land by tick ~192 (fall from <= 50 m); from tick ~340 on a few toppled boxes reach a neighbour, so
    # the run covers the fused path, the safe-zone rollback and the exact pair / island path
    assert _steps_without_body_pairs(_orc(dtype), scene, steps) < steps
    w = _gpu_run(scene, dtype, steps)
    ow = _oracle_run(_orc(dtype), scene, steps, allow_pairs=True)
    _compare(w.state(), ow.state())
    st = w.collision_stats()
    assert st["fast_ticks"] >= 96 and st["pair_ticks"] > 0 and st["careful_ticks"] >= st["pair_ticks"]
    # the exact ticks of a scene this size enqueue their solve before the host has the counts; nearly all of them stand
    assert st["careful_ticks"] >= st["speculated_ticks"] > st["careful_ticks"] // 2
    assert w.last_contact_count() == ow.n_contacts() > 0
    assert abs(w.last_residual() - ow.sor_residual()) <= 1e-6 * max(1.0, ow.sor_residual())


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_tumbling_boxes_land_and_settle(dtype):
    """Spinning boxes with dMassSetBox inertia: edge/corner landings, 1-4 contacts, bounce rule."""
    scene = pkg.scenes.box_grid(32, 32, seed=11, y_range=(0.8, 4.0), spin=True, box_mass=True).astype(dtype)
    scene.avel[:] *= 3.0
    steps = 400                  # bodies roll into one another from tick ~64 on: box-box contacts, multi-body islands
    w = _gpu_run(scene, dtype, steps)
    ow = _oracle_run(_orc(dtype), scene, steps, allow_pairs=True)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts()


def test_mixed_spheres_and_boxes_pile_up():
    """Boxes and spheres dropped into a tight cluster: sphere-sphere, sphere-box, box-box contacts and multi-body
    islands from the first ticks on (crowded safe zones -> exact mode throughout)."""
    scene = pkg.scenes.box_grid(8, 8, seed=21, y_range=(0.6, 6.0), spin=True, box_mass=True).astype("float64")
    scene.pos[:, [0, 2]] *= 0.3                      # pitch 0.75 m: neighbours overlap as they fall
    scene.gtype[::3] = pkg.scenes.GEOM_SPHERE
    scene.sides[::3, 0] = 0.1 + 0.3 * (scene.sides[::3, 0] - 0.2) / 0.8
    w = _gpu_run(scene, "float64", 240)
    from oracle.orc_ctypes import Oracle
    orc = Oracle("float64")
    ow = orc.world()
    ow.add_plane(*scene.plane)
    lib = orc.lib
    for i in range(scene.n):                         # interleaved creation keeps geom order = body order
        b = lib.orc_body_create(ow.w)
        lib.orc_body_set_position(ow.w, b, *scene.pos[i])
        _, qp = orc.arr(scene.quat[i]); lib.orc_body_set_quaternion(ow.w, b, qp)
        lib.orc_body_set_angular_vel(ow.w, b, *scene.avel[i])
        _, ip = orc.arr(np.diag(scene.inertia[i]).ravel()); lib.orc_body_set_mass(ow.w, b, scene.mass[i, 0], ip)
        g = (lib.orc_geom_create_sphere(ow.w, scene.sides[i, 0]) if scene.gtype[i] == pkg.scenes.GEOM_SPHERE
             else lib.orc_geom_create_box(ow.w, *scene.sides[i]))
        lib.orc_geom_set_category_bits(ow.w, g, 2); lib.orc_geom_set_collide_bits(ow.w, g, 3)
        lib.orc_geom_set_body(ow.w, g, b)
    ow.run(H, 240)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts() > scene.n
    assert w.collision_stats()["pair_ticks"] > 100


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_exact_tick_pipelines_agree(dtype):
    """The exact tick's bookkeeping as a launch per stage and as two one-workgroup kernels (dmxBatchSetExactPipeline): the
    same pairs, islands, joints and level schedules, so the same state bit for bit -- and the oracle's.  1 296 bodies in
    144 piles (every tick exact, ~600 body pairs, islands of ~9 bodies) and the tumbling-box scene (mostly one-body islands,
    a few pairs: what the one-workgroup form is for)."""
    piles = pkg.scenes.box_grid(36, 36, seed=5, y_range=(0.6, 6.0), spin=True, box_mass=True).astype(dtype)
    ix, iz = np.arange(piles.n) % 36, np.arange(piles.n) // 36
    piles.pos[:, 0] = (ix // 3) * 7.5 + (ix % 3) * 0.6
    piles.pos[:, 2] = (iz // 3) * 7.5 + (iz % 3) * 0.6
    tumble = pkg.scenes.box_grid(24, 24, seed=11, y_range=(0.8, 4.0), spin=True, box_mass=True).astype(dtype)
    tumble.avel[:] *= 3.0
    for scene, steps in ((piles, 90), (tumble, 240)):
        runs = {}
        for mode in (pkg.batch.EXACT_STAGED, pkg.batch.EXACT_ONE_WORKGROUP):
            w = _gpu_run(scene, dtype, steps, setup=lambda w, m=mode: w.set_exact_pipeline(m))
            runs[mode] = (w.state(), w.last_contact_count(), w.collision_stats())
        a, b = runs[pkg.batch.EXACT_STAGED], runs[pkg.batch.EXACT_ONE_WORKGROUP]
        for x, y in zip(a[0], b[0]):
            assert np.array_equal(x, y)
        assert a[1] == b[1] and a[2]["pair_ticks"] == b[2]["pair_ticks"] > 0 and a[2]["last_pairs"] == b[2]["last_pairs"]
        ow = _oracle_run(_orc(dtype), scene, steps, allow_pairs=True)
        _compare(b[0], ow.state())
        assert b[1] == ow.n_contacts() > 0


def test_thousands_of_piles():
    """36 864 bodies in 4 096 piles: every tick exact, 4 096+ multi-body islands of ~100 rows each (a wavefront per island,
    a lane per contact), the stage-per-launch bookkeeping at scale.  Bit-identical to the sequential oracle."""
    side = 64
    scene = pkg.scenes.box_grid(3 * side, 3 * side, seed=5, y_range=(0.6, 6.0), spin=True, box_mass=True).astype("float32")
    ix, iz = np.arange(scene.n) % (3 * side), np.arange(scene.n) // (3 * side)
    scene.pos[:, 0] = (ix // 3) * 7.5 + (ix % 3) * 0.6
    scene.pos[:, 2] = (iz // 3) * 7.5 + (iz % 3) * 0.6
    steps = 45
    w = _gpu_run(scene, "float32", steps)
    ow = _oracle_run(_orc("float32"), scene, steps, allow_pairs=True)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts() > 10000
    assert w.collision_stats()["last_pairs"] > 4096


def test_body_collisions_off_is_the_plain_fused_path():
    scene = pkg.scenes.box_grid(32, 32, seed=1, spin=True, plane=False).astype("float32")
    a = _gpu_run(scene, "float32", 100).state()
    b = _gpu_run(scene, "float32", 100, setup=lambda w: w.set_body_collisions(False)).state()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_config3_shape_reduced():
    """configs[2] at 64x64: drop from y in [1,3], all in contact after ~60 steps."""
    scene = pkg.scenes.config3(64).astype("float32")
    w = _gpu_run(scene, "float32", 180)
    ow = _oracle_run(_orc("float32"), scene, 180)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts() > 3 * scene.n      # most boxes rest on a face


def test_spheres_on_plane():
    scene = pkg.scenes.box_grid(24, 24, seed=2, y_range=(0.3, 2.0), spin=True).astype("float64")
    scene.sides[:, 0] = 0.1 + 0.3 * (scene.sides[:, 0] - 0.2) / 0.8      # radius in [0.1, 0.4]  main.c:516
    scene.gtype[:] = pkg.scenes.GEOM_SPHERE
    w = _gpu_run(scene, "float64", 240)
    ow = _oracle_run(_orc("float64"), scene, 240, spheres=True)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts() > 0


def test_solver_parameters_are_honoured():
    scene = pkg.scenes.box_grid(16, 16, seed=4, y_range=(0.6, 1.2), spin=False).astype("float64")

    def gs(w):
        w.set_erp(0.35); w.set_cfm(1e-6); w.set_quickstep(7, 1.1); w.set_surface(mu=0.0, bounce=0.5, bounce_vel=0.3)

    def os_(orc, ow):
        lib = orc.lib
        lib.orc_world_set_erp(ow.w, 0.35); lib.orc_world_set_cfm(ow.w, 1e-6)
        lib.orc_world_set_quickstep(ow.w, 7, 1.1)
        lib.orc_world_set_surface(ow.w, 0x004, 0.0, 0.5, 0.3)
    w = _gpu_run(scene, "float64", 120, setup=gs)
    ow = _oracle_run(_orc("float64"), scene, 120, setup=os_)
    _compare(w.state(), ow.state())


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("mu", [0.0, 0.4, float("inf")])
def test_piles_with_and_without_friction(dtype, mu):
    """Multi-body islands under the three shapes of a contact: one row (mu = 0: the island solve's row-per-lane form), three
    rows with bounded friction, three with unbounded (a lane per contact, its rows made in registers).  36 piles of 9 boxes,
    some of them more than 64 contacts (two contacts per lane in f32, the row form in f64)."""
    scene = pkg.scenes.box_grid(18, 18, seed=9, y_range=(0.6, 6.0), spin=True, box_mass=True).astype(dtype)
    ix, iz = np.arange(scene.n) % 18, np.arange(scene.n) // 18
    scene.pos[:, 0] = (ix // 3) * 7.5 + (ix % 3) * 0.6
    scene.pos[:, 2] = (iz // 3) * 7.5 + (iz % 3) * 0.6

    def gs(w):
        w.set_surface(mu=mu, bounce=0.2, bounce_vel=0.1)

    def os_(orc, ow):
        orc.lib.orc_world_set_surface(ow.w, 0x004, mu, 0.2, 0.1)
    steps = 100
    w = _gpu_run(scene, dtype, steps, setup=gs)
    ow = _oracle_run(_orc(dtype), scene, steps, setup=os_, allow_pairs=True)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts() > scene.n


def test_tilted_plane():
    scene = pkg.scenes.box_grid(16, 16, seed=6, y_range=(3.0, 5.0), spin=True, box_mass=True).astype("float64")
    scene.plane = (0.1, 1.0, -0.2, -0.5)
    steps = 200                  # landed on the slope, tumbling downhill into one another
    w = _gpu_run(scene, "float64", steps)
    ow = _oracle_run(_orc("float64"), scene, steps, allow_pairs=True)
    _compare(w.state(), ow.state())


# ----------------------------------------------------------------- pose snapshot (GetTransformMat)
def test_pack_transforms_matches_reference_layout():
    import ctypes as C
    scene = pkg.scenes.box_grid(20, 10, seed=9, spin=True, plane=False).astype("float64")
    w = _gpu_run(scene, "float64", 30)
    T = w.transforms()
    orc = _orc("float64")
    ow = _oracle_run(orc, scene, 30)
    RP = C.POINTER(C.c_double)
    out = np.zeros(16)
    for i in (0, 1, 57, 199):
        orc.lib.orc_pack_transform(out.ctypes.data_as(RP), orc.lib.orc_body_get_position(ow.w, i),
                                   orc.lib.orc_body_get_rotation(ow.w, i))
        assert np.array_equal(T[i], out)
    part = w.transforms(first=50, count=7)
    assert np.array_equal(part, T[50:57])


# ----------------------------------------------------------------- upload / download, gather / scatter
def test_upload_download_roundtrip_and_partial_ranges():
    n = 777
    rng = np.random.default_rng(0)
    w = pkg.BatchWorld(n, dtype="float32")
    B = pkg.batch
    a = rng.standard_normal((n, 3)).astype(np.float32)
    w.upload(B.POS, a)
    assert np.array_equal(w.download(B.POS), a)
    w.upload(B.LVEL, a[100:200] * 2, first=300)
    got = w.download(B.LVEL)
    assert np.array_equal(got[300:400], a[100:200] * 2) and not got[:300].any() and not got[400:].any()
    assert np.array_equal(w.download(B.LVEL, first=350, count=10), a[150:160] * 2)
    # defaults of untouched fields: m = 1, I = identity, q = identity (dBodyCreate)
    assert np.all(w.download(B.MASS) == 1) and np.all(w.download(B.INERTIA) == 1)
    assert np.array_equal(w.download(B.QUAT), np.tile([1, 0, 0, 0], (n, 1)).astype(np.float32))
    # quaternions are normalised on upload (dBodySetQuaternion)
    q = rng.standard_normal((n, 4)).astype(np.float32)
    w.upload(B.QUAT, q)
    assert np.allclose(np.linalg.norm(w.download(B.QUAT), axis=1), 1.0, atol=1e-6)


def test_state_field_is_the_four_fields_in_one_transfer():
    scene = pkg.scenes.box_grid(9, 7, seed=8, spin=True, plane=False).astype("float32")
    w = _gpu_run(scene, "float32", 11)
    st = w.download(pkg.batch.STATE)
    assert st.shape == (scene.n, 13)
    assert np.array_equal(st, np.concatenate(w.state(), axis=1))
    assert np.array_equal(w.download(pkg.batch.STATE, 5, 20), st[5:25])
    w2 = pkg.BatchWorld(scene.n, dtype="float32")
    w2.upload(pkg.batch.STATE, st)                      # stored as given (no renormalisation of the quaternion)
    assert np.array_equal(np.concatenate(w2.state(), axis=1), st)


def test_external_force_is_applied_once_and_cleared():
    B = pkg.batch
    w = pkg.BatchWorld(300, dtype="float64", gravity=(0, 0, 0))
    f = np.zeros((300, 3)); f[:, 0] = 6.0
    w.upload(B.FORCE, f)
    w.step(H, 1)
    v = w.download(B.LVEL)
    assert np.allclose(v[:, 0], 6.0 * H) and not w.download(B.FORCE).any()
    w.step(H, 3)
    assert np.allclose(w.download(B.LVEL)[:, 0], 6.0 * H)      # no further acceleration


def test_gather_scatter_bodies_roundtrip():
    import torch
    B = pkg.batch
    scene = pkg.scenes.box_grid(32, 8, seed=3, spin=True, plane=False).astype("float32")
    w = pkg.BatchWorld(scene.n, dtype="float32"); w.load_scene(scene); w.step(H, 5)
    idx = torch.tensor([0, 31, 32, 255, 100], dtype=torch.int32, device="cuda")
    buf = torch.empty((5, 13), dtype=torch.float32, device="cuda")
    assert w.lib.dmxBatchGatherBodies(w.h, idx.data_ptr(), 5, buf.data_ptr()) == 0
    w.synchronize()
    pos, quat, lvel, avel = w.state()
    ref = np.concatenate([pos, quat, lvel, avel], axis=1)[[0, 31, 32, 255, 100]]
    assert np.array_equal(buf.cpu().numpy(), ref)
    w2 = pkg.BatchWorld(scene.n, dtype="float32")
    assert w2.lib.dmxBatchScatterBodies(w2.h, idx.data_ptr(), 5, buf.data_ptr()) == 0
    w2.synchronize()
    assert np.array_equal(w2.download(B.POS)[[0, 31, 32, 255, 100]], ref[:, :3])


# ----------------------------------------------------------------- full-size properties (configs[1], configs[2])
def test_config2_full_size_properties():
    """1 048 576 free bodies: closed-form free fall, conserved horizontal motion, unit quaternions,
    and bit-parity with the oracle on a strided sample of bodies."""
    scene = pkg.scenes.config2().astype("float32")
    steps = 120
    w = _gpu_run(scene, "float32", steps)
    pos, quat, lvel, avel = w.state()
    assert np.all(np.abs(np.linalg.norm(quat.astype(np.float64), axis=1) - 1.0) < 2e-6)
    assert np.array_equal(pos[:, 0], scene.pos[:, 0]) and np.array_equal(pos[:, 2], scene.pos[:, 2])
    g, n = -9.8, steps
    assert np.allclose(lvel[:, 1], n * H * g, rtol=1e-5)
    y = scene.pos[:, 1].astype(np.float64) + g * H * H * n * (n + 1) / 2
    # closed form vs f32 state: 120 roundings of y += h*v accumulate to a few 1e-5 relative; parity with
    # the f32 oracle (below) is exact, and test_config2_closed_form_f64 pins the closed form at 1e-12
    assert np.max(np.abs(pos[:, 1] - y) / np.maximum(np.abs(y), 1.0)) < 1e-4
    # isotropic inertia: |omega| is conserved
    assert np.allclose(np.linalg.norm(avel, axis=1), np.linalg.norm(scene.avel, axis=1), rtol=1e-4)
    sel = np.arange(0, scene.n, 257)
    sub = pkg.scenes.Scene(scene.pos[sel], scene.quat[sel], scene.lvel[sel], scene.avel[sel], scene.mass[sel],
                           scene.inertia[sel], scene.sides[sel], scene.gtype[sel], None)
    ow = _oracle_run(_orc("float32"), sub, steps)
    for a, b in zip((pos, quat, lvel, avel), ow.state()):
        assert np.array_equal(a[sel], b)


def test_config2_closed_form_f64():
    """KAT-1 on the device in f64: y_n = y_0 + g h^2 n(n+1)/2, v_n = n h g, at 262 144 bodies."""
    scene = pkg.scenes.config2(512).astype("float64")
    n = 240
    w = _gpu_run(scene, "float64", n)
    pos, quat, lvel, avel = w.state()
    y = scene.pos[:, 1] + (-9.8) * H * H * n * (n + 1) / 2
    assert np.max(np.abs(pos[:, 1] - y) / np.maximum(np.abs(y), 1.0)) < 1e-12
    assert np.allclose(lvel[:, 1], n * H * -9.8, rtol=1e-13)
    assert np.max(np.abs(np.linalg.norm(quat, axis=1) - 1.0)) < 1e-15


def test_config3_full_size_properties():
    """262 144 boxes onto the plane: everything lands, rests with 4 contacts, no penetration runaway."""
    scene = pkg.scenes.config3().astype("float32")
    w = _gpu_run(scene, "float32", 240)
    pos, quat, lvel, avel = w.state()
    assert 3 * scene.n < w.last_contact_count() <= 4 * scene.n
    assert np.all(np.abs(np.linalg.norm(quat.astype(np.float64), axis=1) - 1.0) < 2e-6)
    rest = scene.sides[:, 1] / 2
    # the bulk rests flat at half the vertical side; a few thin boxes are still toppling
    assert np.mean(np.abs(pos[:, 1] - rest) < 5e-3) > 0.9
    assert pos[:, 1].min() > 0.05 and pos[:, 1].max() < 1.5 and np.all(np.isfinite(pos))
    assert np.array_equal(pos[:, [0, 2]][np.abs(avel).max(axis=1) == 0], scene.pos[:, [0, 2]][np.abs(avel).max(axis=1) == 0])
    sel = np.arange(0, scene.n, 521)
    sub = pkg.scenes.Scene(scene.pos[sel], scene.quat[sel], scene.lvel[sel], scene.avel[sel], scene.mass[sel],
                           scene.inertia[sel], scene.sides[sel], scene.gtype[sel], scene.plane)
    ow = _oracle_run(_orc("float32"), sub, 240)
    for a, b in zip((pos, quat, lvel, avel), ow.state()):
        assert np.array_equal(a[sel], b)


# ----------------------------------------------------------------- island sharding (configs[3])
@pytest.mark.parametrize("plane", [False, True])
def test_boundary_first_range_stepping_equals_whole_step(plane):
    """The sharded tick (boundary rows, then interior, dmxBatchStepRange) is the same tick."""
    scene = pkg.scenes.box_grid(64, 8, seed=8, y_range=(0.7, 3.0), spin=True, box_mass=True, plane=plane).astype("float32")
    L = pkg.shard.SlabLayout(64, 8)
    a = _gpu_run(scene, "float32", 60)
    b = pkg.BatchWorld(L.n_total, dtype="float32")       # with ghost slots behind the active bodies
    b.load_scene(scene)
    b.set_active_count(scene.n)
    first, count = L.interior
    for _ in range(60):
        b.step_range(H, 0, L.side, reset_diag=True)
        b.step_range(H, L.n - L.side, L.side)
        b.step_range(H, first, count)
    b.synchronize()
    for x, y in zip(a.state(), b.state()):
        assert np.array_equal(x, y[:scene.n])
    assert a.last_contact_count() == b.last_contact_count()
    # ghost slots were never stepped: still the defaults of dBodyCreate
    assert not b.download(pkg.batch.POS, first=scene.n).any()
    # ranges must start on a wave (64 bodies) and respect the 16-byte packs
    with pytest.raises(pkg.batch.DmxError):
        b.step_range(H, 2, 8)
    with pytest.raises(pkg.batch.DmxError):
        b.step_range(H, 64, 6)


def test_slabs_of_config4_step_independently():
    """configs[3] layout: the scene split into disjoint slabs gives the same bodies whether stepped as one
    batch or one batch per slab (what each GPU does)."""
    slabs, side = 4, 32
    scene = pkg.scenes.box_grid(side, side, seed=1, spin=True, plane=False, slabs=slabs, slab_gap=10.0).astype("float64")
    whole = _gpu_run(scene, "float64", 100).state()
    per = scene.n // slabs
    for r in range(slabs):
        part = _gpu_run(scene.slice(r * per, (r + 1) * per), "float64", 100).state()
        for x, y in zip(whole, part):
            assert np.array_equal(x[r * per:(r + 1) * per], y)


def test_fused_boundary_pack_matches_state():
    """dmxBatchSetBoundaryPack: the step kernels drop the boundary rows' new state, packed, into the send buffer."""
    import torch
    for plane, dtype, tdt in ((False, "float32", torch.float32), (True, "float64", torch.float64)):
        scene = pkg.scenes.box_grid(64, 8, seed=9, y_range=(0.7, 3.0), spin=True, box_mass=True, plane=plane).astype(dtype)
        L = pkg.shard.SlabLayout(64, 8)
        w = pkg.BatchWorld(L.n_total, dtype=dtype)
        w.load_scene(scene)
        w.set_active_count(scene.n)
        w.set_body_collisions(False)
        buf = torch.full((L.n_send, 13), -7.0, dtype=tdt, device="cuda")
        w.set_boundary_pack(buf.data_ptr(), L.side, L.n - L.side)
        w.step(H, 25)
        w.synchronize()
        st = np.concatenate(w.state(), axis=1)
        assert np.array_equal(buf.cpu().numpy(), st[L.send_idx])
        w.set_boundary_pack(None, 0, 0)
        w.step(H, 1); w.synchronize()
        assert np.array_equal(buf.cpu().numpy(), st[L.send_idx])          # untouched once the pack is off


def test_midair_collisions_in_free_flight():
    """No ground plane, gravity along y: the safe-zone proof is checked at the first and last tick of a chunk only
    (straight-line horizontal motion).  Bodies flying sideways into one another must still be caught and resolved."""
    scene = pkg.scenes.box_grid(16, 16, seed=13, y_range=(10.0, 12.0), spin=True, box_mass=True, plane=False).astype("float64")
    rng = np.random.default_rng(5)
    scene.lvel[:, 0] = rng.uniform(-3.0, 3.0, scene.n)        # up to 6 m/s closing speed at 2.5 m pitch
    scene.lvel[:, 2] = rng.uniform(-3.0, 3.0, scene.n)
    w = _gpu_run(scene, "float64", 150)
    ow = _oracle_build(_orc("float64"), scene)
    pairs = 0
    for _ in range(150):
        ow.tick(H)
        pairs += ow.n_body_pairs()
    assert pairs > 50                                           # plenty of mid-air box-box encounters
    _compare(w.state(), ow.state())
    st = w.collision_stats()
    assert st["pair_ticks"] > 0 and st["careful_ticks"] >= st["pair_ticks"]


# ----------------------------------------------------------------- sharded loop with the collision proof
def _one_rank_group():
    import socket
    import torch.distributed as dist
    import os
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    return dist


@pytest.mark.parametrize("graph_steps,speed,lazy", [(0, 3.0, False), (8, 3.0, False), (8, 0.1, False), (0, 0.1, False),
                                                    (0, 3.0, True), (0, 0.1, True)])
def test_sharded_loop_carries_the_collision_proof(graph_steps, speed, lazy):
    """shard.ShardedStepper(collide=True) on a one-rank group (the collective degenerates to a copy): chunks, ghost
    checks, collective rollback and exact replay give the oracle's bits, mid-air box-box contacts included."""
    import torch
    scene = pkg.scenes.box_grid(16, 16, seed=13, y_range=(10.0, 12.0), spin=True, box_mass=True, plane=False).astype("float64")
    rng = np.random.default_rng(5)
    scene.lvel[:, 0] = rng.uniform(-speed, speed, scene.n)
    scene.lvel[:, 2] = rng.uniform(-speed, speed, scene.n)
    steps = 150
    ow = _oracle_build(_orc("float64"), scene)
    ow.run(H, steps)
    dist = _one_rank_group()
    try:
        L = pkg.shard.SlabLayout(16, 16)
        w = pkg.BatchWorld(L.n_total, dtype="float64")
        w.load_scene(scene)
        w.set_active_count(scene.n)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            w.set_stream(stream.cuda_stream)
            ops = pkg.shard.DeviceOps(w, torch.device("cuda", 0), stream)
            st = pkg.shard.ShardedStepper(w, L, 0, 1, collide=True, geometry=(scene.sides, scene.gtype), ops=ops, lazy=lazy)
            if graph_steps:
                st.capture(H, graph_steps, stream)
            if lazy:                              # what bench.py does: a few ticks per call, chunks closed lazily, close() at the end
                for k in range(0, steps, 7):
                    st.run(H, min(7, steps - k))
                st.close()
            else:
                st.run(H, steps)
            st.drain()
            w.synchronize()
            got = [a[:scene.n] for a in w.state()]
            stats = w.collision_stats()
        _compare(got, ow.state())
        assert stats["fast_ticks"] + stats["careful_ticks"] >= steps
        if speed > 1.0:
            assert stats["pair_ticks"] > 0            # fast sideways motion: most chunks end in an exact replay
        else:
            assert stats["fast_ticks"] >= 64          # slow drift: quiet chunks commit on the fast path
        w.close()
    finally:
        dist.destroy_process_group()


def test_ghost_slots_are_checked_and_cross_rank_pairs_are_reported():
    import torch
    scene = pkg.scenes.box_grid(8, 4, seed=3, y_range=(5.0, 6.0), spin=False, plane=False).astype("float32")
    L = pkg.shard.SlabLayout(8, 4)
    w = pkg.BatchWorld(L.n_total, dtype="float32")
    w.load_scene(scene)
    # ghosts: a row of boxes one pitch beyond the slab's last row (what the upper neighbour would send)
    g = scene.slice(scene.n - 8, scene.n)
    gpos = g.pos.copy(); gpos[:, 2] += 2.5
    first = int(L.ghost_hi[0])
    w.upload(pkg.batch.POS, gpos, first=first)
    w.upload(pkg.batch.SIDES, g.sides, first=first)
    w.upload_geom_type(g.gtype, first=first)
    w.set_active_count(scene.n)
    assert w.chunk_begin() == (False, True)
    s = torch.cuda.Stream()
    w.check_zones_on(s.cuda_stream, scene.n, 2 * L.side)
    s.synchronize()
    assert w.chunk_end() == (False, False)
    w.chunk_tick(H, True)                   # the chunk's first tick (from here on a rollback undoes what the exchange writes)
    # a ghost arrives 1 m closer than where its zone was built: outside the zone
    state = np.concatenate([gpos, g.quat, g.lvel, g.avel], axis=1).astype(np.float32)
    state[3, 2] -= 1.0
    idx = torch.arange(first, first + 8, dtype=torch.int32, device="cuda")
    src = torch.from_numpy(state).cuda()
    w.scatter_bodies(idx.data_ptr(), 8, src.data_ptr())
    w.synchronize()
    w.check_zones_on(s.cuda_stream, scene.n, 2 * L.side)
    s.synchronize()
    violated, _ = w.chunk_end()
    assert violated
    # the fused per-tick refresh: rows back where the zones were built -> clean; one row displaced -> violation
    w.chunk_rollback(); w.chunk_begin()
    good = np.concatenate([gpos, g.quat, g.lvel, g.avel], axis=1).astype(np.float32)
    src = torch.from_numpy(good).cuda()
    w.refresh_ghosts_on(s.cuda_stream, scene.n, L.side, None, L.side, src.data_ptr(), True)
    s.synchronize()
    assert w.chunk_end() == (False, False)
    assert np.array_equal(w.download(pkg.batch.POS, first, 8), gpos.astype(np.float32))
    assert np.all(w.download(pkg.batch.POS, scene.n, 8) == 0)                  # the lower range had no source: untouched
    bad = good.copy(); bad[5, 0] += 1.0
    src = torch.from_numpy(bad).cuda()
    w.refresh_ghosts_on(s.cuda_stream, scene.n, L.side, None, L.side, src.data_ptr(), True)
    s.synchronize()
    assert w.chunk_end()[0]
    # and one that overlaps a body of this rank: the exact tick refuses (island spanning two ranks)
    state[3, 2] = scene.pos[scene.n - 8 + 3, 2] + 0.1
    state[3, 1] = scene.pos[scene.n - 8 + 3, 1]
    src = torch.from_numpy(state).cuda()
    w.scatter_bodies(idx.data_ptr(), 8, src.data_ptr())
    w.synchronize()
    with pytest.raises(pkg.batch.DmxError, match="-6"):
        w.exact_tick(H)
    w.close()


# ----------------------------------------------------------------- two ranks on one GPU (collectives staged through gloo)
def _two_rank_worker(rank, port, steps, speed, out_q):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        from __graft_entry__ import load_package
        p = load_package()

        nx, rows = 16, 8
        full = _two_rank_scene(p, nx, rows, speed)
        scene = full.slice(rank * nx * rows, (rank + 1) * nx * rows)
        L = p.shard.SlabLayout(nx, rows)
        w = p.BatchWorld(L.n_total, dtype="float64")
        w.load_scene(scene)
        w.set_active_count(scene.n)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            w.set_stream(stream.cuda_stream)
            ops = p.shard.StagedDeviceOps(w, torch.device("cuda", 0), stream)
            st = p.shard.ShardedStepper(w, L, rank, 2, collide=True, geometry=(scene.sides, scene.gtype), ops=ops)
            st.run(H, steps)
            st.drain()
            w.synchronize()
            state = [a.copy() for a in w.state()]
            stats = w.collision_stats()
        out_q.put((rank, state, stats, st.exchange.count))
        w.close()
    finally:
        dist.destroy_process_group()


def _two_rank_scene(p, nx, rows, speed):
    scene = p.scenes.box_grid(nx, 2 * rows, seed=21, y_range=(10.0, 12.0), spin=True, box_mass=True, plane=False).astype("float64")
    rng = np.random.default_rng(7)
    scene.lvel[:, 0] = rng.uniform(-speed, speed, scene.n)
    scene.lvel[:, 2] = rng.uniform(-speed, speed, scene.n)
    # keep the four rows either side of the shared face slow (a fast body covers 6 m = 2.4 rows in the run): bodies may
    # collide inside a slab, never across the face
    z_row = np.arange(scene.n) // nx
    near_face = (z_row >= rows - 4) & (z_row < rows + 4)
    scene.lvel[near_face, 0] *= 0.02
    scene.lvel[near_face, 2] *= 0.02
    return scene


@pytest.mark.parametrize("speed", [0.1, 1.0, 3.0])
def test_two_ranks_share_the_gpu_and_match_the_unsharded_oracle(speed):
    """Two processes, one slab each, one GPU: the whole N>1 loop (geometry sharing, boundary pack, ghost refresh and
    ghost zone checks, collective rollback, exact replay with per-tick exchange) against the oracle stepping the full
    scene in one world."""
    import socket
    import torch.multiprocessing as mp
    steps, nx, rows = 120, 16, 8
    full = _two_rank_scene(pkg, nx, rows, speed)
    ow = _oracle_build(_orc("float64"), full)
    pairs = 0
    for _ in range(steps):
        ow.tick(H)
        pairs += ow.n_body_pairs()
    ref = ow.state()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, port, steps, speed, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = {}
    try:
        for _ in range(2):
            r, state, stats, n_ex = q.get(timeout=150)
            got[r] = (state, stats, n_ex)
        for pr in procs:
            pr.join(timeout=60)
            assert pr.exitcode == 0
    finally:
        for pr in procs:
            if pr.is_alive():
                pr.terminate()
    L = pkg.shard.SlabLayout(nx, rows)
    for r in range(2):
        state, stats, n_ex = got[r]
        lo, hi = r * L.n, (r + 1) * L.n
        _compare([a[:L.n] for a in state], [a[lo:hi] for a in ref])
        # the ghost rows hold the neighbour's boundary row as of the last tick
        other = 1 - r
        ghost = L.ghost_hi if r == 0 else L.ghost_lo
        src = (L.lower if r == 0 else L.upper) + other * L.n
        _compare([a[ghost] for a in state], [a[src] for a in ref])
        assert stats["fast_ticks"] + stats["careful_ticks"] >= steps
    assert got[0][2] == got[1][2]                          # both ranks issued the same number of exchanges
    if speed > 1.0:
        assert pairs > 0 and got[0][1]["pair_ticks"] + got[1][1]["pair_ticks"] > 0
        assert got[0][2] > 4                               # rolled-back chunks replay with an exchange every tick
    elif speed < 0.5:
        # quiet chunks exchange once, at their end; a chunk that ends in an exact replay exchanges every tick
        assert got[0][1]["fast_ticks"] >= 64 and got[0][2] == got[1][2] < steps // 2


# ----------------------------------------------------------------- convex hulls (BASELINE configs[4])
def _teapot_hull():
    import os
    from __graft_entry__ import ROOT
    gold = np.load(os.path.join(ROOT, "tests", "golden", "teapot_hull.npz"))
    return pkg.hull.build(gold["points"], 0.01)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_teapot_hulls_dropping_on_the_plane_match_oracle(dtype):
    """64 teapot hulls (1 265 points each), tilted and spinning, dropped on the ground plane: np_convex_plane (one
    wavefront per hull) + the 8-slot fused step against the oracle's sequential dCollideConvexPlane + QuickStep."""
    hull = _teapot_hull()
    scene = pkg.scenes.hull_grid(hull, 8, 8, seed=4, y_range=(0.8, 2.5), spin=True, tilt=0.6).astype(dtype)
    steps = 240
    w = _gpu_run(scene, dtype, steps)
    ow = _oracle_run(_orc(dtype), scene, steps, allow_pairs=True)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts() > 0
    y = w.state()[0][:, 1]
    # the hulls are held by the plane (dCollideConvexPlane keeps the FIRST eight penetrating points in array order, so a
    # tipped hull can be propped on one side and sag on the other -- ODE's behaviour, reproduced by oracle and kernel alike)
    assert 0.3 < np.median(y) < 0.7 and np.all(y > -0.5) and np.all(y < 1.5)


def test_boxes_and_hulls_share_a_batch():
    hull = _teapot_hull()
    hs = pkg.scenes.hull_grid(hull, 8, 4, seed=6, y_range=(0.8, 2.0), spin=True, tilt=0.3)
    bs = pkg.scenes.box_grid(8, 4, seed=7, y_range=(0.8, 2.0), spin=True, box_mass=True, plane=True)
    bs.pos[:, 2] -= 20.0                                        # the boxes' rows well clear of the hulls' rows
    cat = lambda a, b: np.concatenate([a, b])
    scene = pkg.scenes.Scene(cat(bs.pos, hs.pos), cat(bs.quat, hs.quat), cat(bs.lvel, hs.lvel), cat(bs.avel, hs.avel),
                             cat(bs.mass, hs.mass), cat(bs.inertia, hs.inertia), cat(bs.sides, hs.sides),
                             cat(bs.gtype, hs.gtype), bs.plane, hs.hull_points).astype("float64")
    steps = 200
    w = _gpu_run(scene, "float64", steps)
    ow = _oracle_run(_orc("float64"), scene, steps, allow_pairs=True)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts() > 0


# ----------------------------------------------------------------- several ticks per launch
@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("ticks", [2, 5, 8])
def test_ticks_per_launch_changes_nothing_but_the_launch_count(dtype, ticks):
    """dmxBatchSetTicksPerLaunch: contact-free ticks fused into one integrate_free launch give the bits of one launch per
    tick -- with the collision proof riding along (chunks of 32+, not multiples of `ticks`) and without it."""
    scene = pkg.scenes.box_grid(40, 25, seed=17, spin=True, box_mass=True, plane=False).astype(dtype)
    steps = 77
    ref = _gpu_run(scene, dtype, steps)
    for collide in (True, False):
        w = _gpu_run(scene, dtype, steps, setup=lambda w: (w.set_ticks_per_launch(ticks), w.set_body_collisions(collide)))
        _compare(w.state(), ref.state())
        if collide:
            assert w.collision_stats()["fast_ticks"] == steps
    ow = _oracle_run(_orc(dtype), scene, steps)
    _compare(ref.state(), ow.state())


def test_fused_ticks_still_catch_midair_collisions():
    scene = pkg.scenes.box_grid(16, 16, seed=13, y_range=(10.0, 12.0), spin=True, box_mass=True, plane=False).astype("float64")
    rng = np.random.default_rng(5)
    scene.lvel[:, 0] = rng.uniform(-3.0, 3.0, scene.n)
    scene.lvel[:, 2] = rng.uniform(-3.0, 3.0, scene.n)
    w = _gpu_run(scene, "float64", 150, setup=lambda w: w.set_ticks_per_launch(8))
    ow = _oracle_run(_orc("float64"), scene, 150, allow_pairs=True)
    _compare(w.state(), ow.state())
    assert w.collision_stats()["pair_ticks"] > 0


def test_fused_ticks_apply_external_force_once():
    scene = pkg.scenes.box_grid(8, 8, seed=2, spin=True, plane=False).astype("float64")
    f = np.zeros((scene.n, 3)); f[:, 0] = 3.0
    outs = []
    for ticks in (1, 6):
        w = pkg.BatchWorld(scene.n, dtype="float64")
        w.load_scene(scene)
        w.set_body_collisions(False)
        w.set_ticks_per_launch(ticks)
        w.upload(pkg.batch.FORCE, f)
        w.step(H, 13)
        w.synchronize()
        outs.append(w.state())
    _compare(outs[1], outs[0])
    assert np.allclose(outs[0][2][:, 0], 3.0 * H)          # one tick's worth of impulse (m = 1), whatever the fusion


def test_elongated_scene_falls_back_to_the_scrambled_broadphase_table():
    """4 096 bodies in one row: the torus-addressed broadphase table would wrap the row onto itself dozens of times
    (bucket overflow), so the batch switches to the scrambled hash; results and the fast path are unaffected."""
    scene = pkg.scenes.box_grid(4096, 1, seed=31, spin=True, plane=False).astype("float32")
    w = _gpu_run(scene, "float32", 64)
    ow = _oracle_run(_orc("float32"), scene, 64)
    _compare(w.state(), ow.state())
    assert w.collision_stats()["fast_ticks"] == 64


def test_checkpoint_and_resume_continue_bit_for_bit():
    """BatchWorld.checkpoint / restore: a run resumed from a checkpoint (in the same batch after more stepping, or in a
    fresh batch) lands on the bits of the uninterrupted run -- plane contacts, a mid-air collision phase and chunk
    boundaries that differ between the runs notwithstanding."""
    for plane in (True, False):
        scene = pkg.scenes.box_grid(24, 20, seed=23, y_range=(0.7, 9.0), spin=True, box_mass=True, plane=plane).astype("float32")
        if not plane:
            rng = np.random.default_rng(2)
            scene.lvel[:, 0] = rng.uniform(-1.5, 1.5, scene.n).astype(np.float32)
        w = pkg.BatchWorld(scene.n, dtype="float32")
        w.load_scene(scene)
        w.step(H, 70)
        ck = w.checkpoint()
        w.step(H, 90); w.synchronize()
        ref = w.state()
        w.step(H, 13)                                       # wander off, then come back
        w.upload(pkg.batch.FORCE, np.full((scene.n, 3), 5.0, np.float32))     # dBodyAddForce since the checkpoint: must not survive the restore
        w.restore(ck)
        w.step(H, 90); w.synchronize()
        _compare(w.state(), ref)
        w2 = pkg.BatchWorld(scene.n, dtype="float32")
        w2.load_scene(scene)                                # geometry types and the plane come from the scene
        w2.restore(ck)
        w2.step(H, 45); w2.step(H, 45); w2.synchronize()
        _compare(w2.state(), ref)
        w.close(); w2.close()


# ----------------------------------------------------------------- ping-pong snapshot, lazily closed chunks
def _midair_scene(dtype="float64", speed=3.0):
    scene = pkg.scenes.box_grid(16, 16, seed=13, y_range=(10.0, 12.0), spin=True, box_mass=True, plane=False).astype(dtype)
    rng = np.random.default_rng(5)
    scene.lvel[:, 0] = rng.uniform(-speed, speed, scene.n)
    scene.lvel[:, 2] = rng.uniform(-speed, speed, scene.n)
    return scene


@pytest.mark.parametrize("per_call", [1, 2, 7, 20, 150])
@pytest.mark.parametrize("speed", [0.1, 3.0])
def test_step_calls_of_any_length_give_the_oracles_state(per_call, speed):
    """dmxBatchStep leaves its collision-proof chunk open across calls and validates it later; a violation found then
    rolls back ticks of EARLIER calls and replays them.  What a caller can observe must not depend on how the ticks
    were split over calls (the reference issues a tick or two per frame, main.c:211)."""
    scene = _midair_scene(speed=speed)
    steps = 150
    ow = _oracle_build(_orc("float64"), scene)
    ow.run(H, steps)
    w = pkg.BatchWorld(scene.n, dtype="float64")
    w.load_scene(scene)
    done = 0
    while done < steps:
        k = min(per_call, steps - done)
        w.step(H, k)
        done += k
    _compare(w.state(), ow.state())
    st = w.collision_stats()
    assert st["fast_ticks"] + st["careful_ticks"] >= steps
    if speed > 1.0:
        assert st["pair_ticks"] > 0
    w.close()


def test_observations_between_calls_see_validated_state():
    """every entry point that reads the batch settles the open chunk first: poses read after each call equal the oracle's"""
    scene = _midair_scene(speed=3.0)
    ow = _oracle_build(_orc("float64"), scene)
    w = pkg.BatchWorld(scene.n, dtype="float64")
    w.load_scene(scene)
    for k in range(60):
        w.step(H, 2)
        ow.run(H, 2)
        if k % 5 == 4:
            assert np.array_equal(w.download(pkg.batch.POS), ow.state()[0]), k
    _compare(w.state(), ow.state())
    w.close()


@pytest.mark.parametrize("mode", ["pingpong", "copy"])
def test_snapshot_modes_roll_back_to_the_same_state(mode):
    scene = _midair_scene(speed=3.0)
    ow = _oracle_build(_orc("float64"), scene)
    ow.run(H, 120)
    w = pkg.BatchWorld(scene.n, dtype="float64")
    w.set_snapshot_mode(pkg.batch.SNAPSHOT_COPY if mode == "copy" else pkg.batch.SNAPSHOT_PINGPONG)
    w.load_scene(scene)
    w.step(H, 120)
    _compare(w.state(), ow.state())
    w.close()


def test_constants_reach_both_slabs():
    """the state ping-pongs between two slabs; constants (mass, inertia, extents) and safe zones must be the same in both"""
    scene = pkg.scenes.box_grid(16, 8, seed=2, spin=True, box_mass=True, plane=False).astype("float32")
    w = pkg.BatchWorld(scene.n, dtype="float32")
    w.load_scene(scene)
    for chunks in range(3):                    # after every closed chunk the state sits in the other slab
        w.step(H, 32)
        assert np.array_equal(w.download(pkg.batch.SIDES), scene.sides)
        assert np.array_equal(w.download(pkg.batch.MASS), scene.mass)
        assert np.array_equal(w.download(pkg.batch.INERTIA), scene.inertia)
    heavier = scene.mass * 2
    w.upload(pkg.batch.MASS, heavier)
    w.step(H, 32)
    assert np.array_equal(w.download(pkg.batch.MASS), heavier)
    w.step(H, 32)
    assert np.array_equal(w.download(pkg.batch.MASS), heavier)
    # and the whole run equals the oracle's (mass does not enter free flight with gravity only; inertia does through the gyro term)
    ow = _oracle_run(_orc("float32"), scene, 160)
    _compare(w.state(), ow.state())
    w.close()


def test_chunk_api_rollback_restores_own_and_ghost_slots():
    """ChunkBegin -> ticks -> ChunkRollback through the C ABI: the start state comes back bit for bit, ghost slots included"""
    scene = pkg.scenes.box_grid(8, 4, seed=3, y_range=(5.0, 6.0), spin=True, plane=False).astype("float32")
    L = pkg.shard.SlabLayout(8, 4)
    w = pkg.BatchWorld(L.n_total, dtype="float32")
    w.load_scene(scene)
    g = scene.slice(scene.n - 8, scene.n)
    gpos = g.pos.copy(); gpos[:, 2] += 2.5
    first = int(L.ghost_hi[0])
    w.upload(pkg.batch.POS, gpos, first=first)
    w.upload(pkg.batch.SIDES, g.sides, first=first)
    w.upload_geom_type(g.gtype, first=first)
    w.set_active_count(scene.n)
    before = w.download(pkg.batch.STATE)
    assert w.chunk_begin() == (False, True)
    w.chunk_ticks(H, 5, True, True)
    w.chunk_tick(H, True)
    moved = w.download(pkg.batch.STATE)
    assert not np.array_equal(moved[:scene.n], before[:scene.n])
    assert np.array_equal(moved[scene.n:], before[scene.n:])            # ghost slots follow the state into the other slab
    assert w.chunk_end() == (False, False)
    w.chunk_rollback()
    assert np.array_equal(w.download(pkg.batch.STATE), before)
    # and a committed chunk keeps going from where it is
    assert w.chunk_begin() == (False, True)
    w.chunk_ticks(H, 6, True, True)
    assert w.chunk_end() == (False, False)
    w.chunk_commit(6)
    assert np.array_equal(w.download(pkg.batch.STATE), moved)
    w.close()


# ----------------------------------------------------------------- static box geoms in the batch path (AddBodyMap, main.c:735-761)
def _oracle_with_map(orc, scene, boxes, spheres_from=None):
    """plane (if any), then the static boxes, then the bodies: the reference's creation order (main.c:115-121, then AddBody)"""
    ow = orc.world()
    if scene.plane is not None:
        ow.add_plane(*scene.plane)
    for sides, pos, R12 in boxes:
        ow.add_static_box(sides, pos, R12)
    nb = scene.n if spheres_from is None else spheres_from
    if nb:
        ow.add_boxes(scene.pos[:nb], scene.quat[:nb], scene.lvel[:nb], scene.avel[:nb], scene.mass[:nb, 0], scene.inertia[:nb], scene.sides[:nb])
    if nb < scene.n:
        ow.add_spheres(scene.pos[nb:], scene.quat[nb:], scene.lvel[nb:], scene.avel[nb:], scene.mass[nb:, 0], scene.inertia[nb:], scene.sides[nb:, 0])
    return ow


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_reference_pen_in_the_batch_path(dtype):
    """The reference's own scene -- floor + three walls as static boxes (main.c:115-121), boxes and spheres spawned above
    them as the key-M spawner draws them (main.c:502-521) -- through the batch path: bit-identical to the oracle while the
    bodies fall, hit the floor, the walls and one another."""
    spawn = pkg.scenes.reference_spawn(96, seed=7, y_range=(3.0, 12.0))
    spawn.sort(key=lambda s: -s[0])                      # boxes (type 2) first, spheres behind them
    n = len(spawn)
    nb = sum(1 for s in spawn if s[0] == pkg.scenes.GEOM_BOX)
    sc = pkg.scenes.Scene(np.array([s[2] for s in spawn], float), np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros((n, 3)), np.zeros((n, 3)),
                          np.ones((n, 1)), np.ones((n, 3)), np.array([s[1] for s in spawn], float),
                          np.array([s[0] for s in spawn], np.uint8), None).astype(dtype)
    boxes = pkg.scenes.reference_map()
    steps = 240
    ow = _oracle_with_map(_orc(dtype), sc, boxes, spheres_from=nb)
    contacts = 0
    for _ in range(steps):
        ow.tick(H)
        contacts = max(contacts, ow.n_contacts())
    assert contacts > 50
    w = pkg.BatchWorld(n, dtype=dtype)
    w.load_scene(sc)
    w.set_static_boxes(boxes)
    w.step(H, steps)
    _compare(w.state(), ow.state())
    assert w.state()[0][:, 1].min() > 0.4              # nobody fell through the floor (its top is at y = 0.5)
    st = w.collision_stats()
    assert st["careful_ticks"] > 0
    w.close()


def test_static_floor_and_ground_plane_together():
    """a ground plane AND a static box above it: plane contacts come first in creation order, then the static boxes'"""
    scene = pkg.scenes.box_grid(12, 12, seed=9, y_range=(1.2, 2.5), spin=True, box_mass=True, plane=True).astype("float64")
    slab = [((20.0, 0.5, 8.0), (0.0, 0.25, 0.0), pkg.scenes._rot_z(0.05))]     # a tilted plank across the middle rows
    ow = _oracle_with_map(_orc("float64"), scene, slab)
    ow.run(H, 150)
    w = pkg.BatchWorld(scene.n, dtype="float64")
    w.load_scene(scene)
    w.set_static_boxes(slab)
    w.step(H, 150)
    _compare(w.state(), ow.state())
    w.close()


def test_hundred_thousand_boxes_on_a_static_floor():
    """102 400 boxes dropping onto one static floor box: every body is stepped by the exact path (device pair search,
    device narrowphase against the floor, one-body islands) -- bit-identical to the oracle."""
    side = 320
    scene = pkg.scenes.box_grid(side, side, seed=4, y_range=(1.2, 2.0), spin=False, plane=False).astype("float32")
    floor = [((1000.0, 1.0, 1000.0), (0.0, 0.0, 0.0), pkg.scenes._rot_z(0.0))]
    steps = 45
    ow = _oracle_with_map(_orc("float32"), scene, floor)
    ow.run(H, steps)
    assert ow.n_contacts() > scene.n                      # landed
    w = pkg.BatchWorld(scene.n, dtype="float32")
    w.load_scene(scene)
    w.set_static_boxes(floor)
    w.step(H, steps)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts()
    w.close()


# ----------------------------------------------------------------- convex hulls against boxes (BASELINE configs[4] as written)
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_teapot_hulls_dropping_on_a_static_box_floor_match_oracle(dtype):
    """configs[4] as BASELINE states it, reduced: teapot hulls dropping on a static BOX floor (the reference's floor is one,
    main.c:115), box-convex contacts from the wave-per-pair collider, bit-identical to the oracle's sequential restatement"""
    hull = _teapot_hull()
    scene = pkg.scenes.hull_grid(hull, 8, 8, seed=4, y_range=(0.8, 2.5), spin=True, tilt=0.6, floor_box=True).astype(dtype)
    assert scene.plane is None and len(scene.static_boxes) == 1 and scene.hull_planes.shape[1] == 4
    steps = 240
    w = _gpu_run(scene, dtype, steps)
    ow = _oracle_run(_orc(dtype), scene, steps, allow_pairs=True)
    _compare(w.state(), ow.state())
    assert w.last_contact_count() == ow.n_contacts() > 64
    y = w.state()[0][:, 1]
    assert 0.3 < np.median(y) < 0.7 and np.all(y > -0.5) and np.all(y < 1.5)          # held by the floor's top face (y = 0)
    # spinning, tilted hulls skid across the floor and some come within reach of a neighbour: hull-hull has no collider
    # (in ODE terms: not built here; the oracle returns no contact either) and the library says so instead of staying silent
    assert w.collision_stats()["unsupported_pairs"] >= 0
    w.close()


def test_boxes_dropped_onto_hulls():
    """box bodies falling onto teapot hulls that rest on the ground plane: hull-box BODY pairs in both orders (box first /
    hull first in slot order), corners of small boxes inside the hulls and hull vertices inside the boxes"""
    hull = _teapot_hull()
    hs = pkg.scenes.hull_grid(hull, 4, 4, seed=6, y_range=(0.45, 0.55), spin=False, tilt=0.0)
    bs = pkg.scenes.box_grid(4, 4, seed=7, y_range=(1.6, 2.4), spin=True, box_mass=True, plane=True)
    bs.pos[:, 0] = hs.pos[:, 0] + 0.1
    bs.pos[:, 2] = hs.pos[:, 2] - 0.05
    half = 8
    cat = lambda *a: np.concatenate(a)
    # slot order: 8 boxes, 16 hulls, 8 boxes -- so both (box, hull) and (hull, box) pairs occur.  The oracle helper wants
    # boxes first, hulls behind: give the LAST eight boxes their own world order by running two layouts is not needed --
    # the oracle adds bodies in slot order below.
    scene = pkg.scenes.Scene(cat(bs.pos[:half], hs.pos, bs.pos[half:]), cat(bs.quat[:half], hs.quat, bs.quat[half:]),
                             cat(bs.lvel[:half], hs.lvel, bs.lvel[half:]), cat(bs.avel[:half], hs.avel, bs.avel[half:]),
                             cat(bs.mass[:half], hs.mass, bs.mass[half:]), cat(bs.inertia[:half], hs.inertia, bs.inertia[half:]),
                             cat(bs.sides[:half], hs.sides, bs.sides[half:]), cat(bs.gtype[:half], hs.gtype, bs.gtype[half:]),
                             bs.plane, hs.hull_points, hs.hull_planes).astype("float64")
    orc = _orc("float64")
    ow = orc.world()
    ow.add_plane(*scene.plane)
    ow.set_hull(scene.hull_points); ow.set_hull_faces(scene.hull_planes)
    a, b = half, half + hs.n
    ow.add_boxes(scene.pos[:a], scene.quat[:a], scene.lvel[:a], scene.avel[:a], scene.mass[:a, 0], scene.inertia[:a], scene.sides[:a])
    ow.add_convex(scene.pos[a:b], scene.quat[a:b], scene.lvel[a:b], scene.avel[a:b], scene.mass[a:b, 0], scene.inertia[a:b])
    ow.add_boxes(scene.pos[b:], scene.quat[b:], scene.lvel[b:], scene.avel[b:], scene.mass[b:, 0], scene.inertia[b:], scene.sides[b:])
    steps, hull_box_contacts = 150, 0
    for _ in range(steps):
        ow.tick(H)
        hull_box_contacts += sum(1 for b1, b2, *_ in ow.joints() if b2 >= 0 and ((a <= b1 < b) != (a <= b2 < b)))
    assert hull_box_contacts > 200
    w = _gpu_run(scene, "float64", steps)
    _compare(w.state(), ow.state())
    w.close()


def test_config5_full_size_teapots_on_the_box_floor():
    """BASELINE configs[4] at its full size: 16 384 teapot hulls on the static box floor.  Size-independent properties
    for all of them, and a strided sample (every 64th hull: bodies on this grid never meet, so a body's trajectory does
    not depend on who else is in the world) against the oracle, bit for bit."""
    hull = _teapot_hull()
    scene = pkg.scenes.hull_grid(hull, 128, 128, seed=1, y_range=(0.6, 1.6), spin=False, tilt=0.2, floor_box=True).astype("float32")
    steps = 180
    w = _gpu_run(scene, "float32", steps)
    pos, quat, lvel, avel = w.state()
    assert np.all(np.isfinite(pos)) and np.all(np.isfinite(quat))
    assert np.max(np.abs(np.linalg.norm(quat, axis=1) - 1.0)) < 1e-5
    # nobody fell through the floor (top at y = 0, 1 m thick); the first-eight-vertices rule props tipped hulls up on one
    # side, so some are still hopping after three seconds -- the bound is generous on that side
    assert np.all(pos[:, 1] > -0.5) and np.all(pos[:, 1] < 3.0)
    assert 0.3 < np.median(pos[:, 1]) < 0.8
    assert w.last_contact_count() > 2 * scene.n
    idx = np.arange(0, scene.n, 64)
    sub = pkg.scenes.Scene(scene.pos[idx], scene.quat[idx], scene.lvel[idx], scene.avel[idx], scene.mass[idx], scene.inertia[idx],
                           scene.sides[idx], scene.gtype[idx], None, scene.hull_points, scene.hull_planes, scene.static_boxes)
    ow = _oracle_run(_orc("float32"), sub, steps, allow_pairs=True)
    for name, got, ref in zip(("pos", "quat", "lvel", "avel"), (pos, quat, lvel, avel), ow.state()):
        assert np.array_equal(got[idx], ref), name
    w.close()


