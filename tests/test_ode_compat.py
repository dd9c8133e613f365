"""The ODE C API boundary (include/ode/ode.h): a C program that makes the reference's own sequence of
ODE calls (tests/harness/ode_tick_harness.c) compiles against the shim headers, links against
libode_mi355*.so, and -- on the GPU box -- reproduces the CPU oracle's poses on the reference's scene."""
import os
import re
import subprocess

import numpy as np
import pytest

from __graft_entry__ import load_package, ROOT

pkg = load_package()
PKG = os.path.join(ROOT, "rl-ode-physics_amd")
HARNESS = os.path.join(ROOT, "tests", "harness", "ode_tick_harness.c")


def _build_harness(tmp, single):
    exe = os.path.join(tmp, "harness_s" if single else "harness_d")
    cmd = ["gcc", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), HARNESS, "-o", exe,
           "-L" + PKG, "-lode_mi355_single" if single else "-lode_mi355", "-Wl,-rpath," + PKG, "-lm"]
    if single:
        cmd.insert(1, "-DdSINGLE")
    subprocess.run(cmd, check=True)
    return exe


def _scene_text(dt, steps, use_plane, statics, bodies):
    ident = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]
    lines = [f"{dt!r} {steps} {int(use_plane)}", str(len(statics))]
    for size, pos, R in statics:
        lines.append(" ".join(repr(float(v)) for v in (*size, *pos, *R)))
    lines.append(str(len(bodies)))
    for kind, size, pos in bodies:
        lines.append(f"{kind} " + " ".join(repr(float(v)) for v in (*size, *pos, *ident)))
    return "\n".join(lines) + "\n"


def _oracle_poses(dtype, dt, steps, use_plane, statics, bodies, up_front=None, every=0, exact=False, ode_order_seed=None,
                  exact_after=None):
    from oracle.orc_ctypes import Oracle
    import ctypes as C
    orc = Oracle(dtype)
    lib = orc.lib
    ow = orc.world()
    ow.set_stepper(exact)
    if ode_order_seed is not None:          # stock ODE's row order: joint-discovery numbering + the LCG shuffle every 8th sweep
        lib.orc_world_set_row_order(ow.w, 1)
        lib.orc_rand_seed(ode_order_seed)
    if use_plane:
        ow.add_plane(0, 1, 0, 0)
    for size, pos, R in statics:
        g = lib.orc_geom_create_box(ow.w, *size)
        lib.orc_geom_set_position(ow.w, g, *pos)
        _, rp = orc.arr(R)
        lib.orc_geom_set_rotation(ow.w, g, rp)
        lib.orc_geom_set_category_bits(ow.w, g, 0xFFFFFFFE)      # the double SetCategoryBits of main.c:751-752
    ident = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]

    def create(kind, size, pos):
        b = lib.orc_body_create(ow.w)
        lib.orc_body_set_position(ow.w, b, *pos)
        _, rp = orc.arr(ident)
        lib.orc_body_set_rotation(ow.w, b, rp)
        g = lib.orc_geom_create_sphere(ow.w, size[0]) if kind == 1 else lib.orc_geom_create_box(ow.w, *size)
        lib.orc_geom_set_category_bits(ow.w, g, 2)
        lib.orc_geom_set_collide_bits(ow.w, g, 3)
        lib.orc_geom_set_body(ow.w, g, b)

    # bodies spawn as the harness spawns them (HARNESS_SPAWN): `up_front` at the start, then one every `every` ticks
    up_front = len(bodies) if up_front is None else up_front
    created = 0
    for s in range(steps + 1):
        while created < len(bodies) and (created < up_front or s == steps
                                         or (every > 0 and s >= (created - up_front + 1) * every)):
            create(*bodies[created])
            created += 1
        if s < steps:
            if exact_after is not None:                 # HARNESS_EXACT_AFTER: QuickStep while the pile settles, dWorldStep from then on
                ow.set_stepper(s >= exact_after)
            ow.run(dt, 1)
    out = np.zeros((len(bodies), 16), orc.dtype)
    RP = C.POINTER(orc.real)
    for i in range(len(bodies)):
        lib.orc_pack_transform(out[i].ctypes.data_as(RP), lib.orc_body_get_position(ow.w, i),
                               lib.orc_body_get_rotation(ow.w, i))
    return out, ow


# ------------------------------------------------------------------------------------------------ CPU
def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dmx[A-Z]\w*|d[A-Z]\w*)\s*\(", src)) - {"dNearCallback"})


# the 33 symbols /root/reference/src/main.c references
REFERENCE_SYMBOLS = """dInitODE dCloseODE dWorldCreate dWorldDestroy dWorldSetGravity dHashSpaceCreate dJointGroupCreate
dJointGroupEmpty dJointGroupDestroy dBodyCreate dBodyDestroy dBodySetPosition dBodySetRotation dBodySetKinematic
dBodyGetPosition dBodyGetRotation dBodyAddForce dCreateBox dCreateSphere dGeomDestroy dGeomSetBody dGeomGetBody
dGeomSetPosition dGeomSetRotation dGeomGetPosition dGeomGetRotation dGeomSetCategoryBits dGeomSetCollideBits
dSpaceCollide dCollide dJointCreateContact dJointAttach dWorldStep""".split()


@pytest.mark.parametrize("libname", ["libode_mi355.so", "libode_mi355_single.so"])
def test_ode_symbols_exported(libname):
    import ctypes as C
    pkg._lib.load()
    lib = C.CDLL(os.path.join(PKG, libname))
    names = _declared("ode/ode.h")
    assert len(REFERENCE_SYMBOLS) == 33 and set(REFERENCE_SYMBOLS) <= set(names)
    for n in names + ["dWorldQuickStep", "dCreatePlane", "dBodySetMass", "dMassSetBox"]:
        assert hasattr(lib, n), f"{n} declared in include/ode/ode.h but not exported by {libname}"
    for n in _declared("dmx_batch.h"):
        assert hasattr(lib, n)


@pytest.mark.skipif(not os.path.exists("/root/reference/src/main.c"), reason="reference tree not present on this box")
def test_reference_symbol_list_is_what_main_c_calls():
    """Reads the reference's main.c as text: every ODE function it calls is in REFERENCE_SYMBOLS, and every ODE
    type / constant / field it names is spelled out in include/ode/*.h."""
    src = open("/root/reference/src/main.c").read()
    called = set(re.findall(r"\b(d[A-Z]\w*)\s*\(", src))          # dBodyAddForce sits in a // comment (main.c:532)
    assert called == set(REFERENCE_SYMBOLS)
    hdr = "".join(open(os.path.join(ROOT, "include", "ode", h)).read() for h in ("common.h", "ode.h"))
    for ident in sorted(set(re.findall(r"\b(d[A-Z]\w*)\b", src)) - called):
        assert re.search(r"\b%s\b" % ident, hdr), f"main.c names {ident}; include/ode does not define it"
    for field in set(re.findall(r"\.(surface|geom|mode|mu|bounce_vel|bounce|soft_cfm|soft_erp|depth|normal|pos)\b", src)):
        assert re.search(r"\b%s\b" % field, hdr)


@pytest.mark.parametrize("single", [False, True])
def test_reference_call_sequence_compiles_and_links(tmp_path, single):
    exe = _build_harness(str(tmp_path), single)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert ("libode_mi355_single.so" if single else "libode_mi355.so") in out


@pytest.mark.parametrize("single", [False, True])
def test_contact_structs_have_the_ode_0_16_layout(tmp_path, single):
    """include/ode/ode.h claims the ODE 0.13 - 0.16 line's dSurfaceParameters / dContactGeom / dContact [ODE-recall contact.h]:
    int mode, then dReal mu, mu2, rho, rho2, rhoN, bounce, bounce_vel, soft_erp, soft_cfm, motion1, motion2, motionN, slip1,
    slip2; dContactGeom = pos[4], normal[4], depth, g1, g2, side1, side2; dContact = surface, geom, fdir1[4].  An object
    compiled against stock headers of that line passes `bounce` where this library reads it."""
    src = tmp_path / "layout.c"
    src.write_text(r"""
#include <stdio.h>
#include <stddef.h>
#include <ode/ode.h>
#define P(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main(void) {
    P(dSurfaceParameters, mode); P(dSurfaceParameters, mu); P(dSurfaceParameters, mu2); P(dSurfaceParameters, rho);
    P(dSurfaceParameters, rho2); P(dSurfaceParameters, rhoN); P(dSurfaceParameters, bounce); P(dSurfaceParameters, bounce_vel);
    P(dSurfaceParameters, soft_erp); P(dSurfaceParameters, soft_cfm); P(dSurfaceParameters, motion1); P(dSurfaceParameters, motion2);
    P(dSurfaceParameters, motionN); P(dSurfaceParameters, slip1); P(dSurfaceParameters, slip2);
    printf("sizeof.dSurfaceParameters %zu\n", sizeof(dSurfaceParameters));
    P(dContactGeom, pos); P(dContactGeom, normal); P(dContactGeom, depth); P(dContactGeom, g1); P(dContactGeom, g2);
    P(dContactGeom, side1); P(dContactGeom, side2);
    P(dContact, surface); P(dContact, geom); P(dContact, fdir1);
    printf("flags %d %d %d %d %d\n", dContactBounce, dContactRolling, dContactApprox1_N, dContactApprox1, dContactSoftCFM);
    return 0;
}
""")
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror"] + (["-DdSINGLE"] if single else []) +
                   ["-I" + os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
    got = dict(line.rsplit(" ", 1) for line in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()
               if not line.startswith("flags"))
    r = 4 if single else 8
    first = r                                        # `int mode` is padded up to dReal's alignment
    names = ["mu", "mu2", "rho", "rho2", "rhoN", "bounce", "bounce_vel", "soft_erp", "soft_cfm", "motion1", "motion2", "motionN", "slip1", "slip2"]
    assert got["dSurfaceParameters.mode"] == "0"
    for k, nm in enumerate(names):
        assert int(got[f"dSurfaceParameters.{nm}"]) == first + k * r, nm
    assert int(got["sizeof.dSurfaceParameters"]) == first + len(names) * r
    assert [int(got[f"dContactGeom.{f}"]) for f in ("pos", "normal", "depth", "g1", "g2")] == [0, 4 * r, 8 * r, 8 * r + 8, 8 * r + 16] if not single \
        else [int(got[f"dContactGeom.{f}"]) for f in ("pos", "normal", "depth")] == [0, 16, 32]
    assert int(got["dContact.surface"]) == 0 and int(got["dContact.geom"]) >= first + len(names) * r
    out = subprocess.run([exe], capture_output=True, text=True).stdout
    assert "flags 4 1024 16384 28672 16" in out


def test_reference_spawner_draws():
    b = pkg.scenes.reference_spawn(200, seed=1)
    kinds = [k for k, _, _ in b]
    assert 60 < kinds.count(2) < 140                      # Rand_Int(0,2) == 0 -> box, about half
    for kind, size, pos in b:
        assert -4 <= pos[0] <= 4 and 20 <= pos[1] <= 50 and -4 <= pos[2] <= 4      # main.c:504
        if kind == 2:
            assert all(0.2 <= s <= 1.0 for s in size)     # main.c:508
        else:
            assert 0.1 <= size[0] <= 0.4                  # main.c:516
    m = pkg.scenes.reference_map()
    assert m[0][0] == (100.0, 1.0, 100.0) and len(m) == 4


# ------------------------------------------------------------------------------------------------ GPU
def _run_harness(exe, text, env=None, stepper="quick"):
    """stepper: "quick" -> the harness calls dWorldQuickStep (bit-identical to the oracle's SOR); "exact" -> dWorldStep, the
    reference's own call (main.c:213), every island's LCP solved exactly"""
    p = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=600,
                       env={**os.environ, "HARNESS_STEPPER": stepper, **(env or {})})
    assert p.returncode == 0, p.stderr[-2000:]
    return np.array([[float(v) for v in line.split()] for line in p.stdout.strip().splitlines()])


@pytest.mark.gpu
@pytest.mark.parametrize("single", [False, True])
def test_reference_scene_through_ode_api_matches_oracle(tmp_path, single):
    """The reference's map (floor + 3 walls, static boxes) and 48 spawned boxes/spheres dropped from
    y in [1.5, 9]: box-box, sphere-box, sphere-sphere contacts and multi-body islands, QuickStep at
    1/120 s (main.c:208), 360 ticks."""
    dtype = "float32" if single else "float64"
    statics = pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(48, seed=7, y_range=(1.5, 9.0))
    dt, steps = 1.0 / 120.0, 360
    exe = _build_harness(str(tmp_path), single)
    got = _run_harness(exe, _scene_text(dt, steps, False, statics, bodies))
    ref, ow = _oracle_poses(dtype, dt, steps, False, statics, bodies)
    assert ow.n_contacts() > 48 and ow.n_body_pairs() > 0       # resting on the floor and on each other
    assert got.shape == ref.shape
    assert np.all(np.isfinite(got))
    assert np.array_equal(got.astype(ref.dtype), ref), np.abs(got - ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("big_rows", ["1", "96", "1000000"])
def test_large_islands_take_the_workgroup_kernel_with_identical_bits(tmp_path, big_rows):
    """A pile in the reference's pen (floor + walls): its island is solved by solve_island_wg under a level
    schedule; the result must equal the sequential sweep of the oracle bit for bit."""
    statics = pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(120, seed=11, y_range=(1.5, 6.0))
    dt, steps = 1.0 / 120.0, 300
    exe = _build_harness(str(tmp_path), False)
    got = _run_harness(exe, _scene_text(dt, steps, False, statics, bodies), env={"DMX_BIG_ISLAND_ROWS": big_rows})
    ref, ow = _oracle_poses("float64", dt, steps, False, statics, bodies)
    assert ow.n_contacts() > 100 and ow.n_body_pairs() > 20
    assert np.array_equal(got, ref), np.abs(got - ref).max()


@pytest.mark.gpu
def test_plane_scene_through_ode_api_matches_oracle(tmp_path):
    bodies = pkg.scenes.reference_spawn(64, seed=3, y_range=(0.8, 6.0))
    dt, steps = 1.0 / 60.0, 300
    exe = _build_harness(str(tmp_path), False)
    got = _run_harness(exe, _scene_text(dt, steps, True, [], bodies))
    ref, ow = _oracle_poses("float64", dt, steps, True, [], bodies)
    assert np.array_equal(got, ref), np.abs(got - ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("device_pairs", ["0", "1"])
@pytest.mark.parametrize("seed", range(10))
def test_random_spawns_through_ode_api_match_oracle(tmp_path, seed, device_pairs):
    """The reference's spawner (main.c:502-521) with other seeds, counts, drop heights and step sizes, in the reference's
    pen or on a plane, both precisions: poses after the run are the oracle's, bit for bit -- with dSpaceCollide's pairs
    found by the host sweep (small worlds' default) and by the device search (DMX_COMPAT_DEVICE_PAIRS=1: what worlds of
    thousands of bodies use), the user's callback seeing the same pairs in the same order either way."""
    rng = np.random.default_rng(500 + seed)
    single = bool(seed % 2)
    n = int(rng.integers(8, 160))
    use_plane = bool(rng.random() < 0.4)
    statics = [] if use_plane else pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(n, seed=100 + seed, y_range=(1.2, float(rng.uniform(4.0, 25.0))))
    dt = 1.0 / float(rng.choice([60.0, 120.0]))
    steps = int(rng.integers(120, 320))
    exe = _build_harness(str(tmp_path), single)
    got = _run_harness(exe, _scene_text(dt, steps, use_plane, statics, bodies), env={"DMX_COMPAT_DEVICE_PAIRS": device_pairs})
    ref, ow = _oracle_poses("float32" if single else "float64", dt, steps, use_plane, statics, bodies)
    assert np.all(np.isfinite(ref))
    assert np.array_equal(got.astype(ref.dtype), ref), (seed, n, use_plane, dt, steps, np.abs(got - ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_bodies_spawned_between_ticks_and_poses_read_every_frame(tmp_path, seed):
    """The reference's run-time pattern: bodies are added between ticks (key M, main.c:502-521) and every live body's
    pose is read each broadcast frame (main.c:221-237) -- host mirrors and device state change hands every tick."""
    rng = np.random.default_rng(900 + seed)
    single = bool(seed % 2)
    n = int(rng.integers(20, 90))
    up_front, every = int(rng.integers(0, 8)), int(rng.integers(1, 5))
    steps = (n - up_front) * every + int(rng.integers(5, 60))
    use_plane = bool(seed % 3 == 0)
    statics = [] if use_plane else pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(n, seed=300 + seed, y_range=(1.2, 6.0))
    dt = 1.0 / 120.0
    exe = _build_harness(str(tmp_path), single)
    got = _run_harness(exe, _scene_text(dt, steps, use_plane, statics, bodies),
                       env={"HARNESS_SPAWN": f"{up_front} {every}", "HARNESS_READBACK": str(int(rng.integers(1, 4))),
                            "DMX_COMPAT_DEVICE_PAIRS": str(seed % 2)})
    ref, ow = _oracle_poses("float32" if single else "float64", dt, steps, use_plane, statics, bodies, up_front, every)
    assert np.array_equal(got.astype(ref.dtype), ref), (seed, n, up_front, every, steps, np.abs(got - ref).max())


@pytest.mark.gpu
def test_growth_beyond_512_bodies(tmp_path):
    """More bodies than MAX_BODIES (inc/body.h:6): the world's device batch doubles transparently."""
    bodies = pkg.scenes.reference_spawn(700, seed=5, y_range=(2.0, 400.0))
    dt, steps = 1.0 / 60.0, 20
    exe = _build_harness(str(tmp_path), False)
    got = _run_harness(exe, _scene_text(dt, steps, True, [], bodies))
    ref, _ = _oracle_poses("float64", dt, steps, True, [], bodies)
    assert np.array_equal(got, ref)


# ------------------------------------------------------------------------------------------------ ODE API details (GPU)
def _ode(single=False):
    import ctypes as C
    pkg._lib.load()
    lib = C.CDLL(os.path.join(PKG, "libode_mi355_single.so" if single else "libode_mi355.so"))
    real = C.c_float if single else C.c_double
    P = C.c_void_p
    for name, res, args in [
        ("dWorldCreate", P, []), ("dWorldDestroy", None, [P]), ("dWorldSetGravity", None, [P, real, real, real]),
        ("dWorldQuickStep", C.c_int, [P, real]), ("dBodyCreate", P, [P]), ("dBodyDestroy", None, [P]),
        ("dBodySetPosition", None, [P, real, real, real]), ("dBodySetLinearVel", None, [P, real, real, real]),
        ("dBodyGetPosition", C.POINTER(real), [P]), ("dBodyGetLinearVel", C.POINTER(real), [P]),
        ("dBodyGetRotation", C.POINTER(real), [P]), ("dBodySetKinematic", None, [P]), ("dBodyAddForce", None, [P, real, real, real]),
        ("dBodySetAngularVel", None, [P, real, real, real]),
        ("dmxWorldSnapshotTransforms", C.c_int, [P, C.POINTER(P), C.c_int, C.POINTER(real)]),
        ("dmxWorldSnapshotBodyStates", C.c_int, [P, P, C.c_size_t, C.c_int, P, C.c_size_t]),
    ]:
        f = getattr(lib, name); f.restype = res; f.argtypes = args
    return lib, real


@pytest.mark.gpu
def test_ode_api_forces_kinematic_slot_reuse_and_snapshot():
    import ctypes as C
    lib, real = _ode()
    h = 1.0 / 60.0
    w = lib.dWorldCreate()
    lib.dWorldSetGravity(w, 0.0, -9.8, 0.0)
    a, k, f = lib.dBodyCreate(w), lib.dBodyCreate(w), lib.dBodyCreate(w)
    lib.dBodySetPosition(a, 1.0, 10.0, 0.0)
    lib.dBodySetPosition(k, 2.0, 5.0, 0.0); lib.dBodySetKinematic(k); lib.dBodySetLinearVel(k, 0.5, 0.0, 0.0)
    lib.dBodySetPosition(f, 3.0, 0.0, 0.0); lib.dBodyAddForce(f, 6.0, 9.8, 0.0)      # cancels gravity for one tick
    lib.dBodySetAngularVel(a, 0.0, 2.0, 0.0)
    assert lib.dWorldQuickStep(w, h) == 1
    pa, pk, pf = (lib.dBodyGetPosition(x) for x in (a, k, f))
    assert abs(pa[1] - (10.0 - 9.8 * h * h)) < 1e-12                       # KAT-1, one tick
    assert abs(pk[0] - (2.0 + 0.5 * h)) < 1e-15 and pk[1] == 5.0          # kinematic: moves with its velocity, ignores gravity
    vf = lib.dBodyGetLinearVel(f)
    assert abs(vf[0] - 6.0 * h) < 1e-15 and abs(vf[1]) < 1e-15             # force applied ...
    lib.dWorldQuickStep(w, h)
    vf = lib.dBodyGetLinearVel(f)
    assert abs(vf[0] - 6.0 * h) < 1e-15 and abs(vf[1] + 9.8 * h) < 1e-15   # ... exactly once
    # device-side GetTransformMat for a list of bodies
    ids = (C.c_void_p * 2)(a, k)
    out = (real * 32)()
    assert lib.dmxWorldSnapshotTransforms(w, ids, 2, out) == 1
    R, p = lib.dBodyGetRotation(a), lib.dBodyGetPosition(a)
    assert list(out[:16]) == [R[0], R[4], R[8], 0, R[1], R[5], R[9], 0, R[2], R[6], R[10], 0, p[0], p[1], p[2], 1]
    assert out[16 + 12] == lib.dBodyGetPosition(k)[0]
    # the same over arrays laid out like the reference's Body[] / BodyState[] (body.h:20-31): strided handles, strided
    # states, a null handle (static geom / empty slot) skipped and its state left alone
    class Body(C.Structure):
        _fields_ = [("body", C.c_void_p), ("geom", C.c_void_p), ("type", C.c_int)]

    class BodyState(C.Structure):
        _fields_ = [("type", C.c_int), ("transform", real * 16), ("size", C.c_float * 3), ("col", C.c_ubyte * 4)]
    bodies = (Body * 3)(Body(a, None, 2), Body(None, None, 2), Body(k, None, 1))
    states = (BodyState * 3)()
    states[1].transform[5] = 42.0
    off = BodyState.transform.offset
    assert lib.dmxWorldSnapshotBodyStates(w, C.addressof(bodies), C.sizeof(Body), 3,
                                          C.addressof(states) + off, C.sizeof(BodyState)) == 2
    assert list(states[0].transform) == list(out[:16]) and list(states[2].transform) == list(out[16:32])
    assert states[1].transform[5] == 42.0 and states[0].type == 0
    # destroy + create reuses the slot; the new body starts from ODE's defaults
    lib.dBodyDestroy(a)
    n = lib.dBodyCreate(w)
    lib.dWorldQuickStep(w, h)
    pn = lib.dBodyGetPosition(n)
    assert abs(pn[1] + 9.8 * h * h) < 1e-15 and pn[0] == 0.0
    lib.dWorldDestroy(w)


# ------------------------------------------------------------------------------------------------ dWorldStep: the exact solve
def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


@pytest.mark.gpu
@pytest.mark.parametrize("single", [False, True])
def test_dworldstep_stack_of_three_matches_the_oracles_exact_solve(tmp_path, single):
    """dWorldStep (main.c:213) on a stack of three boxes on the reference's floor: the island's LCP is solved exactly on
    the device (lcp_island_wg); poses agree with the oracle's exact stepper within north_star's 1e-5."""
    dtype = "float32" if single else "float64"
    statics = pkg.scenes.reference_map()[:1]
    bodies = [(2, (1.0, 1.0, 1.0), (0.0, 1.0 + 1.0 * k, 0.0)) for k in range(3)]      # floor top at y = 0.5: resting, depth 0
    dt, steps = 1.0 / 120.0, 120
    exe = _build_harness(str(tmp_path), single)
    got = _run_harness(exe, _scene_text(dt, steps, False, statics, bodies), stepper="exact")
    ref, ow = _oracle_poses(dtype, dt, steps, False, statics, bodies, exact=True)
    assert ow.n_contacts() == 12
    assert _rel(got.astype(ref.dtype), ref) <= 1e-5
    # and the stack stands: the exact solve carries the weight without the drift 20 SOR sweeps leave behind
    assert np.max(np.abs(got[:, 13] - np.array([1.0, 2.0, 3.0]))) < 2e-3


@pytest.mark.gpu
def test_dworldstep_in_the_reference_pen_matches_the_oracles_exact_solve(tmp_path):
    """the reference's scene stepped with the reference's call: 24 boxes and spheres dropping into the pen, 200 ticks of
    dWorldStep at 1/120 s -- multi-body islands, box-box / sphere-box contacts against floor, walls and one another"""
    statics = pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(24, seed=21, y_range=(1.2, 5.0))
    dt, steps = 1.0 / 120.0, 200
    exe = _build_harness(str(tmp_path), False)
    got = _run_harness(exe, _scene_text(dt, steps, False, statics, bodies), stepper="exact")
    ref, ow = _oracle_poses("float64", dt, steps, False, statics, bodies, exact=True)
    assert ow.n_contacts() > 24
    assert _rel(got, ref) <= 1e-5
    quick, _ = _oracle_poses("float64", dt, steps, False, statics, bodies, exact=False)
    assert _rel(quick, ref) > 1e-5                       # the two steppers are not the same computation


@pytest.mark.gpu
def test_large_world_through_ode_api_takes_its_pairs_from_the_device(tmp_path):
    """3 000 bodies through the ODE API (well past MAX_BODIES, inc/body.h:6): above 2 048 bodies dSpaceCollide's pair search
    runs on the device by default; poses equal the oracle's, which equal the host search's"""
    rng = np.random.default_rng(77)
    n = 3000
    bodies = []
    for k in range(n):                        # a 60 x 50 grid of the spawner's boxes / spheres, close enough to collide as they tumble
        kind, size, _ = pkg.scenes.reference_spawn(1, seed=1000 + k)[0]
        bodies.append((kind, size, (1.1 * (k % 60) - 33.0, 1.0 + 2.0 * rng.random(), 1.1 * (k // 60) - 27.0)))
    statics = [((100.0, 1.0, 100.0), (0.0, 0.0, 0.0), pkg.scenes._rot_z(0.0))]
    dt, steps = 1.0 / 60.0, 40
    exe = _build_harness(str(tmp_path), False)
    text = _scene_text(dt, steps, False, statics, bodies)
    got = _run_harness(exe, text)
    ref, ow = _oracle_poses("float64", dt, steps, False, statics, bodies)
    assert ow.n_body_pairs() > 100
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    assert np.array_equal(_run_harness(exe, text, env={"DMX_COMPAT_DEVICE_PAIRS": "0"}), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("single,seed", [(False, 12345), (True, 7), (False, 0)])
def test_quickstep_with_odes_own_row_order_and_shuffle_matches_the_oracle(tmp_path, single, seed):
    """DMX_ROW_ORDER=ode:<seed>: dWorldQuickStep numbers an island's rows as ODE's island builder discovers the joints and
    re-shuffles them with ODE's LCG before sweeps 0, 8 and 16 (RANDOMLY_REORDER_CONSTRAINTS) -- the oracle's ORC_ORDER_ODE
    mode, bit for bit, in the reference's pen with piles (multi-body islands, so the order matters)."""
    dtype = "float32" if single else "float64"
    statics = pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(60, seed=41, y_range=(1.2, 7.0))
    dt, steps = 1.0 / 120.0, 240
    exe = _build_harness(str(tmp_path), single)
    text = _scene_text(dt, steps, False, statics, bodies)
    got = _run_harness(exe, text, env={"DMX_ROW_ORDER": f"ode:{seed}"})
    ref, ow = _oracle_poses(dtype, dt, steps, False, statics, bodies, ode_order_seed=seed)
    assert ow.n_contacts() > 60 and ow.n_body_pairs() > 5
    assert np.array_equal(got.astype(ref.dtype), ref), np.abs(got - ref).max()
    fixed, _ = _oracle_poses(dtype, dt, steps, False, statics, bodies)
    assert not np.array_equal(fixed, ref)                  # and it is not the creation-order sweep
