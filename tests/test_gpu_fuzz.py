"""Randomised parity: small scenes drawn at random -- boxes, spheres and (sometimes) convex hulls in a loose cluster over
a tilted ground plane, random sizes / spin / lateral velocities, random solver and surface parameters, both precisions --
stepped by the HIP path and by the oracle, compared bit for bit.  Every seed is a fixed scene (numpy Generator), so a
failure names a reproducible case."""
import numpy as np
import pytest

from __graft_entry__ import load_package

pkg = load_package()
pytestmark = pytest.mark.gpu
H = 1.0 / 60.0


def _scene(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(24, 97))
    dtype = "float64" if seed % 2 == 0 else "float32"
    with_hulls = seed % 3 == 0
    nh = int(rng.integers(4, 12)) if with_hulls else 0
    nb = n - nh
    spread = rng.uniform(1.5, 5.0)                    # metres: tight clusters collide at once, loose ones later
    pos = np.stack([rng.uniform(-spread, spread, n), rng.uniform(0.8, 6.0, n), rng.uniform(-spread, spread, n)], axis=1)
    if nh and seed % 6 != 0:
        pos[nb:, 0] += 40.0                           # hulls have no collider against other bodies: usually parked aside,
        pos[nb:, 0] += np.arange(nh) * 3.0            # every other time left in the cluster (they pass through the others)
    axis = rng.normal(size=(n, 3)); axis /= np.linalg.norm(axis, axis=1)[:, None]
    ang = rng.uniform(0, np.pi, n)
    quat = np.concatenate([np.cos(ang / 2)[:, None], axis * np.sin(ang / 2)[:, None]], axis=1)
    lvel = rng.uniform(-1.5, 1.5, (n, 3)); lvel[:, 1] = rng.uniform(-2.0, 0.5, n)
    avel = rng.uniform(-3.0, 3.0, (n, 3))
    gtype = np.where(rng.random(n) < 0.4, pkg.scenes.GEOM_SPHERE, pkg.scenes.GEOM_BOX).astype(np.uint8)
    sides = rng.uniform(0.2, 1.0, (n, 3))
    sph = gtype == pkg.scenes.GEOM_SPHERE
    sides[sph, 0] = rng.uniform(0.1, 0.45, int(sph.sum()))
    mass = np.where(sph, 4.0 / 3.0 * np.pi * sides[:, 0] ** 3, sides.prod(axis=1))[:, None]
    s2 = sides * sides
    inertia = np.where(sph[:, None], (0.4 * mass * s2[:, :1]) * np.ones((1, 3)),
                       mass / 12.0 * np.stack([s2[:, 1] + s2[:, 2], s2[:, 0] + s2[:, 2], s2[:, 0] + s2[:, 1]], 1))
    hull_points = None
    if nh:
        hp = rng.normal(size=(int(rng.integers(12, 150)), 3)) * rng.uniform(0.2, 0.6, 3)
        hull = pkg.hull.build(hp)
        hull_points = hull.points
        gtype[nb:] = pkg.scenes.GEOM_CONVEX
        sides[nb:] = 0.0; sides[nb:, 0] = hull.radius
        mass[nb:, 0] = hull.volume
        inertia[nb:] = hull.inertia
    tilt = rng.uniform(-0.15, 0.15, 2)
    plane = (float(tilt[0]), 1.0, float(tilt[1]), float(rng.uniform(-0.3, 0.3)))
    params = dict(iters=int(rng.integers(8, 31)), sor_w=float(rng.uniform(1.0, 1.4)), erp=float(rng.uniform(0.1, 0.8)),
                  cfm=float(10 ** rng.uniform(-9, -4)), gyro=int(rng.choice([0, 2])),      # explicit mode (1) overflows on slender spinning boxes; it has its own test
                  mu=float("inf") if rng.random() < 0.5 else float(rng.uniform(0.0, 2.0)),
                  bounce_on=bool(rng.random() < 0.7), bounce=float(rng.uniform(0.0, 0.8)),
                  bounce_vel=float(rng.uniform(0.0, 0.5)), max_contacts=int(rng.integers(1, 9)))
    # round 2: static box geoms (AddBodyMap, main.c:735-761) in two scenes out of three -- slabs and posts in and around the
    # cluster, turned about z -- and the hull's faces whenever there are hulls (contacts of hulls with boxes, both kinds)
    statics = None
    if seed % 3 != 1:
        statics = []
        for _ in range(int(rng.integers(1, 4))):
            sz = (float(rng.uniform(0.5, 6.0)), float(rng.uniform(0.3, 2.0)), float(rng.uniform(0.5, 6.0)))
            at = (float(rng.uniform(-spread, spread)), float(rng.uniform(-0.2, 1.5)), float(rng.uniform(-spread, spread)))
            statics.append((sz, at, pkg.scenes._rot_z(float(rng.uniform(-0.4, 0.4)))))
    hull_planes = pkg.hull.planes(hull_points) if hull_points is not None else None
    sc = pkg.scenes.Scene(pos, quat, lvel, avel, mass, inertia, sides, gtype, plane, hull_points, hull_planes, statics).astype(dtype)
    return sc, dtype, params, nb


@pytest.mark.parametrize("seed", range(64))
def test_random_scene_matches_oracle(seed):
    from oracle.orc_ctypes import Oracle
    scene, dtype, p, nb = _scene(seed)
    steps = 150
    mode = pkg.batch.CONTACT_BOUNCE if p["bounce_on"] else 0

    w = pkg.BatchWorld(scene.n, dtype=dtype)
    w.set_quickstep(p["iters"], p["sor_w"]); w.set_erp(p["erp"]); w.set_cfm(p["cfm"]); w.set_gyro_mode(p["gyro"])
    w.set_surface(mode, p["mu"], p["bounce"], p["bounce_vel"]); w.set_max_contacts(p["max_contacts"])
    w.load_scene(scene)
    w.step(H, steps)
    w.synchronize()

    orc = Oracle(dtype)
    lib = orc.lib
    ow = orc.world()
    lib.orc_world_set_quickstep(ow.w, p["iters"], p["sor_w"]); lib.orc_world_set_erp(ow.w, p["erp"])
    lib.orc_world_set_cfm(ow.w, p["cfm"]); lib.orc_world_set_gyro_mode(ow.w, p["gyro"])
    lib.orc_world_set_surface(ow.w, mode, p["mu"], p["bounce"], p["bounce_vel"])
    lib.orc_world_set_max_contacts(ow.w, p["max_contacts"])
    ow.add_plane(*scene.plane)
    if scene.hull_points is not None:
        ow.set_hull(scene.hull_points)
        ow.set_hull_faces(scene.hull_planes)
    for sz, at, R12 in (scene.static_boxes or []):
        ow.add_static_box(sz, at, R12)
    for i in range(scene.n):                         # body by body: geometry classes are interleaved
        b = lib.orc_body_create(ow.w)
        lib.orc_body_set_position(ow.w, b, *scene.pos[i])
        _, qp = orc.arr(scene.quat[i]); lib.orc_body_set_quaternion(ow.w, b, qp)
        lib.orc_body_set_linear_vel(ow.w, b, *scene.lvel[i])
        lib.orc_body_set_angular_vel(ow.w, b, *scene.avel[i])
        _, ip = orc.arr(np.diag(scene.inertia[i]).ravel()); lib.orc_body_set_mass(ow.w, b, scene.mass[i, 0], ip)
        gt = scene.gtype[i]
        g = (lib.orc_geom_create_sphere(ow.w, scene.sides[i, 0]) if gt == pkg.scenes.GEOM_SPHERE
             else lib.orc_geom_create_convex(ow.w) if gt == pkg.scenes.GEOM_CONVEX
             else lib.orc_geom_create_box(ow.w, *scene.sides[i]))
        lib.orc_geom_set_category_bits(ow.w, g, 2); lib.orc_geom_set_collide_bits(ow.w, g, 3)
        lib.orc_geom_set_body(ow.w, g, b)
    ow.run(H, steps)

    for name, a, r in zip(("pos", "quat", "lvel", "avel"), w.state(), ow.state()):
        assert np.all(np.isfinite(r)), f"oracle {name} not finite (seed {seed}: {p})"
        assert np.array_equal(a, r), (f"seed {seed} {dtype} {name}: max abs diff {np.max(np.abs(a - r))} "
                                      f"at body {int(np.argmax(np.abs(a - r).max(axis=1)))} of {scene.n} (boxes/spheres {nb}); {p}")
    assert w.last_contact_count() == ow.n_contacts()


# ----------------------------------------------------------------- random sequences of API calls
@pytest.fixture(scope="module")
def one_rank_group():
    """a one-rank RCCL group for the seeds that step through shard.ShardedStepper"""
    import os
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("seed", range(32))
def test_random_call_sequences_match_oracle(seed, one_rank_group):
    """A random walk over the batch API between steps -- forces and torques on some bodies, bodies teleported or given
    new velocities, ticks-per-launch changed -- mirrored call for call on the oracle; the state
    is compared bit for bit after every step.  Exercises the bookkeeping around the kernels: pending accumulators, stale
    safe zones, chunk lengths, fused launches."""
    from oracle.orc_ctypes import Oracle
    rng = np.random.default_rng(7000 + seed)
    dtype = "float64" if seed % 2 == 0 else "float32"
    plane = seed % 3 != 0
    sharded = seed % 4 == 3                  # every fourth seed steps through the N>1 loop (one rank: chunk protocol, ghosts slots)
    nx, nz = int(rng.integers(6, 14)), int(rng.integers(6, 14))
    if sharded:
        nx = 4 * int(rng.integers(2, 4))
    scene = pkg.scenes.box_grid(nx, nz, seed=50 + seed, y_range=(0.8, 8.0), spin=True, box_mass=bool(seed % 4 == 1),
                                plane=plane).astype(dtype)
    n = scene.n
    stepper = None
    if sharded:
        import torch
        L = pkg.shard.SlabLayout(nx, nz)
        w = pkg.BatchWorld(L.n_total, dtype=dtype)
        w.load_scene(scene)
        w.set_active_count(n)
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        w.set_stream(stream.cuda_stream)
        stepper = pkg.shard.ShardedStepper(w, L, 0, 1, collide=True, geometry=(scene.sides, scene.gtype),
                                           ops=pkg.shard.DeviceOps(w, torch.device("cuda", 0), stream))
    else:
        w = pkg.BatchWorld(n, dtype=dtype)
        w.load_scene(scene)
    # every fifth seed drops the grid onto a static plank as well (AddBodyMap): static contacts, bodies involved every tick
    statics = []
    if not sharded and seed % 5 == 2:
        statics = [((float(rng.uniform(6, 20)), 0.6, float(rng.uniform(6, 20))), (float(rng.uniform(-3, 3)), 0.3, float(rng.uniform(-3, 3))),
                    pkg.scenes._rot_z(float(rng.uniform(-0.1, 0.1))))]
        w.set_static_boxes(statics)
    orc = Oracle(dtype)
    lib = orc.lib
    ow = orc.world()
    if plane:
        ow.add_plane(*scene.plane)
    for sz, at, R12 in statics:
        ow.add_static_box(sz, at, R12)
    ow.add_boxes(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia, scene.sides)
    rt = orc.dtype.type

    def check(tag):
        w.synchronize()
        for name, a, r in zip(("pos", "quat", "lvel", "avel"), w.state(), ow.state()):
            assert np.array_equal(a[:n], r), f"seed {seed} after {tag}: {name} differs by {np.max(np.abs(a[:n] - r))}"

    log = []
    for op_i in range(int(rng.integers(10, 18))):
        op = int(rng.integers(0, 5))            # 4 = just step
        if op == 0:                                   # forces / torques on a few bodies, consumed by the next tick
            ids = rng.choice(n, size=int(rng.integers(1, 6)), replace=False)
            if sharded:
                continue                              # (the sharded loop takes pending forces on its exact path: covered below by teleports)
            f = np.zeros((n, 3), orc.dtype); t = np.zeros((n, 3), orc.dtype)
            f[ids] = rng.uniform(-20, 20, (len(ids), 3)); t[ids] = rng.uniform(-2, 2, (len(ids), 3))
            w.upload(pkg.batch.FORCE, f); w.upload(pkg.batch.TORQUE, t)
            for i in ids:
                lib.orc_body_add_force(ow.w, int(i), rt(f[i, 0]), rt(f[i, 1]), rt(f[i, 2]))
                lib.orc_body_add_torque(ow.w, int(i), rt(t[i, 0]), rt(t[i, 1]), rt(t[i, 2]))
            log.append(f"force{list(ids)}")
        elif op == 1:                                 # teleport some bodies sideways / upwards (zones go stale)
            pos = w.download(pkg.batch.POS, 0, n)
            ids = rng.choice(n, size=int(rng.integers(1, 4)), replace=False)
            pos[ids] += rng.uniform(-0.3, 0.3, (len(ids), 3)).astype(orc.dtype) + np.array([0, 0.5, 0], orc.dtype)
            w.upload(pkg.batch.POS, pos)
            for i in ids:
                lib.orc_body_set_position(ow.w, int(i), rt(pos[i, 0]), rt(pos[i, 1]), rt(pos[i, 2]))
            log.append(f"teleport{list(ids)}")
        elif op == 2:                                 # new velocities, some of them sideways
            v = w.download(pkg.batch.LVEL, 0, n)
            ids = rng.choice(n, size=int(rng.integers(1, 6)), replace=False)
            v[ids] = rng.uniform(-2, 2, (len(ids), 3)).astype(orc.dtype)
            w.upload(pkg.batch.LVEL, v)
            for i in ids:
                lib.orc_body_set_linear_vel(ow.w, int(i), rt(v[i, 0]), rt(v[i, 1]), rt(v[i, 2]))
            log.append(f"vel{list(ids)}")
        elif op == 3:
            k = int(rng.choice([1, 2, 5, 8, 32]))
            w.set_ticks_per_launch(k)
            log.append(f"tpl{k}")
        steps = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 70]))
        if stepper is not None:
            stepper.run(H, steps)
            stepper.drain()
        else:
            w.step(H, steps)
        ow.run(H, steps)
        log.append(f"step{steps}")
        check(" ".join(log[-6:]))
    if sharded:
        import torch
        torch.cuda.set_stream(torch.cuda.default_stream())
    w.close()


@pytest.mark.parametrize("dtype,ticks", [("float32", 8), ("float64", 32)])
def test_sixteen_thousand_drifting_bodies_with_rollbacks(dtype, ticks):
    """Free flight at a size where chunks grow long (up to 256 ticks, fused launches) while bodies drift sideways into
    their neighbours: violations inside long chunks, rollbacks, zone rebuilds and exact ticks with hundreds of pairs."""
    from oracle.orc_ctypes import Oracle
    scene = pkg.scenes.box_grid(128, 128, seed=77, y_range=(20.0, 26.0), spin=True, box_mass=True, plane=False).astype(dtype)
    rng = np.random.default_rng(9)
    scene.lvel[:, 0] = rng.uniform(-0.25, 0.25, scene.n).astype(scene.lvel.dtype)
    scene.lvel[:, 2] = rng.uniform(-0.25, 0.25, scene.n).astype(scene.lvel.dtype)
    steps = 400
    w = pkg.BatchWorld(scene.n, dtype=dtype)
    w.set_ticks_per_launch(ticks)
    w.load_scene(scene)
    w.step(H, steps)
    w.synchronize()
    orc = Oracle(dtype)
    ow = orc.world()
    ow.add_boxes(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia, scene.sides)
    pairs = 0
    for _ in range(steps):
        ow.tick(H)
        pairs += ow.n_body_pairs()
    for name, a, r in zip(("pos", "quat", "lvel", "avel"), w.state(), ow.state()):
        assert np.array_equal(a, r), f"{name}: max abs diff {np.max(np.abs(a - r))}"
    st = w.collision_stats()
    assert pairs > 100 and st["pair_ticks"] > 0 and st["fast_ticks"] > 0
