"""bench.py -- body-steps/s of the rigid-body step hot path on MI355X.

One "step" = one tick (collide -> QuickStep -> clear contacts, main.c:211-215)
over the whole batch of synthetic bodies.  At N=1 the workload is BASELINE.json
configs[1]: 1 048 576 free-falling boxes, no contacts, dt = 1/60.  With N>1
every rank owns one such slab of disjoint islands (weak scaling) and exchanges
the state of its slab-boundary bodies with an RCCL all-gather -- at the end of
every collision-proof chunk for these ballistic scenes, every tick otherwise
or with --exchange-every-tick.

Prints ONE JSON line on rank 0 (see the contract in the task description),
including `roofline` (HBM, algorithmic bytes / measured kernel time) and
`cpu_baseline` (the CPU oracle timed on the host cores, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only does dmabuf IPC (RCCL across processes)

H = 1.0 / 60.0
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
BYTES_PER_BODY_STEP = {"free": 30, "plane": 33,    # reals; SURVEY.md 8(d)
                       "convex": 33 + 2 * 33}          # + the 8 x 4 + 1 contact slots np_convex_plane writes and the step reads back


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[] index + 1 (4 = configs[3]: the configs[1] scene, one 1 Mi-body slab per GPU, i.e. what --gpus N runs; the same as 2)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--side", type=int, default=0, help="grid side (bodies = side^2 per GPU); 0 = config default")
    ap.add_argument("--exchange", default="boundary", choices=["boundary", "none"])
    ap.add_argument("--graph-steps", type=int, default=16, help="ticks per captured HIP graph when exchanging (0 = eager)")
    ap.add_argument("--no-body-collisions", action="store_true", help="skip the body-body broadphase proof (caller asserts single-body islands)")
    ap.add_argument("--exchange-every-tick", action="store_true",
                    help="all-gather the boundary rows every tick even where the collision proof only needs them at chunk ends")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 ranks all on cuda:0 with the collectives staged through gloo: a functional rehearsal of the "
                         "multi-GPU loop on a one-GPU box, not a measurement")
    ap.add_argument("--ticks-per-launch", type=int, default=1,
                    help="contact-free ticks fused into one launch for the HEADLINE run (default 1 = one launch per tick; "
                         "the roofline object then counts one launch's ticks, so frac can exceed 1: temporal reuse)")
    ap.add_argument("--fused-ticks", type=int, default=32,
                    help="also time the contact-free scene with this many ticks per launch (extra 'fused' object; 1 = skip)")
    ap.add_argument("--force-exchange", action="store_true", help="run the exchange path even with one rank (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gyro", type=int, default=2, choices=[0, 1, 2], help="0 off, 1 explicit, 2 implicit (ODE default)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def pmc_traffic(kind, dtype, n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), if one
    exists for this workload: (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction of MI355X_MICROARCH.md."""
    path = os.path.join(ROOT, "profiles", f"hbm_pmc_{kind}_{dtype}_{n}.json")
    try:
        return json.load(open(path))["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(pkg, scene, dtype, kind, budget_s):
    """Time the CPU oracle (oracle/, the checker) on a bounded sample of the same workload."""
    from oracle.orc_ctypes import Oracle
    orc = Oracle(dtype)
    ow = orc.world()
    if scene.plane is not None:
        ow.add_plane(*scene.plane)
    if scene.hull_points is not None:
        ow.set_hull(scene.hull_points)
        ow.add_convex(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia)
    else:
        ow.add_boxes(scene.pos, scene.quat, scene.lvel, scene.avel, scene.mass[:, 0], scene.inertia, scene.sides)
    t = ow.run(H, 2)                                   # probe
    steps = int(max(2, min(2000, budget_s / max(t / 2, 1e-9))))
    t = ow.run(H, steps)
    import ctypes.util
    have_ode = ctypes.util.find_library("ode") is not None      # say whether genuine ODE is on this host (it is never faked)
    return {"value": scene.n * steps / t, "unit": "body-steps/s", "cores": 1, "kind": "port",
            "sample": f"{scene.n} bodies x {steps} steps of the same scene ({kind}), oracle/ C restatement "
                      f"-O2 single thread, {t:.1f} s",
            "system_libode": have_ode}


def cpu_baseline_all_cores(scene, dtype, kind, budget_s):
    """The same oracle, one independent slice of the scene per host core (islands are independent), one child
    process per core (oracle/cpu_worker.py); None if a child fails or overruns."""
    import subprocess
    import tempfile
    cores = max(1, min(os.cpu_count() or 1, 64))
    per = scene.n // cores
    plane = np.array(scene.plane if scene.plane is not None else [], dtype=np.float64)
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for c in range(cores):
            sl = slice(c * per, (c + 1) * per)
            path = os.path.join(tmp, f"slice{c}.npz")
            np.savez(path, pos=scene.pos[sl], quat=scene.quat[sl], lvel=scene.lvel[sl], avel=scene.avel[sl],
                     mass=scene.mass[sl, 0], inertia=scene.inertia[sl], sides=scene.sides[sl], plane=plane,
                     hull=scene.hull_points if scene.hull_points is not None else np.zeros((0, 3)))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), path, dtype,
                                           str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        rate = 0.0
        for p in procs:
            try:
                out, _ = p.communicate(timeout=5 * budget_s + 90)
                n, t = out.split()
                rate += float(n) / float(t)
            except Exception:      # noqa: BLE001 -- a baseline line must never take the bench down
                for q in procs:
                    if q.poll() is None:
                        q.kill()
                return None
    return {"value": rate, "unit": "body-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes x {per} bodies of the same scene ({kind}), each the single-thread oracle for "
                      f"~{budget_s:.0f} s; rates summed"}


def main():
    a = parse()
    # stdout carries exactly one line, the JSON: native libraries print banners there (RCCL's version block at
    # communicator set-up), so fd 1 points at stderr until the result is ready
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(1)
    if a.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or a.force_exchange
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if a.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            import datetime
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(minutes=5))      # a wedged collective fails the run instead of hanging it

    dtype = "float32" if a.dtype == "f32" else "float64"
    rsize = np.dtype(dtype).itemsize
    if a.config in (2, 4):
        side = a.side or 1024
        kind = "free"
        workload = (f"configs[1]: {side * side} free-falling boxes per GPU, no contacts, dt=1/60"
                    + (f" (configs[3] layout: {world} disjoint slabs 10 m apart, one per GPU)" if world > 1 or a.config == 4 else ""))
        # every rank draws its own slab with its own seed; slabs are disjoint islands (configs[3] layout)
        scene = pkg.scenes.box_grid(side, side, seed=1 + rank, spin=True, plane=False).astype(dtype)
    elif a.config == 5:
        side = a.side or 128
        kind = "convex"
        workload = (f"configs[4]: {side * side} convex hulls of res/teapot.obj (1 265 points, scale 0.01) per GPU on the "
                    f"ground plane, convex-plane contacts (<= 8 per hull), 20 SOR iterations, dt=1/60")
        gold = np.load(os.path.join(ROOT, "tests", "golden", "teapot_hull.npz"))     # the hull's vertices (data fixture)
        hull = pkg.hull.build(gold["points"], 0.01)
        scene = pkg.scenes.hull_grid(hull, side, side, seed=1 + rank, y_range=(0.6, 1.6), spin=False, tilt=0.2).astype(dtype)
    else:
        side = a.side or 512
        kind = "plane"
        workload = f"configs[2]: {side * side} boxes on the ground plane per GPU, 20 SOR iterations, dt=1/60"
        scene = pkg.scenes.box_grid(side, side, seed=1 + rank, y_range=(1.0, 3.0), spin=False, plane=True).astype(dtype)

    # configs[3] layout: rank r's slab sits r slab-depths (+ 10 m) further along z, so the slabs are disjoint islands and
    # a neighbour's boundary row (this rank's ghosts) lies >= 10 m beyond this rank's own last row
    scene.pos[:, 2] += rank * (side * (pkg.scenes.HULL_PITCH if kind == "convex" else pkg.scenes.PITCH) + 10.0)
    layout = pkg.shard.SlabLayout(side, side)
    exchanging = use_dist and a.exchange == "boundary"
    w = pkg.BatchWorld(layout.n_total if exchanging else scene.n, dtype=dtype, device=local_rank)
    w.load_scene(scene)
    if exchanging:
        w.set_active_count(scene.n)         # the slots behind are ghosts of the neighbours' boundary rows
    w.set_gyro_mode(a.gyro)
    if a.ticks_per_launch > 1:
        w.set_ticks_per_launch(a.ticks_per_launch)
    collide = not a.no_body_collisions
    if not collide:
        w.set_body_collisions(False)
    stream = torch.cuda.Stream()            # a real (non-null) stream: the batch launches on it and the
    torch.cuda.set_stream(stream)           # timing events below are recorded on it, so they bracket the kernels
    assert stream.cuda_stream != 0
    w.set_stream(stream.cuda_stream)

    forced_ops = None
    if a.rehearse_on_one_gpu and exchanging:
        forced_ops = pkg.shard.StagedDeviceOps(w, torch.device("cuda", 0), stream)
    if a.force_exchange and world == 1 and exchanging:
        # one-rank group: the collective degenerates to a copy, every other step of the path is exercised
        forced_ops = pkg.shard.DeviceOps(w, torch.device("cuda", local_rank), stream)
    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(with_exchange):
        """build the tick loop, warm it up, time a.steps ticks; -> (stepper, graphed, dt, dev_ms, n_exchanges)"""
        st = pkg.shard.ShardedStepper(w, layout, rank, world,
                                      exchange=a.exchange if with_exchange else "none", device=torch.device("cuda", local_rank),
                                      stream=stream, collide=collide and with_exchange,
                                      geometry=(scene.sides, scene.gtype), ops=forced_ops if with_exchange else None,
                                      exchange_every_tick=a.exchange_every_tick)
        if a.config in (3, 5):
            st.run(H, 120)                      # let the boxes land: timed steps are all in contact
        st.run(H, a.warmup)
        if with_exchange and os.environ.get("BENCH_INJECT_EXCHANGE_FAILURE"):
            raise RuntimeError("injected by BENCH_INJECT_EXCHANGE_FAILURE (tests the N>1 safety net)")
        graphed = False
        if st.exchange is not None and a.graph_steps > 0:
            torch.cuda.synchronize()
            graphed = st.capture(H, a.graph_steps, stream)
            st.run(H, a.graph_steps)            # one replay outside the timed region
        fence()
        ex0 = st.exchange.count if st.exchange is not None else 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        st.run(H, a.steps)
        st.drain()                              # the last tick's exchange is part of the timed work
        e1.record(stream)
        fence()
        dt = time.perf_counter() - t0
        n_ex = (a.steps if graphed else st.exchange.count - ex0) if st.exchange is not None else 0
        return st, graphed, dt, e0.elapsed_time(e1), n_ex

    exchange_fallback = None
    try:
        stepper, graphed, dt, dev_ms, n_exchanges = measure(exchanging)
    except Exception as e:      # noqa: BLE001
        # Safety net for the N>1 run only (it cannot be rehearsed on real multi-GPU RCCL before the driver runs it): if the
        # exchanging loop raises -- in rank-symmetric code, so on every rank -- the slabs, which are disjoint islands 10 m
        # apart, are timed without the exchange instead, and the JSON line says so.  Never taken at N=1.
        if not (world > 1 and exchanging):
            raise
        exchange_fallback = f"{type(e).__name__}: {e}"
        print(f"bench.py[rank {rank}]: exchanging loop failed ({exchange_fallback}); timing the slabs without the exchange",
              file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        w.upload_geom_type(np.zeros(layout.n_total - scene.n, np.uint8), first=scene.n)    # ghost slots: inert again
        stepper, graphed, dt, dev_ms, n_exchanges = measure(False)

    def run(nsteps):
        stepper.run(H, nsteps)


    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if a.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    stats = w.collision_stats()
    total_bodies = scene.n * world
    value = total_bodies * a.steps / dt
    tpl = a.ticks_per_launch if (kind == "free" and a.ticks_per_launch > 1) else 1
    kernel_s = dev_ms * 1e-3 / a.steps * tpl                 # average launch duration, HIP events on the launch stream
    alg_bytes = BYTES_PER_BODY_STEP[kind] * rsize * scene.n * tpl   # per launch (one GPU): a launch takes tpl ticks
    achieved = alg_bytes / kernel_s / 1e9
    out = {
        "metric": "body-steps/sec at 1M rigid bodies, dt=1/60",
        "value": value, "unit": "body-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, collectives staged through host memory; not a measurement)" if a.rehearse_on_one_gpu else ""),
        "config": {"workload": workload, "bodies_per_gpu": scene.n, "bodies_total": total_bodies, "dt": "1/60",
                   "parallelism": f"islands sharded over {world} GPU(s), one slab per rank; "
                                  + (f"boundary rows all-gathered over RCCL on a side stream, overlapped with the next tick"
                                     f"{', HIP-graph replay' if graphed else ''}: {n_exchanges} exchanges in the {a.steps} timed ticks "
                                     f"({'every tick' if n_exchanges >= a.steps else 'at the end of every collision-proof chunk: inside a ballistic chunk nothing reads the ghost rows'})"
                                     if stepper.exchange is not None
                                     else "no exchange (one rank)" if world == 1 else
                                     "no exchange" + (f" (FALLBACK: the exchanging loop raised {exchange_fallback})" if exchange_fallback else "")),
                   "collide": ("body-body pairs: none by assertion (check off)" if a.no_body_collisions else
                               f"body-body pairs proven absent per tick by broadphase safe zones ({stats['fast_ticks']} fast ticks, "
                               f"{stats['careful_ticks']} exact-search ticks, {stats['rebuilds']} zone rebuilds, {stats['pair_ticks']} ticks with pairs)")
                              + ("; ground plane fused into the step kernel" if kind == "plane" else "")
                              + ("; convex-plane narrowphase: one wavefront per hull, then the fused step with 8 contact slots" if kind == "convex" else ""),
                   "integrator": "QuickStep semantics: gravity + implicit gyroscopic torque + semi-implicit Euler + "
                                 "quaternion renormalise" + ("; box-plane contacts, 20 SOR sweeps" if kind == "plane" else "")
                                 + ("; convex-plane contacts, 20 SOR sweeps" if kind == "convex" else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(kind, a.dtype, scene.n),
                     "kernel": {"free": "integrate_free", "plane": "step_plane", "convex": "np_convex_plane + step_plane<8>"}[kind],
                     "kernel_us": kernel_s * 1e6,
                     "algorithmic_bytes_per_body_step": BYTES_PER_BODY_STEP[kind] * rsize,
                     "ticks_per_launch": tpl},
    }
    if exchange_fallback:
        out["exchange_fallback"] = exchange_fallback
    if kind == "free" and stepper.exchange is None and a.fused_ticks > 1 and tpl == 1:
        # the same scene and step count with several ticks per launch (state in registers between ticks; results are
        # bit-identical, tests/test_gpu_parity.py).  Reported beside the headline, which stays one launch per tick.
        w.set_ticks_per_launch(a.fused_ticks)
        run(a.warmup)
        fence()
        t0 = time.perf_counter()
        run(a.steps)
        fence()
        dtf = time.perf_counter() - t0
        w.set_ticks_per_launch(1)
        out["fused"] = {"ticks_per_launch": a.fused_ticks, "value": total_bodies * a.steps / dtf, "unit": "body-steps/s",
                        "ms_per_step": dtf * 1e3 / a.steps,
                        "algorithmic_GBps": alg_bytes * a.steps / dtf / 1e9,
                        "note": "one read + one write of the state per launch instead of per tick: past the per-tick HBM "
                                "roofline, bounded by VALU issue"}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, scene, dtype, kind, a.cpu_seconds)
        allc = cpu_baseline_all_cores(scene, dtype, kind, min(6.0, a.cpu_seconds))
        if allc is not None:
            out["cpu_baseline_all_cores"] = allc
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)
    w.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
