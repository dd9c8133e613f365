"""BatchWorld: Python view of one dmxBatch (include/dmx_batch.h).

Mirrors the reference's physics-facing calls in batch form: world setup
(main.c:94-98), AddBody (main.c:695-733), the tick loop (main.c:211-215) and
the pose read-back (main.c:221-237), all through the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib

DMX_F32, DMX_F64 = 0, 1
POS, QUAT, LVEL, AVEL, MASS, INERTIA, SIDES, FORCE, TORQUE = range(9)
QUAT_RAW, STATE = 9, 10      # quaternion stored as given; pos3 quat4 lvel3 avel3 in one piece
_K = {POS: 3, QUAT: 4, LVEL: 3, AVEL: 3, MASS: 1, INERTIA: 3, SIDES: 3, FORCE: 3, TORQUE: 3, QUAT_RAW: 4, STATE: 13}
GEOM_NONE, GEOM_SPHERE, GEOM_BOX = 0, 1, 2
GYRO_OFF, GYRO_EXPLICIT, GYRO_IMPLICIT = 0, 1, 2
CONTACT_BOUNCE = 0x004
SNAPSHOT_PINGPONG, SNAPSHOT_COPY = 0, 1
EXACT_AUTO, EXACT_STAGED, EXACT_ONE_WORKGROUP = 0, 1, 2


class DmxError(RuntimeError):
    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


def _check(rc, what):
    if rc != 0:
        raise DmxError(f"{what} failed with code {rc}", rc)


class BatchWorld:
    def __init__(self, n_bodies, dtype="float32", device=0, gravity=(0.0, -9.8, 0.0)):
        self.lib = _lib.load()
        self.dtype = np.dtype(dtype)
        prec = {4: DMX_F32, 8: DMX_F64}[self.dtype.itemsize]
        self.n = int(n_bodies)
        h = C.c_void_p()
        _check(self.lib.dmxBatchCreate(C.byref(h), self.n, prec, device), "dmxBatchCreate")
        self.h = h
        self.set_gravity(*gravity)       # dWorldSetGravity(world, 0, -9.8, 0)  main.c:96

    # -- lifecycle -----------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.dmxBatchDestroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- parameters ------------------------------------------------------------
    def set_gravity(self, x, y, z):
        _check(self.lib.dmxBatchSetGravity(self.h, x, y, z), "dmxBatchSetGravity")

    def set_erp(self, erp):
        _check(self.lib.dmxBatchSetERP(self.h, erp), "dmxBatchSetERP")

    def set_cfm(self, cfm):
        _check(self.lib.dmxBatchSetCFM(self.h, cfm), "dmxBatchSetCFM")

    def set_quickstep(self, iters, sor_w=1.3):
        _check(self.lib.dmxBatchSetQuickStep(self.h, iters, sor_w), "dmxBatchSetQuickStep")

    def set_gyro_mode(self, mode):
        _check(self.lib.dmxBatchSetGyroMode(self.h, mode), "dmxBatchSetGyroMode")

    def set_surface(self, mode=CONTACT_BOUNCE, mu=float("inf"), bounce=0.2, bounce_vel=0.1):
        _check(self.lib.dmxBatchSetSurface(self.h, mode, mu, bounce, bounce_vel), "dmxBatchSetSurface")

    def set_max_contacts(self, n):
        _check(self.lib.dmxBatchSetMaxContacts(self.h, n), "dmxBatchSetMaxContacts")

    def set_plane(self, a, b, c, d, enable=True):
        _check(self.lib.dmxBatchSetPlane(self.h, a, b, c, d, int(enable)), "dmxBatchSetPlane")

    # -- data ----------------------------------------------------------------------
    def upload(self, field, arr, first=0):
        a = np.ascontiguousarray(arr, dtype=self.dtype).reshape(-1, _K[field])
        _check(self.lib.dmxBatchUpload(self.h, field, a.ctypes.data, first, a.shape[0]), "dmxBatchUpload")

    def download(self, field, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.empty((count, _K[field]), self.dtype)
        _check(self.lib.dmxBatchDownload(self.h, field, out.ctypes.data, first, count), "dmxBatchDownload")
        return out

    def upload_geom_type(self, types, first=0):
        t = np.ascontiguousarray(types, dtype=np.uint8)
        _check(self.lib.dmxBatchUploadGeomType(self.h, t.ctypes.data, first, t.shape[0]), "dmxBatchUploadGeomType")

    def set_static_boxes(self, boxes):
        """AddBodyMap (main.c:735-761): `boxes` = [(sides3, pos3, R12)], the static floor / walls, e.g. scenes.reference_map()"""
        n = len(boxes)
        sides = np.ascontiguousarray([b[0] for b in boxes], dtype=np.float64).reshape(n, 3)
        pos = np.ascontiguousarray([b[1] for b in boxes], dtype=np.float64).reshape(n, 3)
        rot = np.ascontiguousarray([b[2] for b in boxes], dtype=np.float64).reshape(n, 12)
        _check(self.lib.dmxBatchSetStaticBoxes(self.h, n, sides.ctypes.data, pos.ctypes.data, rot.ctypes.data),
               "dmxBatchSetStaticBoxes")

    def set_convex_hull(self, points):
        """Body-frame points of the hull every GEOM_CONVEX body uses; returns the hull's bounding radius (upload it as
        sides[:, 0] of the convex bodies)."""
        p = np.ascontiguousarray(points, dtype=np.float64)
        r = C.c_double()
        _check(self.lib.dmxBatchSetConvexHull(self.h, p.shape[0], p.ctypes.data, C.byref(r)), "dmxBatchSetConvexHull")
        return r.value

    def set_convex_hull_faces(self, planes):
        """the hull's faces (nf x 4: unit outward normal, offset; body frame), e.g. hull.planes(points)"""
        pl = np.ascontiguousarray(planes, dtype=np.float64).reshape(-1, 4)
        _check(self.lib.dmxBatchSetConvexHullFaces(self.h, pl.shape[0], pl.ctypes.data), "dmxBatchSetConvexHullFaces")

    def load_scene(self, scene):
        """Upload a scenes.Scene (the batch form of the AddBody loop, main.c:695-733)."""
        self.upload(POS, scene.pos)
        self.upload(QUAT, scene.quat)
        self.upload(LVEL, scene.lvel)
        self.upload(AVEL, scene.avel)
        self.upload(MASS, scene.mass)
        self.upload(INERTIA, scene.inertia)
        if getattr(scene, "hull_points", None) is not None:
            self.set_convex_hull(scene.hull_points)
            if getattr(scene, "hull_planes", None) is not None:
                self.set_convex_hull_faces(scene.hull_planes)
        self.upload(SIDES, scene.sides)
        self.upload_geom_type(scene.gtype)
        if scene.plane is not None:
            self.set_plane(*scene.plane, enable=True)
        if getattr(scene, "static_boxes", None):
            self.set_static_boxes(scene.static_boxes)

    # -- checkpoint / resume ---------------------------------------------------------------------------------
    def checkpoint(self):
        """The per-body data a later `restore` needs to continue bit for bit: the 13-real state, the force / torque
        accumulators and the per-body constants (mass, inertia, extents).  NOT in it: geometry classes, the hull, static
        boxes, plane, gravity, solver and surface parameters, active count -- a fresh batch is set up from the scene first,
        then restored.  (The reference keeps no resumable state -- its 60 Hz snapshot holds poses only, SURVEY section 5 --
        so this is new surface, built on Download.)"""
        self.synchronize()
        return {"state": self.download(STATE), "force": self.download(FORCE), "torque": self.download(TORQUE),
                "mass": self.download(MASS), "inertia": self.download(INERTIA), "sides": self.download(SIDES)}

    def restore(self, ckpt):
        self.upload(MASS, ckpt["mass"]); self.upload(INERTIA, ckpt["inertia"]); self.upload(SIDES, ckpt["sides"])
        self.upload(STATE, ckpt["state"])                    # stored as given: no renormalisation of the quaternions
        # always, zeros included: accumulators added to the live batch since the checkpoint must not act after the restore
        self.upload(FORCE, ckpt["force"]); self.upload(TORQUE, ckpt["torque"])

    def state(self):
        return (self.download(POS), self.download(QUAT), self.download(LVEL), self.download(AVEL))

    def device_ptr(self, field, comp):
        return self.lib.dmxBatchDevicePtr(self.h, field, comp)

    @property
    def stride(self):
        return self.lib.dmxBatchStride(self.h)

    # -- stepping ----------------------------------------------------------------------
    def step(self, h, nsteps=1):
        _check(self.lib.dmxBatchStep(self.h, h, nsteps), "dmxBatchStep")

    def set_body_collisions(self, enable):
        _check(self.lib.dmxBatchSetBodyCollisions(self.h, int(enable)), "dmxBatchSetBodyCollisions")

    def collision_stats(self):
        out = (C.c_int64 * 6)()
        _check(self.lib.dmxBatchCollisionStats(self.h, out), "dmxBatchCollisionStats")
        d = dict(zip(("fast_ticks", "careful_ticks", "rebuilds", "pair_ticks", "last_pairs", "crowded"), out))
        ex = (C.c_int64 * 8)()
        _check(self.lib.dmxBatchCollisionStatsEx(self.h, ex), "dmxBatchCollisionStatsEx")
        d["unsupported_pairs"] = ex[6]
        d["speculated_ticks"] = ex[7]
        return d

    # -- the collision-checked loop in pieces (include/dmx_batch.h), for callers with per-tick work of their own --
    def chunk_begin(self):
        """-> (exact_only, ballistic)"""
        e, bl = C.c_int(), C.c_int()
        _check(self.lib.dmxBatchChunkBegin(self.h, C.byref(e), C.byref(bl)), "dmxBatchChunkBegin")
        return bool(e.value), bool(bl.value)

    def chunk_tick(self, h, check=True):
        _check(self.lib.dmxBatchChunkTick(self.h, h, int(check)), "dmxBatchChunkTick")

    def chunk_ticks(self, h, nticks, check_first=True, check_last=True):
        _check(self.lib.dmxBatchChunkTicks(self.h, h, nticks, int(check_first), int(check_last)), "dmxBatchChunkTicks")

    def set_snapshot_mode(self, mode):
        """SNAPSHOT_PINGPONG (default) / SNAPSHOT_COPY: how a chunk keeps its start state (include/dmx_batch.h)"""
        _check(self.lib.dmxBatchSetSnapshotMode(self.h, mode), "dmxBatchSetSnapshotMode")

    def set_exact_pipeline(self, mode):
        """EXACT_AUTO (default) / EXACT_STAGED / EXACT_ONE_WORKGROUP: how an exact tick runs its bookkeeping (include/dmx_batch.h)"""
        _check(self.lib.dmxBatchSetExactPipeline(self.h, mode), "dmxBatchSetExactPipeline")

    def set_class_pairs(self, class_a, class_b, enable):
        """whether bodies of two geometry classes collide with one another (the batch's form of ODE's category / collide bits)"""
        _check(self.lib.dmxBatchSetClassPairs(self.h, int(class_a), int(class_b), 1 if enable else 0), "dmxBatchSetClassPairs")

    def set_static_path(self, fused=True):
        """bodies at static boxes: the fused path (default) or the exact tick for every one of them (include/dmx_batch.h)"""
        _check(self.lib.dmxBatchSetStaticPath(self.h, 1 if fused else 0), "dmxBatchSetStaticPath")

    def set_ticks_per_launch(self, ticks):
        _check(self.lib.dmxBatchSetTicksPerLaunch(self.h, ticks), "dmxBatchSetTicksPerLaunch")

    def check_zones_on(self, stream_handle, first, count):
        _check(self.lib.dmxBatchCheckZonesOnStream(self.h, stream_handle, first, count), "dmxBatchCheckZonesOnStream")

    def refresh_ghosts_on(self, stream_handle, first, count_lo, src_lo, count_hi, src_hi, check):
        _check(self.lib.dmxBatchRefreshGhostsOnStream(self.h, stream_handle, first, count_lo, src_lo, count_hi, src_hi, int(check)),
               "dmxBatchRefreshGhostsOnStream")

    def chunk_end(self):
        """-> (violated, warn); waits for the batch stream"""
        v, w = C.c_int(), C.c_int()
        _check(self.lib.dmxBatchChunkEnd(self.h, C.byref(v), C.byref(w)), "dmxBatchChunkEnd")
        return bool(v.value), bool(w.value)

    def chunk_commit(self, ticks, refresh_zones=False):
        _check(self.lib.dmxBatchChunkCommit(self.h, ticks, int(refresh_zones)), "dmxBatchChunkCommit")

    def chunk_rollback(self):
        _check(self.lib.dmxBatchChunkRollback(self.h), "dmxBatchChunkRollback")

    def exact_tick(self, h):
        _check(self.lib.dmxBatchExactTick(self.h, h), "dmxBatchExactTick")

    def find_pairs(self):
        """dSpaceCollide's pair search alone: -> (pairs [np, 2], involved [ni], cross [nc, 2] = (own body, ghost slot))"""
        pp, ip, cp = C.c_void_p(), C.c_void_p(), C.c_void_p()
        npairs, ninv, nc = C.c_int64(), C.c_int64(), C.c_int64()
        _check(self.lib.dmxBatchFindPairs(self.h, C.byref(pp), C.byref(npairs), C.byref(ip), C.byref(ninv)), "dmxBatchFindPairs")
        _check(self.lib.dmxBatchCrossPairs(self.h, C.byref(cp), C.byref(nc)), "dmxBatchCrossPairs")
        grab = lambda p, n: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n,)).copy() if n else np.zeros(0, np.int32)
        return (grab(pp, 2 * npairs.value).reshape(-1, 2), grab(ip, ninv.value), grab(cp, 2 * nc.value).reshape(-1, 2))

    def set_active_count(self, n_active):
        _check(self.lib.dmxBatchSetActiveCount(self.h, n_active), "dmxBatchSetActiveCount")

    def step_range(self, h, first, count, reset_diag=False):
        _check(self.lib.dmxBatchStepRange(self.h, h, first, count, int(reset_diag)), "dmxBatchStepRange")

    def gather_bodies(self, idx_ptr, count, out_ptr):
        _check(self.lib.dmxBatchGatherBodies(self.h, idx_ptr, count, out_ptr), "dmxBatchGatherBodies")

    def scatter_bodies(self, idx_ptr, count, in_ptr):
        _check(self.lib.dmxBatchScatterBodies(self.h, idx_ptr, count, in_ptr), "dmxBatchScatterBodies")

    def scatter_bodies_on(self, stream_handle, idx_ptr, count, in_ptr):
        _check(self.lib.dmxBatchScatterBodiesOnStream(self.h, idx_ptr, count, in_ptr, stream_handle), "dmxBatchScatterBodiesOnStream")

    def set_boundary_pack(self, out_ptr, lo_count, hi_first):
        _check(self.lib.dmxBatchSetBoundaryPack(self.h, out_ptr, lo_count, hi_first), "dmxBatchSetBoundaryPack")

    def step_timed(self, h, nsteps):
        ms = C.c_float()
        _check(self.lib.dmxBatchStepTimed(self.h, h, nsteps, C.byref(ms)), "dmxBatchStepTimed")
        return ms.value

    def synchronize(self):
        _check(self.lib.dmxBatchSynchronize(self.h), "dmxBatchSynchronize")

    def set_stream(self, stream_handle):
        _check(self.lib.dmxBatchSetStream(self.h, stream_handle), "dmxBatchSetStream")

    def last_contact_count(self):
        n = C.c_int64()
        _check(self.lib.dmxBatchLastContactCount(self.h, C.byref(n)), "dmxBatchLastContactCount")
        return n.value

    def last_residual(self):
        r = C.c_double()
        _check(self.lib.dmxBatchLastResidual(self.h, C.byref(r)), "dmxBatchLastResidual")
        return r.value

    def transforms(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.empty((count, 16), self.dtype)
        _check(self.lib.dmxBatchDownloadTransforms(self.h, out.ctypes.data, first, count),
               "dmxBatchDownloadTransforms")
        return out
