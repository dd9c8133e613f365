"""ctypes binding of the CPU oracle (oracle/liboracle_f{32,64}.so).

TEST INFRASTRUCTURE ONLY (see oracle/orc.h): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the
product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

GEOM_SPHERE, GEOM_BOX, GEOM_PLANE = 0, 1, 2
CONTACT_BOUNCE = 0x004
ORDER_FIXED, ORDER_ODE = 0, 1
GYRO_OFF, GYRO_EXPLICIT, GYRO_IMPLICIT = 0, 1, 2


def build():
    """Compile the oracle's C restatement (gcc, a few seconds)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


class _ContactGeom64(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("normal", C.c_double * 3),
                ("depth", C.c_double), ("g1", C.c_int), ("g2", C.c_int)]


class _ContactGeom32(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("normal", C.c_float * 3),
                ("depth", C.c_float), ("g1", C.c_int), ("g2", C.c_int)]


class Oracle:
    """One precision of the oracle library."""

    def __init__(self, dtype):
        self.dtype = np.dtype(dtype)
        name = {4: "liboracle_f32.so", 8: "liboracle_f64.so"}[self.dtype.itemsize]
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        self.lib = lib = C.CDLL(path)
        self.real = real = C.c_float if self.dtype.itemsize == 4 else C.c_double
        self.ContactGeom = _ContactGeom32 if self.dtype.itemsize == 4 else _ContactGeom64
        P = C.c_void_p
        RP = C.POINTER(real)

        def sig(fn, res, *args):
            f = getattr(lib, fn)
            f.restype = res
            f.argtypes = list(args)

        sig("orc_world_create", P)
        sig("orc_world_destroy", None, P)
        sig("orc_world_set_gravity", None, P, real, real, real)
        sig("orc_world_set_erp", None, P, real)
        sig("orc_world_set_cfm", None, P, real)
        sig("orc_world_set_quickstep", None, P, C.c_int, real)
        sig("orc_world_set_row_order", None, P, C.c_int)
        sig("orc_world_set_gyro_mode", None, P, C.c_int)
        sig("orc_world_set_stepper", None, P, C.c_int)
        sig("orc_world_last_lcp_rounds", C.c_int, P)
        sig("orc_world_joint_count", C.c_int, P)
        sig("orc_world_joint_info", None, P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), RP, RP, RP, RP)
        sig("orc_world_set_surface", None, P, C.c_int, real, real, real)
        sig("orc_world_set_max_contacts", None, P, C.c_int)
        sig("orc_world_set_broadphase", None, P, C.c_int)
        sig("orc_rand_seed", None, C.c_uint32)
        sig("orc_body_create", C.c_int, P)
        sig("orc_body_set_position", None, P, C.c_int, real, real, real)
        sig("orc_body_set_rotation", None, P, C.c_int, RP)
        sig("orc_body_set_quaternion", None, P, C.c_int, RP)
        sig("orc_body_set_linear_vel", None, P, C.c_int, real, real, real)
        sig("orc_body_set_angular_vel", None, P, C.c_int, real, real, real)
        sig("orc_body_set_mass", None, P, C.c_int, real, RP)
        sig("orc_body_add_force", None, P, C.c_int, real, real, real)
        sig("orc_body_add_torque", None, P, C.c_int, real, real, real)
        for g in ("position", "quaternion", "linear_vel", "angular_vel", "rotation"):
            sig("orc_body_get_" + g, RP, P, C.c_int)
        sig("orc_geom_create_box", C.c_int, P, real, real, real)
        sig("orc_geom_create_sphere", C.c_int, P, real)
        sig("orc_geom_create_plane", C.c_int, P, real, real, real, real)
        sig("orc_geom_set_body", None, P, C.c_int, C.c_int)
        sig("orc_geom_set_position", None, P, C.c_int, real, real, real)
        sig("orc_geom_set_rotation", None, P, C.c_int, RP)
        sig("orc_geom_set_category_bits", None, P, C.c_int, C.c_uint32)
        sig("orc_geom_set_collide_bits", None, P, C.c_int, C.c_uint32)
        sig("orc_collide", C.c_int, P, C.c_int, C.c_int, C.c_int, C.POINTER(self.ContactGeom))
        sig("orc_collide_bulk", None, P, C.c_int, C.c_int, C.c_int, P, P, P, P, C.c_int, P, P)
        sig("orc_world_tick", None, P, real)
        sig("orc_world_last_contact_count", C.c_int, P)
        sig("orc_world_last_body_pairs", C.c_int, P)
        sig("orc_world_last_sor_residual", C.c_double, P)
        sig("orc_world_body_count", C.c_int, P)
        sig("orc_world_add_boxes", None, P, C.c_int, P, P, P, P, P, P, P)
        sig("orc_world_add_spheres", None, P, C.c_int, P, P, P, P, P, P, P)
        sig("orc_world_set_hull", None, P, C.c_int, P)
        sig("orc_world_set_hull_faces", None, P, C.c_int, P)
        sig("orc_geom_create_convex", C.c_int, P)
        sig("orc_world_add_convex", None, P, C.c_int, P, P, P, P, P, P)
        sig("orc_world_get_state", None, P, P, P, P, P)
        sig("orc_world_run", C.c_double, P, real, C.c_int)
        sig("orc_pack_transform", None, RP, RP, RP)
        sig("orc_ref_rand_seed", None, C.c_uint32)
        sig("orc_ref_rand_next", C.c_uint32)
        sig("orc_ref_rand_int", C.c_int32, C.c_int32, C.c_int32)
        sig("orc_ref_rand_double", C.c_double, C.c_double, C.c_double)
        sig("orc_real_size", C.c_int)
        assert lib.orc_real_size() == self.dtype.itemsize

    def arr(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        return a, a.ctypes.data_as(C.POINTER(self.real))

    def world(self, **kw):
        return World(self, **kw)


class World:
    """Thin object wrapper around orc_world."""

    def __init__(self, orc, gravity=(0.0, -9.8, 0.0)):
        self.o = orc
        self.lib = orc.lib
        self.w = self.lib.orc_world_create()
        self.lib.orc_world_set_gravity(self.w, *gravity)   # main.c:96
        self._keep = []

    def close(self):
        if self.w:
            self.lib.orc_world_destroy(self.w)
            self.w = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _p(self, a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=self.o.dtype)
        self._keep.append(a)
        return a.ctypes.data

    def add_reference_scene(self, statics, bodies):
        """the reference's own scene through the oracle's ODE-shaped calls, in the order main.c makes them: static map boxes
        (AddBodyMap, main.c:735-761, with its double dGeomSetCategoryBits) and the spawner's bodies (AddBody, main.c:695-733);
        statics = [(size3, pos3, R12)], bodies = [(type 1 sphere / 2 box, size3, pos3)]"""
        lib, o = self.lib, self.o
        for size, pos, R in statics:
            g = lib.orc_geom_create_box(self.w, *size)
            lib.orc_geom_set_position(self.w, g, *pos)
            _, rp = o.arr(R)
            lib.orc_geom_set_rotation(self.w, g, rp)
            lib.orc_geom_set_category_bits(self.w, g, 0xFFFFFFFE)
        ident = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]
        for kind, size, pos in bodies:
            b = lib.orc_body_create(self.w)
            lib.orc_body_set_position(self.w, b, *pos)
            _, rp = o.arr(ident)
            lib.orc_body_set_rotation(self.w, b, rp)
            g = lib.orc_geom_create_sphere(self.w, size[0]) if kind == 1 else lib.orc_geom_create_box(self.w, *size)
            lib.orc_geom_set_category_bits(self.w, g, 2)
            lib.orc_geom_set_collide_bits(self.w, g, 3)
            lib.orc_geom_set_body(self.w, g, b)

    # bulk ------------------------------------------------------------------
    def add_boxes(self, pos, quat, lvel, avel, mass, idiag, sides):
        n = len(pos)
        self.lib.orc_world_add_boxes(self.w, n, self._p(pos), self._p(quat), self._p(lvel),
                                     self._p(avel), self._p(mass), self._p(idiag), self._p(sides))
        self._keep.clear()

    def add_spheres(self, pos, quat, lvel, avel, mass, idiag, radius):
        n = len(pos)
        self.lib.orc_world_add_spheres(self.w, n, self._p(pos), self._p(quat), self._p(lvel),
                                       self._p(avel), self._p(mass), self._p(idiag), self._p(radius))
        self._keep.clear()

    def set_hull(self, points):
        pts = np.ascontiguousarray(points, dtype=self.o.dtype)
        self.lib.orc_world_set_hull(self.w, len(pts), pts.ctypes.data_as(C.c_void_p))

    def set_hull_faces(self, planes):
        pl = np.ascontiguousarray(planes, dtype=self.o.dtype)
        self.lib.orc_world_set_hull_faces(self.w, len(pl), pl.ctypes.data_as(C.c_void_p))

    def add_convex(self, pos, quat, lvel, avel, mass, idiag):
        n = len(pos)
        self.lib.orc_world_add_convex(self.w, n, self._p(pos), self._p(quat), self._p(lvel),
                                      self._p(avel), self._p(mass), self._p(idiag))
        self._keep.clear()

    def add_plane(self, a, b, c, d):
        return self.lib.orc_geom_create_plane(self.w, a, b, c, d)

    def add_static_box(self, sides, pos, R12):
        """AddBodyMap (main.c:735-761): a body-less box geom; category ~CMASK_MAP, collide ~0 (the double
        SetCategoryBits of main.c:751-752)"""
        g = self.lib.orc_geom_create_box(self.w, *[float(s) for s in sides])
        self.lib.orc_geom_set_position(self.w, g, *[float(p) for p in pos])
        _, rp = self.o.arr(R12)
        self.lib.orc_geom_set_rotation(self.w, g, rp)
        self.lib.orc_geom_set_category_bits(self.w, g, 1)
        self.lib.orc_geom_set_category_bits(self.w, g, 0xFFFFFFFE)
        return g

    def state(self):
        n = self.lib.orc_world_body_count(self.w)
        dt = self.o.dtype
        pos = np.empty((n, 3), dt)
        quat = np.empty((n, 4), dt)
        lvel = np.empty((n, 3), dt)
        avel = np.empty((n, 3), dt)
        self.lib.orc_world_get_state(self.w, pos.ctypes.data, quat.ctypes.data,
                                     lvel.ctypes.data, avel.ctypes.data)
        return pos, quat, lvel, avel

    def run(self, h, steps):
        """steps ticks of main.c:211-215; returns seconds spent in the loop."""
        return self.lib.orc_world_run(self.w, h, steps)

    def tick(self, h):
        self.lib.orc_world_tick(self.w, h)

    def n_contacts(self):
        return self.lib.orc_world_last_contact_count(self.w)

    def n_body_pairs(self):
        return self.lib.orc_world_last_body_pairs(self.w)

    def sor_residual(self):
        return self.lib.orc_world_last_sor_residual(self.w)

    def set_stepper(self, exact):
        """False: dWorldQuickStep (SOR); True: dWorldStep (every island's LCP solved exactly)"""
        self.lib.orc_world_set_stepper(self.w, 1 if exact else 0)

    def joints(self):
        """the last tick's contact joints: [(b1, b2, pos3, normal3, depth, normal force)]"""
        out = []
        real = self.o.real
        for k in range(self.lib.orc_world_joint_count(self.w)):
            b1, b2 = C.c_int(), C.c_int()
            pos, nrm, dep, lam = (real * 3)(), (real * 3)(), real(), real()
            self.lib.orc_world_joint_info(self.w, k, C.byref(b1), C.byref(b2), pos, nrm, C.byref(dep), C.byref(lam))
            out.append((b1.value, b2.value, list(pos), list(nrm), dep.value, lam.value))
        return out
