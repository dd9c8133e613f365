"""Convex-hull set-up for convex bodies: ctypes binding of include/dmx_hull.h (host code in libode_mi355.so).

The reference has no convex geoms (res/teapot.obj is a render asset); this is the set-up step
BASELINE configs[4] needs: OBJ vertices -> convex hull -> mass properties -> body-frame points for
BatchWorld.set_convex_hull."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib


class _Info(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_faces", C.c_int32), ("volume", C.c_double), ("area", C.c_double),
                ("com", C.c_double * 3), ("axes", C.c_double * 9), ("inertia", C.c_double * 3), ("radius", C.c_double)]


@dataclass
class Hull:
    points: np.ndarray      # n x 3, body frame (origin = centre of mass, axes = principal axes)
    index: np.ndarray       # n, index of each hull vertex in the input point list
    n_faces: int
    volume: float
    area: float
    com: np.ndarray         # centre of mass in the (scaled) input frame
    axes: np.ndarray        # 3 x 3, rows = principal axes in the input frame
    inertia: np.ndarray     # principal moments about the centre of mass, unit density
    radius: float           # bounding radius about the centre of mass

    def upright_quaternion(self):
        """(w,x,y,z) of the body orientation that puts the hull back in the input frame's orientation:
        body -> world rotation = axes^T."""
        Rm = self.axes.T
        tr = np.trace(Rm)
        if tr >= 0:
            s = np.sqrt(tr + 1.0)
            w = 0.5 * s
            s = 0.5 / s
            q = np.array([w, (Rm[2, 1] - Rm[1, 2]) * s, (Rm[0, 2] - Rm[2, 0]) * s, (Rm[1, 0] - Rm[0, 1]) * s])
        else:                              # a principal frame is chosen close to the input axes: never taken in practice
            i = int(np.argmax(np.diag(Rm)))
            j, k = (i + 1) % 3, (i + 2) % 3
            s = np.sqrt(Rm[i, i] - Rm[j, j] - Rm[k, k] + 1.0)
            q = np.zeros(4)
            q[1 + i] = 0.5 * s
            s = 0.5 / s
            q[0] = (Rm[k, j] - Rm[j, k]) * s
            q[1 + j] = (Rm[j, i] + Rm[i, j]) * s
            q[1 + k] = (Rm[k, i] + Rm[i, k]) * s
        return q / np.linalg.norm(q)


def _bind():
    lib = _lib.load()
    lib.dmxObjReadVertices.restype = C.c_int64
    lib.dmxObjReadVertices.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
    lib.dmxHullBuild.restype = C.c_int32
    lib.dmxHullBuild.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(_Info)]
    return lib


def read_obj_vertices(path):
    lib = _bind()
    n = lib.dmxObjReadVertices(str(path).encode(), None, 0)
    if n < 0:
        raise OSError(f"cannot read {path}")
    v = np.zeros((n, 3))
    lib.dmxObjReadVertices(str(path).encode(), v.ctypes.data, n)
    return v


def planes(body_points):
    """face planes (nf x 4: unit outward normal, offset) of the hull of `body_points` (a Hull's .points)"""
    lib = _bind()
    lib.dmxHullPlanes.restype = C.c_int32
    lib.dmxHullPlanes.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
    p = np.ascontiguousarray(body_points, dtype=np.float64)
    nf = lib.dmxHullPlanes(p.ctypes.data, p.shape[0], None, 0)
    if nf < 0:
        raise ValueError(f"dmxHullPlanes failed with code {nf}")
    out = np.zeros((nf, 4))
    lib.dmxHullPlanes(p.ctypes.data, p.shape[0], out.ctypes.data, nf)
    return out


def build(points, scale=1.0):
    """Convex hull + solid mass properties of `points` (n x 3) scaled by `scale`."""
    lib = _bind()
    p = np.ascontiguousarray(points, dtype=np.float64)
    n = p.shape[0]
    out = np.zeros((n, 3))
    idx = np.zeros(n, np.int32)
    info = _Info()
    nv = lib.dmxHullBuild(p.ctypes.data, n, float(scale), out.ctypes.data, idx.ctypes.data, n, C.byref(info))
    if nv < 0:
        raise ValueError(f"dmxHullBuild failed with code {nv} (degenerate point set?)")
    return Hull(out[:nv].copy(), idx[:nv].copy(), info.n_faces, info.volume, info.area, np.array(info.com),
                np.array(info.axes).reshape(3, 3), np.array(info.inertia), info.radius)
