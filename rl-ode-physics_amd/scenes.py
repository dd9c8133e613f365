"""Synthetic scenes of BASELINE.json's configs, drawn with the reference's PRNG
and spawn distributions (main.c:504-509: y in [20,50], box sides in [0.2,1.0])
on a grid with 2.5 m pitch so that boxes (diagonal <= 1.74 m) never touch one
another and every dynamics island is a single body.

Each body consumes 7 draws, in this order: side x, side y, side z, height y,
omega x, omega y, omega z.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .rand import Rand

GEOM_SPHERE, GEOM_BOX, GEOM_CONVEX = 1, 2, 3
PITCH = 2.5
DRAWS_PER_BODY = 7


@dataclass
class Scene:
    pos: np.ndarray       # n x 3
    quat: np.ndarray      # n x 4 (w,x,y,z)
    lvel: np.ndarray      # n x 3
    avel: np.ndarray      # n x 3
    mass: np.ndarray      # n x 1
    inertia: np.ndarray   # n x 3 (body-frame principal moments)
    sides: np.ndarray     # n x 3 (box side lengths; sphere: radius in column 0)
    gtype: np.ndarray     # n uint8
    plane: Optional[Tuple[float, float, float, float]]
    hull_points: Optional[np.ndarray] = None    # body-frame points of the hull the GEOM_CONVEX bodies share
    hull_planes: Optional[np.ndarray] = None    # its faces (nf x 4: unit outward normal, offset), for contacts with boxes
    static_boxes: Optional[list] = None         # [(sides3, pos3, R12)]: body-less box geoms (AddBodyMap, main.c:735-761)

    @property
    def n(self):
        return self.pos.shape[0]

    def astype(self, dtype):
        f = lambda a: np.ascontiguousarray(a, dtype=dtype)
        return Scene(f(self.pos), f(self.quat), f(self.lvel), f(self.avel), f(self.mass),
                     f(self.inertia), f(self.sides), self.gtype, self.plane,
                     None if self.hull_points is None else f(self.hull_points),
                     None if self.hull_planes is None else f(self.hull_planes), self.static_boxes)

    def slice(self, lo, hi):
        return Scene(self.pos[lo:hi], self.quat[lo:hi], self.lvel[lo:hi], self.avel[lo:hi],
                     self.mass[lo:hi], self.inertia[lo:hi], self.sides[lo:hi], self.gtype[lo:hi], self.plane,
                     self.hull_points, self.hull_planes, self.static_boxes)


def box_grid(nx, nz, *, seed=1, y_range=(20.0, 50.0), spin=True, box_mass=False, plane=True,
             slabs=1, slab_gap=0.0):
    """nx x nz boxes; x = column, z = row (row-major body order).

    spin     -- omega_0 components drawn from Rand_Double(-1,1); otherwise 0
    box_mass -- dMassSetBox(density 1) instead of the reference's default m=1, I=identity
    slabs    -- split the rows into `slabs` equal groups, consecutive groups pushed apart along z by
                slab_gap metres (BASELINE config 4: disjoint islands, one slab per GPU)
    """
    n = nx * nz
    r = Rand(seed)
    d = r.next(n * DRAWS_PER_BODY).astype(np.float64).reshape(n, DRAWS_PER_BODY) / float(0xFFFFFFFF)
    sides = 0.2 + d[:, 0:3] * (1.0 - 0.2)
    y = y_range[0] + d[:, 3] * (y_range[1] - y_range[0])
    omega = (-1.0 + d[:, 4:7] * 2.0) if spin else np.zeros((n, 3))
    col = np.tile(np.arange(nx, dtype=np.float64), nz)
    row = np.repeat(np.arange(nz, dtype=np.float64), nx)
    x = (col - (nx - 1) / 2.0) * PITCH
    z = (row - (nz - 1) / 2.0) * PITCH
    if slabs > 1:
        rows_per = nz // slabs
        z = z + np.floor(row / rows_per) * slab_gap
    pos = np.stack([x, y, z], axis=1)
    quat = np.zeros((n, 4))
    quat[:, 0] = 1.0
    if box_mass:
        m = sides[:, 0] * sides[:, 1] * sides[:, 2]
        s2 = sides * sides
        inertia = (m / 12.0)[:, None] * np.stack([s2[:, 1] + s2[:, 2], s2[:, 0] + s2[:, 2], s2[:, 0] + s2[:, 1]], 1)
        mass = m[:, None]
    else:
        mass = np.ones((n, 1))
        inertia = np.ones((n, 3))
    return Scene(pos, quat, np.zeros((n, 3)), omega, mass, inertia, sides,
                 np.full(n, GEOM_BOX, np.uint8), (0.0, 1.0, 0.0, 0.0) if plane else None)


def config1(box_mass=False, spin=False):
    """1 024 free-falling boxes over a ground plane (BASELINE configs[0])."""
    return box_grid(32, 32, seed=1, spin=spin or box_mass, box_mass=box_mass, plane=True)


def config2(n_side=1024, box_mass=False):
    """1 048 576 free-falling boxes, no contacts (BASELINE configs[1])."""
    return box_grid(n_side, n_side, seed=1, spin=True, box_mass=box_mass, plane=False)


def config3(n_side=512):
    """262 144 boxes dropping onto the ground plane from y in [1,3] (BASELINE configs[2])."""
    return box_grid(n_side, n_side, seed=1, y_range=(1.0, 3.0), spin=False, plane=True)


HULL_PITCH = 3.0      # grid pitch of the hull scenes: the 0.01-scale teapot hull is 2.13 m across its bounding sphere


def hull_grid(hull, nx, nz, *, seed=1, y_range=(1.5, 3.5), spin=False, density=1.0, plane=True, tilt=0.0, floor_box=False, pitch=None):
    """nx x nz copies of one convex hull (a hull.Hull) over the ground plane, or -- floor_box -- over one static box whose
    top is at y = 0 (BASELINE configs[4]: "dropping on a static box floor, box-convex contacts").

    Every body starts upright (the hull's input-frame orientation) unless tilt > 0, which turns body i about a
    drawn horizontal axis by a drawn angle in [0, tilt]; heights and spin are drawn like box_grid's.  pitch: the grid's pitch
    (default HULL_PITCH = 3 m: tipped teapots rock and roll into their neighbours -- hull-hull contacts; BASELINE configs[4]
    names box-trimesh contacts only, every island one body as in SURVEY 8(d)'s other grids: bench.py spreads them to 4.5 m)."""
    n = nx * nz
    r = Rand(seed)
    d = r.next(n * DRAWS_PER_BODY).astype(np.float64).reshape(n, DRAWS_PER_BODY) / float(0xFFFFFFFF)
    y = y_range[0] + d[:, 3] * (y_range[1] - y_range[0])
    omega = (-1.0 + d[:, 4:7] * 2.0) if spin else np.zeros((n, 3))
    col = np.tile(np.arange(nx, dtype=np.float64), nz)
    row = np.repeat(np.arange(nz, dtype=np.float64), nx)
    pitch = HULL_PITCH if pitch is None else float(pitch)
    pos = np.stack([(col - (nx - 1) / 2.0) * pitch, y, (row - (nz - 1) / 2.0) * pitch], axis=1)
    q0 = hull.upright_quaternion()
    quat = np.tile(q0, (n, 1))
    if tilt > 0:
        ang = d[:, 0] * tilt
        phi = d[:, 1] * 2.0 * np.pi
        ax = np.stack([np.cos(phi), np.zeros(n), np.sin(phi)], axis=1)
        qt = np.concatenate([np.cos(ang / 2)[:, None], ax * np.sin(ang / 2)[:, None]], axis=1)
        w1, x1, y1, z1 = qt.T
        w2, x2, y2, z2 = q0
        quat = np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                         w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=1)
    mass = np.full((n, 1), density * hull.volume)
    inertia = np.tile(density * hull.inertia, (n, 1))
    sides = np.zeros((n, 3))
    sides[:, 0] = hull.radius                     # the broadphase reads a convex body's bounding radius here
    from . import hull as hull_mod
    statics = None
    if floor_box:
        span = max(nx, nz) * pitch + 20.0
        statics = [((span, 1.0, span), (0.0, -0.5, 0.0), _rot_z(0.0))]
        plane = False
    return Scene(pos, quat, np.zeros((n, 3)), omega, mass, inertia, sides, np.full(n, GEOM_CONVEX, np.uint8),
                 (0.0, 1.0, 0.0, 0.0) if plane else None, hull.points.copy(), hull_mod.planes(hull.points), statics)


def config4(n_side=1024, slabs=8):
    """configs[1]'s scene split into `slabs` disjoint slabs >= 10 m apart (BASELINE configs[3])."""
    return box_grid(n_side, n_side, seed=1, spin=True, plane=False, slabs=slabs, slab_gap=10.0)


# ---------------------------------------------------------------------------------------------------------
# The reference's own scene: static map boxes (StartServer, main.c:115-121) and the key-M spawner
# (main.c:502-521).  Used by the ODE-API tests, which drive these numbers through the C harness.
# ---------------------------------------------------------------------------------------------------------
def _rot_z(angle):
    """3x4 row-major rotation about z, as GetTransformMatV (main.c:624-651) yields for rot = (0,0,angle)."""
    import math
    c, s = math.cos(angle), math.sin(angle)
    return [c, -s, 0.0, 0.0, s, c, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]


def reference_map():
    """(size, pos, R[12]) of the floor and the three walls the reference creates (main.c:115-121)."""
    ident = _rot_z(0.0)
    return [
        ((100.0, 1.0, 100.0), (0.0, 0.0, 0.0), ident),
        ((0.5, 8.0, 12.0), (4.0, 3.0, 0.0), _rot_z(-0.5)),
        ((12.0, 8.0, 0.5), (0.0, 3.0, 6.0), ident),
        ((12.0, 8.0, 0.5), (0.0, 3.0, -6.0), ident),
    ]


def reference_pen(n, seed=7, y_range=(3.0, 12.0)):
    """The reference's own scene as one Scene + its static boxes: reference_map()'s floor and walls, n bodies drawn by the key-M
    spawner (reference_spawn), boxes first and spheres behind them (the order the batch's classes want).  Returns
    (scene, static boxes, number of boxes).  m = 1, I = identity as AddBody leaves them (main.c:695-733)."""
    spawn = reference_spawn(n, seed=seed, y_range=y_range)
    spawn.sort(key=lambda s: -s[0])
    k = len(spawn)
    nb = sum(1 for s in spawn if s[0] == GEOM_BOX)
    sc = Scene(np.array([s[2] for s in spawn], float), np.tile([1.0, 0, 0, 0], (k, 1)), np.zeros((k, 3)), np.zeros((k, 3)),
               np.ones((k, 1)), np.ones((k, 3)), np.array([s[1] for s in spawn], float), np.array([s[0] for s in spawn], np.uint8), None)
    return sc, reference_map(), nb


def reference_spawn(n, seed=1, y_range=(20.0, 50.0)):
    """n bodies as the key-M spawner draws them (main.c:504-519): position x,z in [-4,4], y in y_range,
    then Rand_Int(0,2) == 0 -> box with three sides in [0.2,1.0], else sphere with radius in [0.1,0.4];
    each body also consumes the three Rand_Int draws of its Rand_Color.  Returns a list of
    (type, (sx,sy,sz), (x,y,z)) with type 2 = box, 1 = sphere (BodyType, inc/body.h:14-18)."""
    r = Rand(seed)
    out = []
    for _ in range(n):
        x = r.double(-4.0, 4.0)
        y = r.double(*y_range)
        z = r.double(-4.0, 4.0)
        if r.int(0, 2) == 0:
            size = (r.double(0.2, 1.0), r.double(0.2, 1.0), r.double(0.2, 1.0))
            kind = GEOM_BOX
        else:
            size = (r.double(0.1, 0.4), 0.0, 0.0)
            kind = GEOM_SPHERE
        for _c in range(3):
            r.int(30, 190)          # Rand_Color(30, 190)
        out.append((kind, size, (x, y, z)))
    return out
