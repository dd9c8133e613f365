"""One process of bench.py's all-cores CPU line (TEST / BASELINE INFRASTRUCTURE, see oracle/orc.h):
loads its slice of the scene from an .npz, steps it with the single-thread oracle for about `budget` seconds and
prints `<body-steps> <seconds>`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.orc_ctypes import Oracle  # noqa: E402


def main():
    path, dtype, budget = sys.argv[1], sys.argv[2], float(sys.argv[3])
    d = np.load(path)
    orc = Oracle(dtype)
    ow = orc.world()
    if d["plane"].size == 4:
        ow.add_plane(*[float(v) for v in d["plane"]])
    if "hull" in d and len(d["hull"]):
        ow.set_hull(d["hull"])
        ow.add_convex(d["pos"], d["quat"], d["lvel"], d["avel"], d["mass"], d["inertia"])
    else:
        ow.add_boxes(d["pos"], d["quat"], d["lvel"], d["avel"], d["mass"], d["inertia"], d["sides"])
    h = 1.0 / 60.0
    t = ow.run(h, 2)
    steps = int(max(2, min(2000, budget / max(t / 2, 1e-9))))
    t = ow.run(h, steps)
    print(len(d["pos"]) * steps, t, flush=True)


if __name__ == "__main__":
    main()
