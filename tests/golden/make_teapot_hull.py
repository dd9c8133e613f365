"""Generates tests/golden/teapot_hull.npz from the reference's res/teapot.obj with scipy's Qhull binding (run in the
build container, where /root/reference exists; the GPU box only sees the committed fixture).

The fixture is data derived from a data asset: the 1 265 convex-hull vertices of the OBJ's vertex list (input frame,
OBJ units, ascending by first occurrence in the file) and Qhull's facts about the hull.  SURVEY.md section 8c pins the
same facts: 1 265 vertices, 2 526 facets, volume 928 313.535894, area 49 871.335371.
"""
import os

import numpy as np
from scipy.spatial import ConvexHull

HERE = os.path.dirname(os.path.abspath(__file__))
v = np.array([[float(t) for t in line.split()[1:4]] for line in open("/root/reference/res/teapot.obj") if line.startswith("v ")])
h = ConvexHull(v)
# the OBJ repeats positions (4 884 "v" lines, 4 442 distinct): keep one index per distinct hull position, the earliest
pos = {}
for i in sorted(h.vertices.tolist()):
    pos.setdefault(tuple(v[i]), i)
idx = np.array(sorted(pos.values()), dtype=np.int32)
np.savez_compressed(os.path.join(HERE, "teapot_hull.npz"), points=v[idx], n_input=np.int64(len(v)),
                    n_vertices=np.int64(len(h.vertices)), n_facets=np.int64(len(h.simplices)),
                    volume=np.float64(h.volume), area=np.float64(h.area))
print(len(idx), "hull positions,", len(h.vertices), "Qhull vertices,", len(h.simplices), "facets, volume %.6f area %.6f" % (h.volume, h.area))
