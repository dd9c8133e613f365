/*
 * dmx_hull.h -- convex-hull geometry for the batch ABI (BASELINE configs[4]: hulls of res/teapot.obj).
 *
 * The reference never creates a convex geom: res/teapot.obj is a render asset, so there is no call
 * site to replace here.  What these entries stand in for is the ODE workflow such a scene would use:
 *   dCreateConvex(space, planes, nplanes, points, npoints, polygons)   -> dmxHullBuild + dmxBatchSetConvexHull
 *   dMassSetTrimesh / hand-computed dMass for the hull                  -> dmxHullBuild's mass properties
 * Host side only (plain C++ behind a C ABI, no device work); the device side is dmxBatchSetConvexHull in dmx_batch.h.
 */
#ifndef DMX_HULL_H
#define DMX_HULL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* "v x y z" lines of a Wavefront OBJ file.  Returns the vertex count (<0: cannot read); fills at most `capacity`
 * vertices (3 doubles each) when out_xyz is not NULL, so a first call with NULL sizes the buffer. */
int64_t dmxObjReadVertices(const char *path, double *out_xyz, int64_t capacity);

typedef struct dmxHullInfo {
    int32_t n_vertices, n_faces;     /* hull vertices; triangles of the hull surface */
    double volume, area;             /* of the hull, in the input frame's units (after `scale`) */
    double com[3];                   /* centre of mass in the (scaled) input frame */
    double axes[9];                  /* row-major 3x3: rows are the principal axes in the input frame (right-handed) */
    double inertia[3];               /* principal moments about the centre of mass for unit density */
    double radius;                   /* largest distance of a hull vertex from the centre of mass */
} dmxHullInfo;

/* Convex hull (quickhull) of n points scaled by `scale`, and the uniform-density mass properties of the solid hull.
 * out_points receives the hull vertices in the BODY frame -- origin at the centre of mass, axes along the principal
 * axes, so the body's inertia tensor is diagonal (info->inertia) -- in ascending order of their index in the input;
 * out_index (may be NULL) receives those input indices.  Both hold up to `capacity` vertices; the return value is the
 * hull's vertex count (the call fills nothing beyond capacity), or <0 for degenerate input (fewer than 4 points, all
 * coplanar). */
int32_t dmxHullBuild(const double *xyz, int64_t n, double scale, double *out_points, int32_t *out_index, int32_t capacity,
                     dmxHullInfo *info);

/* The faces of the convex hull of n points as planes (4 doubles each: unit outward normal, offset; n.x <= d inside), one
 * per triangle of the hull's surface.  On a hull's body-frame points this is dCreateConvex's `planes` array; hand it to
 * dmxBatchSetConvexHullFaces.  Returns the face count (fills at most `capacity`; out_planes may be NULL to size it). */
int32_t dmxHullPlanes(const double *xyz, int64_t n, double *out_planes, int32_t capacity);

#ifdef __cplusplus
}
#endif
#endif
