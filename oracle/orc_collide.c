/*
 * oracle/orc_collide.c -- broadphase pair finding, narrowphase colliders and
 * the reference's contact policy.  TEST INFRASTRUCTURE (see orc.h).
 *
 *   dSpaceCollide(space,NULL,NearCallback)   main.c:212
 *   NearCallback                              main.c:674-693
 *   dCollide(o1,o2,8,&c[0].geom,sizeof(dContact))  main.c:678
 *
 * Colliders restate ODE's [ODE-recall]: dCollideBoxPlane (box.cpp),
 * dCollideSpherePlane / dCollideSphereSphere / dCollideSphereBox (sphere.cpp).
 * Box-box (dBoxBox) lives in orc_boxbox.c.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "orc_internal.h"

int orc_collide_box_box(const real *p1, const real *R1, const real *side1,
                        const real *p2, const real *R2, const real *side2,
                        int maxc, orc_contactgeom *out);

/* ---- dCollideBoxPlane ---------------------------------------------------- */
static int collide_box_plane(const real *pos, const real *Rm, const real *side,
                             const real *pl, int maxc, orc_contactgeom *c)
{
    const real *n = pl;
    /* project side lengths along the normal */
    real Q1 = orc_dot3_14(n, Rm + 0);
    real Q2 = orc_dot3_14(n, Rm + 1);
    real Q3 = orc_dot3_14(n, Rm + 2);
    real A[3] = { side[0] * Q1, side[1] * Q2, side[2] * Q3 };
    real B[3] = { orc_fabs(A[0]), orc_fabs(A[1]), orc_fabs(A[2]) };

    real depth = pl[3] + R(0.5) * (B[0] + B[1] + B[2]) - orc_dot3(n, pos);
    if (depth < 0) return 0;

    if (maxc < 1) maxc = 1;
    if (maxc > 4) maxc = 4;

    /* deepest corner */
    real p[3] = { pos[0], pos[1], pos[2] };
    for (int i = 0; i < 3; i++) {
        real hs = R(0.5) * side[i];
        if (A[i] > 0) {
            p[0] -= hs * Rm[0 + i]; p[1] -= hs * Rm[4 + i]; p[2] -= hs * Rm[8 + i];
        } else {
            p[0] += hs * Rm[0 + i]; p[1] += hs * Rm[4 + i]; p[2] += hs * Rm[8 + i];
        }
    }
    c[0].pos[0] = p[0]; c[0].pos[1] = p[1]; c[0].pos[2] = p[2];
    c[0].depth = depth;
    int ret = 1;

    if (maxc > 1) {
        /* second and third contacts: walk from p along the two sides with the
           smallest projected length */
        int s1, s2;
        if (B[0] < B[1]) {
            if (B[2] < B[0]) { s1 = 2; s2 = (B[0] < B[1]) ? 0 : 1; }
            else             { s1 = 0; s2 = (B[1] < B[2]) ? 1 : 2; }
        } else {
            if (B[2] < B[1]) { s1 = 2; s2 = (B[0] < B[1]) ? 0 : 1; }
            else             { s1 = 1; s2 = (B[0] < B[2]) ? 0 : 2; }
        }
        int order[2] = { s1, s2 };
        for (int k = 0; k < 2 && ret < maxc && ret < 3; k++) {
            int s = order[k];
            if (depth - B[s] < 0) break;
            real sg = (A[s] > 0) ? R(1.0) : R(-1.0);
            c[ret].pos[0] = p[0] + sg * side[s] * Rm[0 + s];
            c[ret].pos[1] = p[1] + sg * side[s] * Rm[4 + s];
            c[ret].pos[2] = p[2] + sg * side[s] * Rm[8 + s];
            c[ret].depth = depth - B[s];
            ret++;
        }
        if (maxc == 4 && ret == 3) {
            /* fourth corner of the resting face */
            real d4 = c[1].depth + c[2].depth - depth;
            if (d4 > 0) {
                c[3].pos[0] = c[1].pos[0] + c[2].pos[0] - p[0];
                c[3].pos[1] = c[1].pos[1] + c[2].pos[1] - p[1];
                c[3].pos[2] = c[1].pos[2] + c[2].pos[2] - p[2];
                c[3].depth = d4;
                ret++;
            }
        }
    }
    for (int i = 0; i < ret; i++) {
        c[i].normal[0] = n[0]; c[i].normal[1] = n[1]; c[i].normal[2] = n[2];
    }
    return ret;
}

/* ---- dCollideSpherePlane -------------------------------------------------- */
static int collide_sphere_plane(const real *pos, real radius, const real *pl, orc_contactgeom *c)
{
    real k = orc_dot3(pos, pl);
    real depth = pl[3] - k + radius;
    if (depth >= 0) {
        for (int i = 0; i < 3; i++) {
            c->normal[i] = pl[i];
            c->pos[i] = pos[i] - pl[i] * radius;
        }
        c->depth = depth;
        return 1;
    }
    return 0;
}

/* ---- dCollideSphereSphere -> dCollideSpheres ------------------------------ */
static int collide_sphere_sphere(const real *p1, real r1, const real *p2, real r2, orc_contactgeom *c)
{
    real dx = p1[0] - p2[0], dy = p1[1] - p2[1], dz = p1[2] - p2[2];
    real d = orc_sqrt(dx * dx + dy * dy + dz * dz);
    if (d > (r1 + r2)) return 0;
    if (d <= 0) {
        c->pos[0] = p1[0]; c->pos[1] = p1[1]; c->pos[2] = p1[2];
        c->normal[0] = 1; c->normal[1] = 0; c->normal[2] = 0;
        c->depth = r1 + r2;
    } else {
        real d1 = R(1.0) / d;
        c->normal[0] = dx * d1; c->normal[1] = dy * d1; c->normal[2] = dz * d1;
        real k = R(0.5) * (r2 - r1 - d);
        c->pos[0] = p1[0] + c->normal[0] * k;
        c->pos[1] = p1[1] + c->normal[1] * k;
        c->pos[2] = p1[2] + c->normal[2] * k;
        c->depth = r1 + r2 - d;
    }
    return 1;
}

/* _dSafeNormalize3 */
static void safe_normalize3(real *a)
{
    real aa0 = orc_fabs(a[0]), aa1 = orc_fabs(a[1]), aa2 = orc_fabs(a[2]), m;
    if (aa1 > aa0) m = (aa2 > aa1) ? aa2 : aa1;
    else if (aa2 > aa0) m = aa2;
    else {
        if (aa0 <= 0) { a[0] = 1; a[1] = 0; a[2] = 0; return; }
        m = aa0;
    }
    a[0] /= m; a[1] /= m; a[2] /= m;
    real l = R(1.0) / orc_sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] *= l; a[1] *= l; a[2] *= l;
}

/* ---- dCollideSphereBox ---------------------------------------------------- */
static int collide_sphere_box(const real *sp, real radius, const real *bp, const real *bR,
                              const real *side, orc_contactgeom *c)
{
    real l[3], t[3], p[3], q[3], r[3];
    int onborder = 0;
    p[0] = sp[0] - bp[0]; p[1] = sp[1] - bp[1]; p[2] = sp[2] - bp[2];
    for (int i = 0; i < 3; i++) {
        l[i] = side[i] * R(0.5);
        t[i] = orc_dot3_14(p, bR + i);
        if (t[i] < -l[i]) { t[i] = -l[i]; onborder = 1; }
        if (t[i] > l[i])  { t[i] = l[i];  onborder = 1; }
    }
    if (!onborder) {
        /* centre inside the box: push out through the closest face */
        real min_distance = l[0] - orc_fabs(t[0]);
        int mini = 0;
        for (int i = 1; i < 3; i++) {
            real fd = l[i] - orc_fabs(t[i]);
            if (fd < min_distance) { min_distance = fd; mini = i; }
        }
        c->pos[0] = sp[0]; c->pos[1] = sp[1]; c->pos[2] = sp[2];
        real tmp[3] = { 0, 0, 0 };
        tmp[mini] = (t[mini] > 0) ? R(1.0) : R(-1.0);
        orc_mul0_331(c->normal, bR, tmp);
        c->depth = min_distance + radius;
        return 1;
    }
    orc_mul0_331(q, bR, t);
    r[0] = p[0] - q[0]; r[1] = p[1] - q[1]; r[2] = p[2] - q[2];
    real depth = radius - orc_sqrt(orc_dot3(r, r));
    if (depth < 0) return 0;
    c->pos[0] = q[0] + bp[0]; c->pos[1] = q[1] + bp[1]; c->pos[2] = q[2] + bp[2];
    c->normal[0] = r[0]; c->normal[1] = r[1]; c->normal[2] = r[2];
    safe_normalize3(c->normal);
    c->depth = depth;
    return 1;
}

/* ---- convex - plane [ODE-recall dCollideConvexPlane, convex.cpp]: walk the hull's points in array order; every point
 *      on or below the plane becomes a contact (position = the point, normal = the plane's, depth = distance below)
 *      until maxc are taken; the walk stops early once maxc contacts exist AND points on both sides have been seen;
 *      the result counts only if the hull has points on both sides (or on the plane). */
/* ---- box against convex hull (BASELINE configs[4]: "box-trimesh contacts" of the teapot hulls on the static box floor).
 * ODE itself has nothing to restate here: dCollideConvexBox is an empty stub [ODE-recall].  The collider is this
 * repository's own, defined here and mirrored by the device kernel (csrc/dmx_exact.hip):
 *   1. hull vertices inside the box, in the hull's array order (as dCollideConvexPlane walks them): contact at the
 *      vertex, along the box face it is nearest to (ties: lowest axis), depth = distance to that face;
 *   2. if room is left, box corners inside the hull (corner c = signs from the bits of c): contact at the corner, along
 *      the hull face it is nearest to (ties: lowest face), depth = distance to that face.  Needs the hull's faces.
 * The first maxc contacts found are kept.  Edge-edge penetrations are not detected.  Normals point into the box (o1). */
static int collide_box_convex(const orc_world *w, const real *pb, const real *Rb, const real *side,
                              const real *ph, const real *Rh, int maxc, orc_contactgeom *c)
{
    int n = 0;
    const real half[3] = { R(0.5) * side[0], R(0.5) * side[1], R(0.5) * side[2] };
    for (int i = 0; i < w->hull_n && n < maxc; i++) {
        real v[3], d[3], q[3];
        orc_mul0_331(v, Rh, w->hull + 3 * i);
        v[0] += ph[0]; v[1] += ph[1]; v[2] += ph[2];
        for (int k = 0; k < 3; k++) d[k] = v[k] - pb[k];
        for (int a = 0; a < 3; a++) q[a] = FMA(Rb[8 + a], d[2], FMA(Rb[4 + a], d[1], Rb[a] * d[0]));   /* box frame */
        if (orc_fabs(q[0]) > half[0] || orc_fabs(q[1]) > half[1] || orc_fabs(q[2]) > half[2]) continue;
        int best = 0;
        real dep = half[0] - orc_fabs(q[0]);
        for (int a = 1; a < 3; a++) { const real e = half[a] - orc_fabs(q[a]); if (e < dep) { dep = e; best = a; } }
        const real sg = q[best] < 0 ? R(-1.0) : R(1.0);
        orc_contactgeom *t = &c[n++];
        t->pos[0] = v[0]; t->pos[1] = v[1]; t->pos[2] = v[2];
        for (int k = 0; k < 3; k++) t->normal[k] = -(sg * Rb[4 * k + best]);     /* into the box: against its outward face normal */
        t->depth = dep;
    }
    for (int cn = 0; cn < 8 && n < maxc && w->hull_nf > 0; cn++) {
        real l[3], cw[3], d[3], r[3];
        for (int a = 0; a < 3; a++) l[a] = (cn >> a) & 1 ? half[a] : -half[a];
        orc_mul0_331(cw, Rb, l);
        cw[0] += pb[0]; cw[1] += pb[1]; cw[2] += pb[2];
        for (int k = 0; k < 3; k++) d[k] = cw[k] - ph[k];
        for (int a = 0; a < 3; a++) r[a] = FMA(Rh[8 + a], d[2], FMA(Rh[4 + a], d[1], Rh[a] * d[0]));   /* hull frame */
        int inside = 1, fbest = -1;
        real dep = ORC_INF;
        for (int f = 0; f < w->hull_nf; f++) {
            const real *pl = w->hull_planes + 4 * f;
            const real e = pl[3] - orc_dot3(pl, r);
            if (e < 0) { inside = 0; break; }
            if (e < dep) { dep = e; fbest = f; }
        }
        if (!inside || fbest < 0) continue;
        real nw[3];
        orc_mul0_331(nw, Rh, w->hull_planes + 4 * fbest);
        orc_contactgeom *t = &c[n++];
        t->pos[0] = cw[0]; t->pos[1] = cw[1]; t->pos[2] = cw[2];
        t->normal[0] = nw[0]; t->normal[1] = nw[1]; t->normal[2] = nw[2];        /* the hull's outward normal points into the box */
        t->depth = dep;
    }
    return n;
}

/* ---- sphere against convex hull, and convex hull against convex hull (bodies sharing the world's one hull shape).  ODE has
 * colliders of these names; they are NOT restated here (no call site in the reference creates a convex geom at all, SURVEY F9,
 * and ODE's own convex-convex is an elaborate SAT + clipping): these two are this repository's own, defined here and mirrored by
 * the device kernels (csrc/dmx_collide_wave.hpp), built from the same two primitives as collide_box_convex above -- "a point
 * inside a hull, along the face it is nearest to" -- and sharing its blind spot (edge-edge penetrations are not detected).
 *
 *   sphere (o1) - hull (o2): c = the sphere's centre in the hull's frame; s_f = n_f . c - d_f over the hull's faces in order;
 *      s_max (first face on ties) > radius: no contact; else ONE contact along that face: normal = the face's outward normal
 *      (into the sphere), depth = radius - s_max, position = the sphere's surface point against the face.  Exact when the
 *      centre is over a face's interior; over an edge or a vertex the planes' maximum is less than the true distance, so the
 *      sphere is met up to (1 - cos) of its radius early -- fine facets (the teapot: 2 526) make that small.
 *   hull A (o1) - hull B (o2): (1) B's vertices inside A, in array order: contact at the vertex, along A's face it is nearest
 *      to (first on ties), normal = MINUS that face's outward normal (into A); (2) if room is left, A's vertices inside B:
 *      contact at the vertex, normal = B's nearest face's outward normal (into A).  The first maxc are kept. */
static void to_hull_frame(real r[3], const real *Rh, const real *ph, const real v[3])
{
    const real d[3] = { v[0] - ph[0], v[1] - ph[1], v[2] - ph[2] };
    for (int a = 0; a < 3; a++) r[a] = FMA(Rh[8 + a], d[2], FMA(Rh[4 + a], d[1], Rh[a] * d[0]));
}

/* is hull-frame point r inside the hull?  *dep = distance to the nearest face, *fbest = that face (first on ties) */
static int point_in_hull(const orc_world *w, const real r[3], real *dep, int *fbest)
{
    *dep = ORC_INF; *fbest = -1;
    for (int f = 0; f < w->hull_nf; f++) {
        const real *pl = w->hull_planes + 4 * f;
        const real e = pl[3] - orc_dot3(pl, r);
        if (e < 0) return 0;
        if (e < *dep) { *dep = e; *fbest = f; }
    }
    return *fbest >= 0;
}

static int collide_sphere_convex(const orc_world *w, const real *cs, real radius, const real *ph, const real *Rh,
                                 orc_contactgeom *c)
{
    if (w->hull_nf <= 0) return 0;
    real r[3];
    to_hull_frame(r, Rh, ph, cs);
    real smax = -ORC_INF;
    int fbest = -1;
    for (int f = 0; f < w->hull_nf; f++) {
        const real *pl = w->hull_planes + 4 * f;
        const real sdist = orc_dot3(pl, r) - pl[3];
        if (sdist > smax) { smax = sdist; fbest = f; }
    }
    if (fbest < 0 || smax > radius) return 0;
    real nw[3];
    orc_mul0_331(nw, Rh, w->hull_planes + 4 * fbest);
    c->normal[0] = nw[0]; c->normal[1] = nw[1]; c->normal[2] = nw[2];
    c->depth = radius - smax;
    c->pos[0] = cs[0] - nw[0] * radius; c->pos[1] = cs[1] - nw[1] * radius; c->pos[2] = cs[2] - nw[2] * radius;
    return 1;
}

static int collide_convex_convex(const orc_world *w, const real *pa, const real *Ra, const real *pb, const real *Rb,
                                 int maxc, orc_contactgeom *c)
{
    int n = 0;
    if (w->hull_nf <= 0) return 0;
    for (int pass = 0; pass < 2; pass++) {
        /* pass 0: B's vertices against A; pass 1: A's vertices against B */
        const real *pv = pass == 0 ? pb : pa, *Rv = pass == 0 ? Rb : Ra;      /* the hull whose vertices are walked */
        const real *ph = pass == 0 ? pa : pb, *Rh = pass == 0 ? Ra : Rb;      /* the hull they are tested against */
        for (int i = 0; i < w->hull_n && n < maxc; i++) {
            real v[3], r[3], dep;
            int fbest;
            orc_mul0_331(v, Rv, w->hull + 3 * i);
            v[0] += pv[0]; v[1] += pv[1]; v[2] += pv[2];
            to_hull_frame(r, Rh, ph, v);
            if (!point_in_hull(w, r, &dep, &fbest)) continue;
            real nw[3];
            orc_mul0_331(nw, Rh, w->hull_planes + 4 * fbest);
            orc_contactgeom *t = &c[n++];
            t->pos[0] = v[0]; t->pos[1] = v[1]; t->pos[2] = v[2];
            for (int k = 0; k < 3; k++) t->normal[k] = pass == 0 ? -nw[k] : nw[k];      /* into A either way */
            t->depth = dep;
        }
    }
    return n;
}

static int collide_convex_plane(const orc_world *w, const real *pos, const real *Rm, const real *pl, int maxc,
                                orc_contactgeom *c)
{
    enum { LTEQ_ZERO = 1, GTEQ_ZERO = 2, BOTH_SIGNS = 3 };
    int contacts = 0, totalsign = 0;
    for (int i = 0; i < w->hull_n; i++) {
        real v2[3];
        orc_mul0_331(v2, Rm, w->hull + 3 * i);
        v2[0] += pos[0]; v2[1] += pos[1]; v2[2] += pos[2];
        int sign = GTEQ_ZERO;
        real distance2 = orc_dot3(pl, v2) - pl[3];
        if (distance2 <= 0) {
            sign = distance2 != 0 ? LTEQ_ZERO : BOTH_SIGNS;
            if (contacts != maxc) {
                orc_contactgeom *t = &c[contacts];
                t->normal[0] = pl[0]; t->normal[1] = pl[1]; t->normal[2] = pl[2];
                t->pos[0] = v2[0]; t->pos[1] = v2[1]; t->pos[2] = v2[2];
                t->depth = -distance2;
                contacts++;
            }
        }
        totalsign |= sign;
        if (contacts == maxc && totalsign == BOTH_SIGNS) break;
    }
    return totalsign == BOTH_SIGNS ? contacts : 0;
}

/* ---- dCollide dispatch (main.c:678) --------------------------------------- */
static int collide_ordered(orc_world *w, const orc_geom *a, const orc_geom *b, int maxc,
                           orc_contactgeom *out, int *handled)
{
    *handled = 1;
    const real *pa = orc_geom_pos(w, a), *Ra = orc_geom_R(w, a);
    const real *pb = orc_geom_pos(w, b), *Rb = orc_geom_R(w, b);
    if (a->type == ORC_GEOM_BOX && b->type == ORC_GEOM_PLANE)
        return collide_box_plane(pa, Ra, a->side, b->plane, maxc, out);
    if (a->type == ORC_GEOM_SPHERE && b->type == ORC_GEOM_PLANE)
        return collide_sphere_plane(pa, a->side[0], b->plane, out);
    if (a->type == ORC_GEOM_SPHERE && b->type == ORC_GEOM_SPHERE)
        return collide_sphere_sphere(pa, a->side[0], pb, b->side[0], out);
    if (a->type == ORC_GEOM_SPHERE && b->type == ORC_GEOM_BOX)
        return collide_sphere_box(pa, a->side[0], pb, Rb, b->side, out);
    if (a->type == ORC_GEOM_BOX && b->type == ORC_GEOM_BOX)
        return orc_collide_box_box(pa, Ra, a->side, pb, Rb, b->side, maxc, out);
    if (a->type == ORC_GEOM_CONVEX && b->type == ORC_GEOM_PLANE)
        return collide_convex_plane(w, pa, Ra, b->plane, maxc, out);
    if (a->type == ORC_GEOM_BOX && b->type == ORC_GEOM_CONVEX)
        return collide_box_convex(w, pa, Ra, a->side, pb, Rb, maxc, out);
    if (a->type == ORC_GEOM_SPHERE && b->type == ORC_GEOM_CONVEX)
        return collide_sphere_convex(w, pa, a->side[0], pb, Rb, out);
    if (a->type == ORC_GEOM_CONVEX && b->type == ORC_GEOM_CONVEX)
        return collide_convex_convex(w, pa, Ra, pb, Rb, maxc, out);
    /* (convex, box) and (convex, sphere) have no collider in this order: dCollide swaps and flips the normals */
    *handled = 0;
    return 0;
}

int orc_collide(orc_world *w, int g1, int g2, int maxc, orc_contactgeom *out)
{
    /* [ODE-recall] dCollide: no self / same-body contacts; if only the
       reversed class pair has a collider, call it swapped and flip normals */
    if (g1 == g2) return 0;
    const orc_geom *a = &w->geoms[g1], *b = &w->geoms[g2];
    if (a->body >= 0 && a->body == b->body) return 0;
    int handled, n;
    n = collide_ordered(w, a, b, maxc, out, &handled);
    if (handled) {
        for (int i = 0; i < n; i++) { out[i].g1 = g1; out[i].g2 = g2; }
        return n;
    }
    n = collide_ordered(w, b, a, maxc, out, &handled);
    if (!handled) return 0;   /* plane-plane: no collider */
    for (int i = 0; i < n; i++) {
        out[i].normal[0] = -out[i].normal[0];
        out[i].normal[1] = -out[i].normal[1];
        out[i].normal[2] = -out[i].normal[2];
        out[i].g1 = g1; out[i].g2 = g2;
    }
    return n;
}

/* test helper (tests/test_collider_geometry.py): the two BODY-LESS geoms g1, g2 are given n poses (and, when sizes are passed,
 * n sets of side lengths / a radius in [0]) in turn and collided each time -- the narrowphase alone, thousands of random pairs
 * per second without a Python call per pair.  pose = position (3) + rotation (3x4 row-major, 12).  out holds maxc slots per pair. */
void orc_collide_bulk(orc_world *w, int g1, int g2, int n, const real *pose1, const real *size1, const real *pose2, const real *size2,
                      int maxc, int *counts, orc_contactgeom *out)
{
    orc_geom *a = &w->geoms[g1], *b = &w->geoms[g2];
    for (int k = 0; k < n; k++) {
        const real *pa = pose1 + 15 * (size_t)k, *pb = pose2 + 15 * (size_t)k;
        for (int i = 0; i < 3; i++) { a->pos[i] = pa[i]; b->pos[i] = pb[i]; }
        for (int i = 0; i < 12; i++) { a->R[i] = pa[3 + i]; b->R[i] = pb[3 + i]; }
        if (size1) for (int i = 0; i < 3; i++) a->side[i] = size1[3 * (size_t)k + i];
        if (size2) for (int i = 0; i < 3; i++) b->side[i] = size2[3 * (size_t)k + i];
        counts[k] = orc_collide(w, g1, g2, maxc, out + (size_t)maxc * k);
    }
}

/* ---- NearCallback (main.c:674-693) ---------------------------------------- */
static void joint_push(orc_world *w, const orc_contactgeom *cg, int b1, int b2)
{
    if (w->nj == w->cap_j) {
        w->cap_j = w->cap_j ? 2 * w->cap_j : 256;
        w->joints = (orc_joint *)realloc(w->joints, (size_t)w->cap_j * sizeof(orc_joint));
    }
    orc_joint *j = &w->joints[w->nj++];
    j->geom = *cg;
    j->mode = w->surf_mode;               /* main.c:684 */
    j->bounce = w->surf_bounce;           /* main.c:685 */
    j->bounce_vel = w->surf_bounce_vel;   /* main.c:686 */
    j->mu = w->surf_mu;                   /* main.c:687 */
    /* [ODE-recall] dJointAttach: if body1 is null and body2 is not, swap and
       set dJOINT_REVERSE */
    j->reverse = 0;
    if (b1 < 0 && b2 >= 0) { b1 = b2; b2 = -1; j->reverse = 1; }
    j->b1 = b1; j->b2 = b2;
    j->tag = 0;
}

static void near_callback(orc_world *w, int o1, int o2)
{
    orc_contactgeom cg[16];
    int maxc = w->max_contacts > 16 ? 16 : w->max_contacts;
    int nc = orc_collide(w, o1, o2, maxc, cg);          /* main.c:678 */
    if (nc <= 0) return;
    int b1 = w->geoms[o1].body, b2 = w->geoms[o2].body; /* main.c:691 dGeomGetBody */
    if (b1 < 0 && b2 < 0) return;   /* static-static joints are ignored by the stepper (SURVEY a-5) */
    for (int i = 0; i < nc; i++) joint_push(w, &cg[i], b1, b2);
}

/* ---- broadphase ------------------------------------------------------------ */
typedef struct { real lo[3], hi[3]; int g; } aabb_t;

static int cmp_aabb(const void *a, const void *b)
{
    real x = ((const aabb_t *)a)->lo[0], y = ((const aabb_t *)b)->lo[0];
    if (x < y) return -1;
    if (x > y) return 1;
    int ga = ((const aabb_t *)a)->g, gb = ((const aabb_t *)b)->g;
    return (ga > gb) - (ga < gb);
}

static int cmp_pair(const void *a, const void *b)
{
    const int *p = (const int *)a, *q = (const int *)b;
    if (p[0] != q[0]) return (p[0] > q[0]) - (p[0] < q[0]);
    return (p[1] > q[1]) - (p[1] < q[1]);
}

static int pair_passes(const orc_geom *a, const orc_geom *b)
{
    /* [ODE-recall] collideAABBs: same non-null body -> skip; category/collide test */
    if (a->body >= 0 && a->body == b->body) return 0;
    if (!((a->cat & b->col) || (b->cat & a->col))) return 0;
    return 1;
}

void orc_collide_all(orc_world *w)
{
    w->nj = 0;
    w->last_body_pairs = 0;
    int ng = w->ng;
    aabb_t *bb = (aabb_t *)malloc((size_t)(ng ? ng : 1) * sizeof(aabb_t));
    int nbb = 0;
    int *pairs = NULL; size_t np = 0, cap = 0;
#define PUSH_PAIR(A, B) do { if (np == cap) { cap = cap ? 2 * cap : 1024; \
        pairs = (int *)realloc(pairs, 2 * cap * sizeof(int)); } \
        pairs[2 * np] = (A) < (B) ? (A) : (B); pairs[2 * np + 1] = (A) < (B) ? (B) : (A); np++; } while (0)

    /* finite AABBs: box = centre +- sum_j |R_ij| side_j / 2; sphere = centre +- r; convex = bounds of its
       transformed points [ODE-recall dxConvex::computeAABB] */
    for (int g = 0; g < ng; g++) {
        const orc_geom *ge = &w->geoms[g];
        if (ge->type == ORC_GEOM_PLANE) continue;
        const real *p = orc_geom_pos(w, ge), *Rm = orc_geom_R(w, ge);
        aabb_t *a = &bb[nbb++];
        a->g = g;
        if (ge->type == ORC_GEOM_CONVEX) {
            for (int i = 0; i < 3; i++) { a->lo[i] = ORC_INF; a->hi[i] = -ORC_INF; }
            for (int k = 0; k < w->hull_n; k++) {
                real v[3];
                orc_mul0_331(v, Rm, w->hull + 3 * k);
                for (int i = 0; i < 3; i++) {
                    real c = v[i] + p[i];
                    if (c < a->lo[i]) a->lo[i] = c;
                    if (c > a->hi[i]) a->hi[i] = c;
                }
            }
            continue;
        }
        for (int i = 0; i < 3; i++) {
            real r = (ge->type == ORC_GEOM_SPHERE)
                         ? ge->side[0]
                         : R(0.5) * (orc_fabs(Rm[4 * i] * ge->side[0]) +
                                     orc_fabs(Rm[4 * i + 1] * ge->side[1]) +
                                     orc_fabs(Rm[4 * i + 2] * ge->side[2]));
            a->lo[i] = p[i] - r; a->hi[i] = p[i] + r;
        }
    }
    /* planes are unbounded: test against every non-plane geom */
    for (int g = 0; g < ng; g++) {
        if (w->geoms[g].type != ORC_GEOM_PLANE) continue;
        for (int k = 0; k < nbb; k++)
            if (pair_passes(&w->geoms[g], &w->geoms[bb[k].g])) PUSH_PAIR(g, bb[k].g);
    }
    /* finite AABBs: uniform (x,z) cell grid when the scene is large and evenly sized (what ODE's hash space
       does with its cell levels), else a sweep along x.  Both enumerate exactly the overlapping AABB pairs. */
    int used_grid = 0;
    if (w->bp_mode == 2 || (w->bp_mode == 0 && nbb >= 2048)) {
        real cell = 0, minx = bb[0].lo[0], maxx = bb[0].hi[0], minz = bb[0].lo[2], maxz = bb[0].hi[2];
        for (int i = 0; i < nbb; i++) {
            real ex = bb[i].hi[0] - bb[i].lo[0], ez = bb[i].hi[2] - bb[i].lo[2];
            if (ex > cell) cell = ex;
            if (ez > cell) cell = ez;
            if (bb[i].lo[0] < minx) minx = bb[i].lo[0];
            if (bb[i].hi[0] > maxx) maxx = bb[i].hi[0];
            if (bb[i].lo[2] < minz) minz = bb[i].lo[2];
            if (bb[i].hi[2] > maxz) maxz = bb[i].hi[2];
        }
        double fx = cell > 0 ? ((double)maxx - minx) / cell + 1 : 0, fz = cell > 0 ? ((double)maxz - minz) / cell + 1 : 0;
        if (cell > 0 && fx * fz <= 16.0e6 && (w->bp_mode == 2 || fx * fz >= nbb / 16.0)) {
            int nx = (int)fx + 1, nz = (int)fz + 1;
            size_t ncell = (size_t)nx * nz;
            int *start = (int *)calloc(ncell + 1, sizeof(int));
            int *cellof = (int *)malloc((size_t)nbb * sizeof(int));
            int *order = (int *)malloc((size_t)nbb * sizeof(int));
            real inv = R(1.0) / cell;
            for (int i = 0; i < nbb; i++) {
                /* bin by the AABB centre; overlapping AABBs are then at most one cell apart */
                int ix = (int)((R(0.5) * (bb[i].lo[0] + bb[i].hi[0]) - minx) * inv);
                int iz = (int)((R(0.5) * (bb[i].lo[2] + bb[i].hi[2]) - minz) * inv);
                cellof[i] = iz * nx + ix;
                start[cellof[i] + 1]++;
            }
            for (size_t c = 0; c < ncell; c++) start[c + 1] += start[c];
            int *fill = (int *)malloc(ncell * sizeof(int));
            memcpy(fill, start, ncell * sizeof(int));
            for (int i = 0; i < nbb; i++) order[fill[cellof[i]]++] = i;
            for (int i = 0; i < nbb; i++) {
                int ix = cellof[i] % nx, iz = cellof[i] / nx;
                for (int dz = -1; dz <= 1; dz++) {
                    if (iz + dz < 0 || iz + dz >= nz) continue;
                    for (int dx = -1; dx <= 1; dx++) {
                        if (ix + dx < 0 || ix + dx >= nx) continue;
                        int c = (iz + dz) * nx + ix + dx;
                        for (int t = start[c]; t < start[c + 1]; t++) {
                            int j = order[t];
                            if (j <= i) continue;
                            if (bb[j].lo[0] > bb[i].hi[0] || bb[i].lo[0] > bb[j].hi[0]) continue;
                            if (bb[j].lo[1] > bb[i].hi[1] || bb[i].lo[1] > bb[j].hi[1]) continue;
                            if (bb[j].lo[2] > bb[i].hi[2] || bb[i].lo[2] > bb[j].hi[2]) continue;
                            if (pair_passes(&w->geoms[bb[i].g], &w->geoms[bb[j].g])) { PUSH_PAIR(bb[i].g, bb[j].g); w->last_body_pairs++; }
                        }
                    }
                }
            }
            free(fill); free(order); free(cellof); free(start);
            used_grid = 1;
        }
    }
    if (!used_grid) {
        qsort(bb, (size_t)nbb, sizeof(aabb_t), cmp_aabb);
        for (int i = 0; i < nbb; i++) {
            for (int j = i + 1; j < nbb && bb[j].lo[0] <= bb[i].hi[0]; j++) {
                if (bb[j].lo[1] > bb[i].hi[1] || bb[i].lo[1] > bb[j].hi[1]) continue;
                if (bb[j].lo[2] > bb[i].hi[2] || bb[i].lo[2] > bb[j].hi[2]) continue;
                if (pair_passes(&w->geoms[bb[i].g], &w->geoms[bb[j].g])) { PUSH_PAIR(bb[i].g, bb[j].g); w->last_body_pairs++; }
            }
        }
    }
    /* canonical order: ascending (g1,g2), g1 < g2 */
    qsort(pairs, np, 2 * sizeof(int), cmp_pair);
    for (size_t k = 0; k < np; k++) near_callback(w, pairs[2 * k], pairs[2 * k + 1]);
    free(pairs);
    free(bb);
#undef PUSH_PAIR
}
