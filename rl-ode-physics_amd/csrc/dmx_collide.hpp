// dmx_collide.hpp -- narrowphase colliders (dCollide, /root/reference/src/main.c:678) for the geometry
// classes the reference creates (main.c:717 sphere, 720/743 box) plus the ground half-space of the
// BASELINE scenes.  __host__ __device__ templates: the device kernels call them per body / per pair,
// and the host side of the ODE-compatible API calls the same code from inside the user's near callback.
// Contact normals point into the first geometry ("body 1"), as the contact-joint rows expect.
#pragma once

#include "dmx_math.hpp"

namespace dmx {

template <class T> struct ContactPoint {
    V3<T> pos;
    V3<T> normal;
    T depth;
};

template <class T> DMX_HD V3<T> colv(const M3<T> &R, int j) { return { R.m[0][j], R.m[1][j], R.m[2][j] }; }
template <class T> DMX_HD T pick(const T v[3], int s) { return s == 0 ? v[0] : (s == 1 ? v[1] : v[2]); }

// ---- box vs half-space n.x = d: <= 4 contacts, deepest corner first, then along the two sides with the
// smallest projection on n, then the fourth corner of a resting face --------------------------------------
template <class T>
DMX_HD int box_plane(const V3<T> &pos, const M3<T> &R, const T side[3], const V3<T> &n, T d, int maxc,
                     V3<T> cp[4], T cd[4])
{
    const T Q1 = dot(n, colv(R, 0));
    const T Q2 = dot(n, colv(R, 1));
    const T Q3 = dot(n, colv(R, 2));
    const T A[3] = { side[0] * Q1, side[1] * Q2, side[2] * Q3 };
    const T B[3] = { tabs(A[0]), tabs(A[1]), tabs(A[2]) };
    const T depth = d + T(0.5) * (B[0] + B[1] + B[2]) - dot(n, pos);
    if (depth < 0) return 0;
    if (maxc < 1) maxc = 1;
    if (maxc > 4) maxc = 4;
    V3<T> p = pos;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const T hs = T(0.5) * side[i];
        if (A[i] > 0) { p.x -= hs * R.m[0][i]; p.y -= hs * R.m[1][i]; p.z -= hs * R.m[2][i]; }
        else          { p.x += hs * R.m[0][i]; p.y += hs * R.m[1][i]; p.z += hs * R.m[2][i]; }
    }
    cp[0] = p; cd[0] = depth;
    int ret = 1;
    if (maxc > 1) {
        int s1, s2;
        if (B[0] < B[1]) {
            if (B[2] < B[0]) { s1 = 2; s2 = 0; }
            else             { s1 = 0; s2 = (B[1] < B[2]) ? 1 : 2; }
        } else {
            if (B[2] < B[1]) { s1 = 2; s2 = 1; }
            else             { s1 = 1; s2 = (B[0] < B[2]) ? 0 : 2; }
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int s = k == 0 ? s1 : s2;
            // select column s without dynamic register indexing
            const T Bs = pick(B, s), As = pick(A, s), ss = pick(side, s);
            const T r0 = s == 0 ? R.m[0][0] : (s == 1 ? R.m[0][1] : R.m[0][2]);
            const T r1 = s == 0 ? R.m[1][0] : (s == 1 ? R.m[1][1] : R.m[1][2]);
            const T r2 = s == 0 ? R.m[2][0] : (s == 1 ? R.m[2][1] : R.m[2][2]);
            if (ret == k + 1 && ret < maxc) {
                if (!(depth - Bs < 0)) {
                    const T sg = (As > 0) ? T(1) : T(-1);
                    cp[k + 1] = { p.x + sg * ss * r0, p.y + sg * ss * r1, p.z + sg * ss * r2 };
                    cd[k + 1] = depth - Bs;
                    ret = k + 2;
                }
            }
        }
        if (maxc == 4 && ret == 3) {
            const T d4 = cd[1] + cd[2] - depth;
            if (d4 > 0) {
                cp[3] = { cp[1].x + cp[2].x - p.x, cp[1].y + cp[2].y - p.y, cp[1].z + cp[2].z - p.z };
                cd[3] = d4;
                ret = 4;
            }
        }
    }
    return ret;
}

template <class T>
DMX_HD int sphere_plane(const V3<T> &pos, T radius, const V3<T> &n, T d, V3<T> cp[4], T cd[4])
{
    const T k = dot(pos, n);
    const T depth = d - k + radius;
    if (depth >= 0) {
        cp[0] = { pos.x - n.x * radius, pos.y - n.y * radius, pos.z - n.z * radius };
        cd[0] = depth;
        return 1;
    }
    return 0;
}

// ---- sphere vs sphere ------------------------------------------------------------------------------------
template <class T>
DMX_HD int sphere_sphere(const V3<T> &p1, T r1, const V3<T> &p2, T r2, ContactPoint<T> *c)
{
    const T dx = p1.x - p2.x, dy = p1.y - p2.y, dz = p1.z - p2.z;
    const T d = tsqrt<T>(dx * dx + dy * dy + dz * dz);
    if (d > (r1 + r2)) return 0;
    if (d <= 0) {
        c->pos = p1;
        c->normal = { T(1), T(0), T(0) };
        c->depth = r1 + r2;
    } else {
        const T d1 = T(1) / d;
        c->normal = { dx * d1, dy * d1, dz * d1 };
        const T k = T(0.5) * (r2 - r1 - d);
        c->pos = { p1.x + c->normal.x * k, p1.y + c->normal.y * k, p1.z + c->normal.z * k };
        c->depth = r1 + r2 - d;
    }
    return 1;
}

template <class T> DMX_HD void safe_normalize3(V3<T> &a)
{
    const T aa0 = tabs(a.x), aa1 = tabs(a.y), aa2 = tabs(a.z);
    T m;
    if (aa1 > aa0) m = (aa2 > aa1) ? aa2 : aa1;
    else if (aa2 > aa0) m = aa2;
    else {
        if (aa0 <= 0) { a = { T(1), T(0), T(0) }; return; }
        m = aa0;
    }
    a.x /= m; a.y /= m; a.z /= m;
    const T l = T(1) / tsqrt<T>(a.x * a.x + a.y * a.y + a.z * a.z);
    a.x *= l; a.y *= l; a.z *= l;
}

// ---- sphere (geom 1) vs box (geom 2) -----------------------------------------------------------------------
template <class T>
DMX_HD int sphere_box(const V3<T> &sp, T radius, const V3<T> &bp, const M3<T> &bR, const T side[3],
                      ContactPoint<T> *c)
{
    T l[3], t[3];
    bool onborder = false;
    const V3<T> p = { sp.x - bp.x, sp.y - bp.y, sp.z - bp.z };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        l[i] = side[i] * T(0.5);
        t[i] = dot(p, colv(bR, i));
        if (t[i] < -l[i]) { t[i] = -l[i]; onborder = true; }
        if (t[i] > l[i])  { t[i] = l[i];  onborder = true; }
    }
    if (!onborder) {
        // centre inside the box: push out through the closest face
        T min_distance = l[0] - tabs(t[0]);
        int mini = 0;
#pragma unroll
        for (int i = 1; i < 3; i++) {
            const T fd = l[i] - tabs(t[i]);
            if (fd < min_distance) { min_distance = fd; mini = i; }
        }
        c->pos = sp;
        V3<T> tmp = { T(0), T(0), T(0) };
        const T sg = (pick(t, mini) > 0) ? T(1) : T(-1);
        if (mini == 0) tmp.x = sg; else if (mini == 1) tmp.y = sg; else tmp.z = sg;
        c->normal = mulv(bR, tmp);
        c->depth = min_distance + radius;
        return 1;
    }
    const V3<T> tv = { t[0], t[1], t[2] };
    const V3<T> q = mulv(bR, tv);
    V3<T> r = { p.x - q.x, p.y - q.y, p.z - q.z };
    const T depth = radius - tsqrt<T>(dot(r, r));
    if (depth < 0) return 0;
    c->pos = { q.x + bp.x, q.y + bp.y, q.z + bp.z };
    safe_normalize3(r);
    c->normal = r;
    c->depth = depth;
    return 1;
}

// ---- box vs box: 15-axis separating-axis test, then edge-edge closest points or reference-face /
// incident-face clipping (<= 8 contacts) --------------------------------------------------------------------
namespace detail {

template <class T>
DMX_HD void line_closest_approach(const V3<T> &pa, const V3<T> &ua, const V3<T> &pb, const V3<T> &ub, T &alpha, T &beta)
{
    const V3<T> p = { pb.x - pa.x, pb.y - pa.y, pb.z - pa.z };
    const T uaub = dot(ua, ub);
    const T q1 = dot(ua, p);
    const T q2 = -dot(ub, p);
    T d = 1 - uaub * uaub;
    if (d <= T(0.0001)) { alpha = 0; beta = 0; }
    else {
        d = T(1) / d;
        alpha = (q1 + uaub * q2) * d;
        beta = (uaub * q1 + q2) * d;
    }
}

// ---- the incident face clipped against the reference face's rectangle, in REGISTERS ------------------------------------
// ODE's intersectRectQuad is a Sutherland-Hodgman pass per rectangle side over point arrays indexed by running counters; on a
// GPU such arrays live in scratch memory and every step of the walk is a round trip through the memory pipeline (one lane per
// box pair: a chain of them).  Here a polygon is eight points held in named registers -- every index below is a compile-time
// constant after unrolling -- and "append at position n" is a chain of selects.  Same arithmetic, same order of points, same
// early stop at eight points as the sequential walk (the CPU restatement the tests compare with), operation for operation.
template <class T> struct Poly8 { T x[8], y[8]; };

// point (px, py) to slot `at` (0 <= at < 8); slots above `hi` cannot be meant (at <= hi by construction; hi is a constant
// once the caller's loop is unrolled, so the surplus selects fold away)
template <class T> DMX_HD void poly_put(Poly8<T> &p, int at, int hi, T px, T py)
{
#pragma unroll
    for (int s = 0; s < 8; s++) {
        if (s <= hi) {
            const bool here = s == at;
            p.x[s] = here ? px : p.x[s];
            p.y[s] = here ? py : p.y[s];
        }
    }
}

// one side of the rectangle: keep what lies on the inner side of  sign * coord[DIR] < h, cut the crossing edges.
// NQMAX = the most points the input can hold at this stage (4 for the quad itself); returns the output count, `full` once
// eight points are out (the walk stops there, as ODE's does).
template <class T, int DIR, int NQMAX>
DMX_HD int clip_side(const Poly8<T> &q, int nq, int sign, T h, Poly8<T> &r, bool &full)
{
    int nr = 0;
    const T sh = (T)sign * h;
#pragma unroll
    for (int i = 0; i < NQMAX; i++) {
        if (i < nq && !full) {
            const T px = q.x[i], py = q.y[i];
            // the next point, cyclically: q[i + 1], or q[0] behind the last one
            const bool wrap = !(i + 1 < nq);
            const T nx = (i + 1 < NQMAX) ? (wrap ? q.x[0] : q.x[i + 1 < 8 ? i + 1 : 7]) : q.x[0];
            const T ny = (i + 1 < NQMAX) ? (wrap ? q.y[0] : q.y[i + 1 < 8 ? i + 1 : 7]) : q.y[0];
            const T pd = DIR == 0 ? px : py, po = DIR == 0 ? py : px;      // along the clipped axis / the other one
            const T nd = DIR == 0 ? nx : ny, no = DIR == 0 ? ny : nx;
            const bool in0 = (T)sign * pd < h;
            const bool in1 = (T)sign * nd < h;
            if (in0) {
                poly_put<T>(r, nr, 2 * i, px, py);
                nr++;
                if (nr & 8) full = true;
            }
            if (in0 != in1 && !full) {
                const T cut = po + (no - po) / (nd - pd) * (sh - pd);
                poly_put<T>(r, nr, 2 * i + 1, DIR == 0 ? sh : cut, DIR == 0 ? cut : sh);
                nr++;
                if (nr & 8) full = true;
            }
        }
    }
    return nr;
}

// clip the quad p[8] (4 xy pairs) against |x| <= h[0], |y| <= h[1]; result in `out`, returns the point count
template <class T> DMX_HD int intersect_rect_quad(const T h[2], const T p[8], Poly8<T> &out)
{
    Poly8<T> a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a.x[i] = a.y[i] = b.x[i] = b.y[i] = T(0); }
#pragma unroll
    for (int i = 0; i < 4; i++) { a.x[i] = p[2 * i]; a.y[i] = p[2 * i + 1]; }
    bool full = false;
    int n = clip_side<T, 0, 4>(a, 4, -1, h[0], b, full);                 // (a side clipped after `full` is skipped whole,
    if (!full) { n = clip_side<T, 0, 8>(b, n, +1, h[0], a, full);        //  as the sequential walk's loops end there)
        if (!full) { n = clip_side<T, 1, 8>(a, n, -1, h[1], b, full);
            if (!full) { n = clip_side<T, 1, 8>(b, n, +1, h[1], a, full); out = a; }
            else out = b; }
        else out = a; }
    else out = b;
    return n;
}

}  // namespace detail

// cullPoints [ODE-recall, box.cpp]: of the n clipped 2-D points keep m, point i0 (the deepest) first, the others
// chosen nearest to m evenly spaced directions about the polygon's centroid.
DMX_HD float  tatan2(float y, float x)   { return ::atan2f(y, x); }
DMX_HD double tatan2(double y, double x) { return ::atan2(y, x); }
template <class T> DMX_HD void cull_points(int n, const T p[], int m, int i0, int iret[])
{
    T a, cx, cy, q;
    if (n == 1) { cx = p[0]; cy = p[1]; }
    else if (n == 2) { cx = T(0.5) * (p[0] + p[2]); cy = T(0.5) * (p[1] + p[3]); }
    else {
        a = 0; cx = 0; cy = 0;
        for (int i = 0; i < n - 1; i++) {
            q = p[i * 2] * p[i * 2 + 3] - p[i * 2 + 2] * p[i * 2 + 1];
            a += q;
            cx += q * (p[i * 2] + p[i * 2 + 2]);
            cy += q * (p[i * 2 + 1] + p[i * 2 + 3]);
        }
        q = p[n * 2 - 2] * p[1] - p[0] * p[n * 2 - 1];
        a = T(1.0) / (T(3.0) * (a + q));
        cx = a * (cx + q * (p[n * 2 - 2] + p[0]));
        cy = a * (cy + q * (p[n * 2 - 1] + p[1]));
    }
    T A[8];
    for (int i = 0; i < n; i++) A[i] = tatan2(p[i * 2 + 1] - cy, p[i * 2] - cx);
    int avail[8];
    for (int i = 0; i < n; i++) avail[i] = 1;
    avail[i0] = 0;
    iret[0] = i0;
    int w = 1;
    const T pi = T(3.14159265358979323846);
    for (int j = 1; j < m; j++) {
        a = (T)((T)j * (2 * pi / m) + A[i0]);
        if (a > pi) a -= 2 * pi;
        T maxdiff = T(1e9), diff;
        iret[w] = i0;
        for (int i = 0; i < n; i++) {
            if (avail[i]) {
                diff = tabs(A[i] - a);
                if (diff > pi) diff = 2 * pi - diff;
                if (diff < maxdiff) { maxdiff = diff; iret[w] = i; }
            }
        }
        avail[iret[w]] = 0;
        w++;
    }
}

// static-index selects (a runtime index into a register array would send the array to scratch memory on the device)
template <class T> DMX_HD V3<T> colv_sel(const M3<T> &R, int j)
{
    return { j == 0 ? R.m[0][0] : (j == 1 ? R.m[0][1] : R.m[0][2]), j == 0 ? R.m[1][0] : (j == 1 ? R.m[1][1] : R.m[1][2]),
             j == 0 ? R.m[2][0] : (j == 1 ? R.m[2][1] : R.m[2][2]) };
}
template <class T> DMX_HD M3<T> m3_sel(bool first, const M3<T> &A, const M3<T> &B)
{
    M3<T> r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = first ? A.m[i][j] : B.m[i][j];
    return r;
}

// Returns the contact count; contacts' normal = -(box1 -> box2 separating axis).  maxc >= 8 returns every
// clipped point (the reference asks for 8, main.c:675); with a smaller maxc the surplus points are culled as ODE does
// (cullPoints: the deepest one, then the ones nearest to evenly spaced directions about the centroid).
// Written for one lane per box pair with everything in registers: all array indices are compile-time constants after
// unrolling (selects pick the reference box, its face axes and the clipped points), the face clipping is
// detail::intersect_rect_quad above.  Only the culling branch (maxc below the clipped count: never with the reference's
// MAX_CONTACTS = 8) keeps indexed arrays.  The arithmetic is the sequential dBoxBox's [ODE-recall box.cpp], operation for
// operation, so host, device and the CPU restatement agree bit for bit (tests/test_collider_equivalence.py).
template <class T>
DMX_HD int box_box(const V3<T> &p1, const M3<T> &R1, const T side1[3], const V3<T> &p2, const M3<T> &R2,
                   const T side2[3], int maxc_in, ContactPoint<T> *out)
{
    const T fudge_factor = T(1.05);
    const V3<T> p = { p2.x - p1.x, p2.y - p1.y, p2.z - p1.z };
    const T pp[3] = { dot(colv(R1, 0), p), dot(colv(R1, 1), p), dot(colv(R1, 2), p) };
    T A[3], B[3], Rr[3][3], Q[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) { A[i] = side1[i] * T(0.5); B[i] = side2[i] * T(0.5); }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { Rr[i][j] = dot(colv(R1, i), colv(R2, j)); Q[i][j] = tabs(Rr[i][j]); }

    T s = -Limits<T>::inf(), s2, l, e;
    bool invert_normal = false;
    int code = 0;
    int normalR_box = 0, normalR_col = 0;     // code <= 6: normal is column normalR_col of R{normalR_box}
    V3<T> normalC = { T(0), T(0), T(0) };

#define DMX_TST1(expr1, expr2, box, colj, cc)                                            \
    e = (expr1); s2 = tabs(e) - (expr2);                                                 \
    if (s2 > 0) return 0;                                                                \
    if (s2 > s) { s = s2; normalR_box = (box); normalR_col = (colj); invert_normal = (e < 0); code = (cc); }

    DMX_TST1(pp[0], (A[0] + B[0] * Q[0][0] + B[1] * Q[0][1] + B[2] * Q[0][2]), 1, 0, 1);
    DMX_TST1(pp[1], (A[1] + B[0] * Q[1][0] + B[1] * Q[1][1] + B[2] * Q[1][2]), 1, 1, 2);
    DMX_TST1(pp[2], (A[2] + B[0] * Q[2][0] + B[1] * Q[2][1] + B[2] * Q[2][2]), 1, 2, 3);
    DMX_TST1(dot(colv(R2, 0), p), (A[0] * Q[0][0] + A[1] * Q[1][0] + A[2] * Q[2][0] + B[0]), 2, 0, 4);
    DMX_TST1(dot(colv(R2, 1), p), (A[0] * Q[0][1] + A[1] * Q[1][1] + A[2] * Q[2][1] + B[1]), 2, 1, 5);
    DMX_TST1(dot(colv(R2, 2), p), (A[0] * Q[0][2] + A[1] * Q[1][2] + A[2] * Q[2][2] + B[2]), 2, 2, 6);
#undef DMX_TST1

#define DMX_TST2(expr1, expr2, n1, n2, n3, cc)                                           \
    e = (expr1); s2 = tabs(e) - (expr2);                                                 \
    if (s2 > 0) return 0;                                                                \
    l = tsqrt<T>((n1) * (n1) + (n2) * (n2) + (n3) * (n3));                               \
    if (l > 0) {                                                                         \
        s2 /= l;                                                                         \
        if (s2 * fudge_factor > s) {                                                     \
            s = s2; normalR_box = 0;                                                     \
            normalC = { (n1) / l, (n2) / l, (n3) / l };                                  \
            invert_normal = (e < 0); code = (cc);                                        \
        }                                                                                \
    }
    // ODE's "fudge2" (box.cpp, 0.11 and later): an epsilon on every |R| entry before the nine edge-pair axes -- with two edges
    // (nearly) parallel |expr1| - expr2 is otherwise a difference of rounding errors, and a box lying flat on another is
    // "separated" whenever it happens to come out positive
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Q[i][j] += T(1.0e-5);
    // edge axes u_i x v_j
    DMX_TST2(pp[2] * Rr[1][0] - pp[1] * Rr[2][0], (A[1] * Q[2][0] + A[2] * Q[1][0] + B[1] * Q[0][2] + B[2] * Q[0][1]), T(0), -Rr[2][0], Rr[1][0], 7);
    DMX_TST2(pp[2] * Rr[1][1] - pp[1] * Rr[2][1], (A[1] * Q[2][1] + A[2] * Q[1][1] + B[0] * Q[0][2] + B[2] * Q[0][0]), T(0), -Rr[2][1], Rr[1][1], 8);
    DMX_TST2(pp[2] * Rr[1][2] - pp[1] * Rr[2][2], (A[1] * Q[2][2] + A[2] * Q[1][2] + B[0] * Q[0][1] + B[1] * Q[0][0]), T(0), -Rr[2][2], Rr[1][2], 9);
    DMX_TST2(pp[0] * Rr[2][0] - pp[2] * Rr[0][0], (A[0] * Q[2][0] + A[2] * Q[0][0] + B[1] * Q[1][2] + B[2] * Q[1][1]), Rr[2][0], T(0), -Rr[0][0], 10);
    DMX_TST2(pp[0] * Rr[2][1] - pp[2] * Rr[0][1], (A[0] * Q[2][1] + A[2] * Q[0][1] + B[0] * Q[1][2] + B[2] * Q[1][0]), Rr[2][1], T(0), -Rr[0][1], 11);
    DMX_TST2(pp[0] * Rr[2][2] - pp[2] * Rr[0][2], (A[0] * Q[2][2] + A[2] * Q[0][2] + B[0] * Q[1][1] + B[1] * Q[1][0]), Rr[2][2], T(0), -Rr[0][2], 12);
    DMX_TST2(pp[1] * Rr[0][0] - pp[0] * Rr[1][0], (A[0] * Q[1][0] + A[1] * Q[0][0] + B[1] * Q[2][2] + B[2] * Q[2][1]), -Rr[1][0], Rr[0][0], T(0), 13);
    DMX_TST2(pp[1] * Rr[0][1] - pp[0] * Rr[1][1], (A[0] * Q[1][1] + A[1] * Q[0][1] + B[0] * Q[2][2] + B[2] * Q[2][0]), -Rr[1][1], Rr[0][1], T(0), 14);
    DMX_TST2(pp[1] * Rr[0][2] - pp[0] * Rr[1][2], (A[0] * Q[1][2] + A[1] * Q[0][2] + B[0] * Q[2][1] + B[1] * Q[2][0]), -Rr[1][2], Rr[0][2], T(0), 15);
#undef DMX_TST2

    if (!code) return 0;

    V3<T> normal;
    if (normalR_box == 1) normal = colv_sel(R1, normalR_col);
    else if (normalR_box == 2) normal = colv_sel(R2, normalR_col);
    else normal = mulv(R1, normalC);
    if (invert_normal) normal = { -normal.x, -normal.y, -normal.z };
    const T depth = -s;
    const V3<T> cn = { -normal.x, -normal.y, -normal.z };      // contact normal: into box 1

    if (code > 6) {
        V3<T> pa = p1, pb = p2;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const V3<T> cj = colv(R1, j);
            const T sign = (dot(normal, cj) > 0) ? T(1) : T(-1);
            pa.x += sign * A[j] * cj.x; pa.y += sign * A[j] * cj.y; pa.z += sign * A[j] * cj.z;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const V3<T> cj = colv(R2, j);
            const T sign = (dot(normal, cj) > 0) ? T(-1) : T(1);
            pb.x += sign * B[j] * cj.x; pb.y += sign * B[j] * cj.y; pb.z += sign * B[j] * cj.z;
        }
        const V3<T> ua = colv_sel(R1, (code - 7) / 3), ub = colv_sel(R2, (code - 7) % 3);
        T alpha, beta;
        detail::line_closest_approach(pa, ua, pb, ub, alpha, beta);
        pa.x += ua.x * alpha; pa.y += ua.y * alpha; pa.z += ua.z * alpha;
        pb.x += ub.x * beta; pb.y += ub.y * beta; pb.z += ub.z * beta;
        out[0].pos = { T(0.5) * (pa.x + pb.x), T(0.5) * (pa.y + pb.y), T(0.5) * (pa.z + pb.z) };
        out[0].depth = depth;
        out[0].normal = cn;
        return 1;
    }

    // face contact: 'a' = box owning the reference face, 'b' = incident box
    const bool ref1 = code <= 3;
    const M3<T> Ra = m3_sel(ref1, R1, R2), Rb = m3_sel(ref1, R2, R1);
    const V3<T> pa = ref1 ? p1 : p2, pb = ref1 ? p2 : p1;
    const T Sa[3] = { ref1 ? A[0] : B[0], ref1 ? A[1] : B[1], ref1 ? A[2] : B[2] };
    const T Sb[3] = { ref1 ? B[0] : A[0], ref1 ? B[1] : A[1], ref1 ? B[2] : A[2] };
    const V3<T> normal2 = ref1 ? normal : V3<T>{ -normal.x, -normal.y, -normal.z };
    const T nr[3] = { dot(colv(Rb, 0), normal2), dot(colv(Rb, 1), normal2), dot(colv(Rb, 2), normal2) };
    const T anr[3] = { tabs(nr[0]), tabs(nr[1]), tabs(nr[2]) };
    int lanr, a1, a2;
    if (anr[1] > anr[0]) {
        if (anr[1] > anr[2]) { a1 = 0; lanr = 1; a2 = 2; }
        else { a1 = 0; a2 = 1; lanr = 2; }
    } else {
        if (anr[0] > anr[2]) { lanr = 0; a1 = 1; a2 = 2; }
        else { a1 = 0; a2 = 1; lanr = 2; }
    }
    const V3<T> bl = colv_sel(Rb, lanr);
    const T Sbl = pick(Sb, lanr), nrl = pick(nr, lanr);
    V3<T> center;
    if (nrl < 0)
        center = { pb.x - pa.x + Sbl * bl.x, pb.y - pa.y + Sbl * bl.y, pb.z - pa.z + Sbl * bl.z };
    else
        center = { pb.x - pa.x - Sbl * bl.x, pb.y - pa.y - Sbl * bl.y, pb.z - pa.z - Sbl * bl.z };
    const int codeN = ref1 ? code - 1 : code - 4;
    const int code1 = codeN == 0 ? 1 : 0;
    const int code2 = codeN == 2 ? 1 : 2;

    const V3<T> ra1 = colv_sel(Ra, code1), ra2 = colv_sel(Ra, code2), rb1 = colv_sel(Rb, a1), rb2 = colv_sel(Rb, a2);
    const T c1 = dot(center, ra1), c2 = dot(center, ra2);
    T m11 = dot(ra1, rb1), m12 = dot(ra1, rb2), m21 = dot(ra2, rb1), m22 = dot(ra2, rb2);
    T quad[8];
    {
        const T Sb1 = pick(Sb, a1), Sb2 = pick(Sb, a2);
        const T k1 = m11 * Sb1, k2 = m21 * Sb1, k3 = m12 * Sb2, k4 = m22 * Sb2;
        quad[0] = c1 - k1 - k3; quad[1] = c2 - k2 - k4;
        quad[2] = c1 - k1 + k3; quad[3] = c2 - k2 + k4;
        quad[4] = c1 + k1 + k3; quad[5] = c2 + k2 + k4;
        quad[6] = c1 + k1 - k3; quad[7] = c2 + k2 - k4;
    }
    const T rect[2] = { pick(Sa, code1), pick(Sa, code2) };
    detail::Poly8<T> ret;
    const int n = detail::intersect_rect_quad(rect, quad, ret);
    if (n < 1) return 0;

    // clipped 2-D points -> 3-D points on the incident face and their depths; the ones not below the reference face are kept,
    // in order (slot `cnum` is a running count: select chains again)
    V3<T> point[8];
    T dep[8];
    detail::Poly8<T> kept;      // the 2-D points of the kept contacts (cullPoints reads them)
#pragma unroll
    for (int j = 0; j < 8; j++) { point[j] = { T(0), T(0), T(0) }; dep[j] = T(0); kept.x[j] = kept.y[j] = T(0); }
    const T det1 = T(1) / (m11 * m22 - m12 * m21);
    m11 *= det1; m12 *= det1; m21 *= det1; m22 *= det1;
    const T Sn = pick(Sa, codeN);
    int cnum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j < n) {
            const T k1 = m22 * (ret.x[j] - c1) - m12 * (ret.y[j] - c2);
            const T k2 = -m21 * (ret.x[j] - c1) + m11 * (ret.y[j] - c2);
            const V3<T> pt = { center.x + k1 * rb1.x + k2 * rb2.x, center.y + k1 * rb1.y + k2 * rb2.y,
                               center.z + k1 * rb1.z + k2 * rb2.z };
            const T dj = Sn - dot(normal2, pt);
            if (dj >= 0) {
#pragma unroll
                for (int q = 0; q <= j; q++) {              // (cnum <= j)
                    const bool here = q == cnum;
                    point[q].x = here ? pt.x : point[q].x; point[q].y = here ? pt.y : point[q].y; point[q].z = here ? pt.z : point[q].z;
                    dep[q] = here ? dj : dep[q];
                    kept.x[q] = here ? ret.x[j] : kept.x[q]; kept.y[q] = here ? ret.y[j] : kept.y[q];
                }
                cnum++;
            }
        }
    }
    if (cnum < 1) return 0;

    int maxc = maxc_in;
    if (maxc > cnum) maxc = cnum;
    if (maxc < 1) maxc = 1;
    if (cnum <= maxc) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j < cnum) {
                V3<T> o = { point[j].x + pa.x, point[j].y + pa.y, point[j].z + pa.z };
                if (!ref1) { o.x -= normal.x * dep[j]; o.y -= normal.y * dep[j]; o.z -= normal.z * dep[j]; }
                out[j].pos = o;
                out[j].depth = dep[j];
                out[j].normal = cn;
            }
        }
        return cnum;
    }
    // fewer contacts wanted than found: the deepest point, then cullPoints' angular selection (its arrays are walked with
    // running indices: the one place that stays in memory on the device; never reached with the reference's MAX_CONTACTS = 8)
    T p2d[16];
#pragma unroll
    for (int j = 0; j < 8; j++) { p2d[2 * j] = kept.x[j]; p2d[2 * j + 1] = kept.y[j]; }
    int i1 = 0;
    T maxdepth = dep[0];
#pragma unroll
    for (int i = 1; i < 8; i++) if (i < cnum && dep[i] > maxdepth) { maxdepth = dep[i]; i1 = i; }
    int iret[8];
    cull_points<T>(cnum, p2d, maxc, i1, iret);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j < maxc) {
            const int k = iret[j];
            V3<T> pt = point[0];
            T dk = dep[0];
#pragma unroll
            for (int q = 1; q < 8; q++) { const bool h = q == k; pt.x = h ? point[q].x : pt.x; pt.y = h ? point[q].y : pt.y; pt.z = h ? point[q].z : pt.z; dk = h ? dep[q] : dk; }
            out[j].pos = { pt.x + pa.x, pt.y + pa.y, pt.z + pa.z };
            out[j].depth = dk; out[j].normal = cn;
        }
    }
    return maxc;
}

}  // namespace dmx
