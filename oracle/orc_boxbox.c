/*
 * oracle/orc_boxbox.c -- dCollideBoxBox / dBoxBox restated [ODE-recall box.cpp]:
 * 15-axis separating-axis test, then either the edge-edge closest-point
 * contact or reference-face / incident-face clipping (<= 8 contacts).
 * TEST INFRASTRUCTURE (see orc.h).  The reference reaches this collider for
 * every dynamic box against its static-box floor and walls (main.c:115-121,
 * 720, 743) and for box piles.
 */
#include <math.h>
#include <string.h>
#include "orc_internal.h"

#ifdef ORC_SINGLE
#define orc_atan2 atan2f
#else
#define orc_atan2 atan2
#endif

static real dot44(const real *a, const real *b) { return FMA(a[8], b[8], FMA(a[4], b[4], a[0] * b[0])); }
static real dot41(const real *a, const real *b) { return FMA(a[8], b[2], FMA(a[4], b[1], a[0] * b[0])); }

/* dLineClosestApproach */
static void line_closest_approach(const real *pa, const real *ua, const real *pb, const real *ub,
                                  real *alpha, real *beta)
{
    real p[3] = { pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2] };
    real uaub = orc_dot3(ua, ub);
    real q1 = orc_dot3(ua, p);
    real q2 = -orc_dot3(ub, p);
    real d = 1 - uaub * uaub;
    if (d <= R(0.0001)) { *alpha = 0; *beta = 0; }
    else {
        d = R(1.0) / d;
        *alpha = (q1 + uaub * q2) * d;
        *beta = (uaub * q1 + q2) * d;
    }
}

/* intersectRectQuad: clip quad p[8] against the rectangle |x|<=h[0], |y|<=h[1] */
static int intersect_rect_quad(const real h[2], real p[8], real ret[16])
{
    int nq = 4, nr = 0;
    real buffer[16];
    real *q = p, *r = ret;
    for (int dir = 0; dir <= 1; dir++) {
        for (int sign = -1; sign <= 1; sign += 2) {
            real *pq = q, *pr = r;
            nr = 0;
            for (int i = nq; i > 0; i--) {
                if (sign * pq[dir] < h[dir]) {
                    pr[0] = pq[0]; pr[1] = pq[1];
                    pr += 2; nr++;
                    if (nr & 8) { q = r; goto done; }
                }
                real *nextq = (i > 1) ? pq + 2 : q;
                if ((sign * pq[dir] < h[dir]) ^ (sign * nextq[dir] < h[dir])) {
                    pr[1 - dir] = pq[1 - dir] + (nextq[1 - dir] - pq[1 - dir]) /
                                  (nextq[dir] - pq[dir]) * (sign * h[dir] - pq[dir]);
                    pr[dir] = sign * h[dir];
                    pr += 2; nr++;
                    if (nr & 8) { q = r; goto done; }
                }
                pq += 2;
            }
            q = r;
            r = (q == ret) ? buffer : ret;
            nq = nr;
        }
    }
done:
    if (q != ret) memcpy(ret, q, (size_t)nr * 2 * sizeof(real));
    return nr;
}

/* cullPoints: keep m of n 2-D points, i0 first, spread by angle about the centroid */
static void cull_points(int n, real p[], int m, int i0, int iret[])
{
    real a, cx, cy, q;
    if (n == 1) { cx = p[0]; cy = p[1]; }
    else if (n == 2) { cx = R(0.5) * (p[0] + p[2]); cy = R(0.5) * (p[1] + p[3]); }
    else {
        a = 0; cx = 0; cy = 0;
        for (int i = 0; i < n - 1; i++) {
            q = p[i * 2] * p[i * 2 + 3] - p[i * 2 + 2] * p[i * 2 + 1];
            a += q;
            cx += q * (p[i * 2] + p[i * 2 + 2]);
            cy += q * (p[i * 2 + 1] + p[i * 2 + 3]);
        }
        q = p[n * 2 - 2] * p[1] - p[0] * p[n * 2 - 1];
        a = R(1.0) / (R(3.0) * (a + q));
        cx = a * (cx + q * (p[n * 2 - 2] + p[0]));
        cy = a * (cy + q * (p[n * 2 - 1] + p[1]));
    }
    real A[8];
    for (int i = 0; i < n; i++) A[i] = orc_atan2(p[i * 2 + 1] - cy, p[i * 2] - cx);
    int avail[8];
    for (int i = 0; i < n; i++) avail[i] = 1;
    avail[i0] = 0;
    iret[0] = i0;
    iret++;
    const real pi = R(3.14159265358979323846);
    for (int j = 1; j < m; j++) {
        a = (real)((real)j * (2 * pi / m) + A[i0]);
        if (a > pi) a -= 2 * pi;
        real maxdiff = R(1e9), diff;
        *iret = i0;
        for (int i = 0; i < n; i++) {
            if (avail[i]) {
                diff = orc_fabs(A[i] - a);
                if (diff > pi) diff = 2 * pi - diff;
                if (diff < maxdiff) { maxdiff = diff; *iret = i; }
            }
        }
        avail[*iret] = 0;
        iret++;
    }
}

/* dBoxBox; returns contact count, fills out[].pos/depth and *normal (box1 -> box2) */
static int box_box(const real *p1, const real *R1, const real *side1,
                   const real *p2, const real *R2, const real *side2,
                   real normal[3], int maxc_in, orc_contactgeom *out)
{
    const real fudge_factor = R(1.05);
    real p[3], pp[3], normalC[3] = { 0, 0, 0 };
    const real *normalR = 0;
    real A[3], B[3], Rr[3][3], Q[3][3], s, s2, l, e;
    int invert_normal, code;

    p[0] = p2[0] - p1[0]; p[1] = p2[1] - p1[1]; p[2] = p2[2] - p1[2];
    pp[0] = dot41(R1 + 0, p); pp[1] = dot41(R1 + 1, p); pp[2] = dot41(R1 + 2, p);   /* R1^T p */
    for (int i = 0; i < 3; i++) { A[i] = side1[i] * R(0.5); B[i] = side2[i] * R(0.5); }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { Rr[i][j] = dot44(R1 + i, R2 + j); Q[i][j] = orc_fabs(Rr[i][j]); }

    s = -ORC_INF; invert_normal = 0; code = 0;

#define TST1(expr1, expr2, norm, cc) \
    e = (expr1); s2 = orc_fabs(e) - (expr2); \
    if (s2 > 0) return 0; \
    if (s2 > s) { s = s2; normalR = (norm); invert_normal = (e < 0); code = (cc); }

    /* separating axis = u1,u2,u3 */
    TST1(pp[0], (A[0] + B[0] * Q[0][0] + B[1] * Q[0][1] + B[2] * Q[0][2]), R1 + 0, 1);
    TST1(pp[1], (A[1] + B[0] * Q[1][0] + B[1] * Q[1][1] + B[2] * Q[1][2]), R1 + 1, 2);
    TST1(pp[2], (A[2] + B[0] * Q[2][0] + B[1] * Q[2][1] + B[2] * Q[2][2]), R1 + 2, 3);
    /* separating axis = v1,v2,v3 */
    TST1(dot41(R2 + 0, p), (A[0] * Q[0][0] + A[1] * Q[1][0] + A[2] * Q[2][0] + B[0]), R2 + 0, 4);
    TST1(dot41(R2 + 1, p), (A[0] * Q[0][1] + A[1] * Q[1][1] + A[2] * Q[2][1] + B[1]), R2 + 1, 5);
    TST1(dot41(R2 + 2, p), (A[0] * Q[0][2] + A[1] * Q[1][2] + A[2] * Q[2][2] + B[2]), R2 + 2, 6);
#undef TST1

#define TST2(expr1, expr2, n1, n2, n3, cc) \
    e = (expr1); s2 = orc_fabs(e) - (expr2); \
    if (s2 > 0) return 0; \
    l = orc_sqrt((n1) * (n1) + (n2) * (n2) + (n3) * (n3)); \
    if (l > 0) { \
        s2 /= l; \
        if (s2 * fudge_factor > s) { \
            s = s2; normalR = 0; \
            normalC[0] = (n1) / l; normalC[1] = (n2) / l; normalC[2] = (n3) / l; \
            invert_normal = (e < 0); code = (cc); \
        } \
    }

#define R11 Rr[0][0]
#define R12 Rr[0][1]
#define R13 Rr[0][2]
#define R21 Rr[1][0]
#define R22 Rr[1][1]
#define R23 Rr[1][2]
#define R31 Rr[2][0]
#define R32 Rr[2][1]
#define R33 Rr[2][2]
#define Q11 Q[0][0]
#define Q12 Q[0][1]
#define Q13 Q[0][2]
#define Q21 Q[1][0]
#define Q22 Q[1][1]
#define Q23 Q[1][2]
#define Q31 Q[2][0]
#define Q32 Q[2][1]
#define Q33 Q[2][2]
    /* [ODE-recall box.cpp, 0.11 and later] "fudge2": an epsilon added to every |R| entry before the nine edge-pair axes, to
     * counteract arithmetic error when two edges are (nearly) parallel and their cross product (nearly) vanishes: without it
     * |expr1| - expr2 is a difference of two rounding errors there and a box lying flat on another is "separated" whenever
     * that difference happens to come out positive (tests/test_collider_geometry.py met it twice in 20 000 resting boxes). */
    {
        const real fudge2 = R(1.0e-5);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[i][j] += fudge2;
    }
    /* separating axis = u1 x (v1,v2,v3) */
    TST2(pp[2] * R21 - pp[1] * R31, (A[1] * Q31 + A[2] * Q21 + B[1] * Q13 + B[2] * Q12), 0, -R31, R21, 7);
    TST2(pp[2] * R22 - pp[1] * R32, (A[1] * Q32 + A[2] * Q22 + B[0] * Q13 + B[2] * Q11), 0, -R32, R22, 8);
    TST2(pp[2] * R23 - pp[1] * R33, (A[1] * Q33 + A[2] * Q23 + B[0] * Q12 + B[1] * Q11), 0, -R33, R23, 9);
    /* separating axis = u2 x (v1,v2,v3) */
    TST2(pp[0] * R31 - pp[2] * R11, (A[0] * Q31 + A[2] * Q11 + B[1] * Q23 + B[2] * Q22), R31, 0, -R11, 10);
    TST2(pp[0] * R32 - pp[2] * R12, (A[0] * Q32 + A[2] * Q12 + B[0] * Q23 + B[2] * Q21), R32, 0, -R12, 11);
    TST2(pp[0] * R33 - pp[2] * R13, (A[0] * Q33 + A[2] * Q13 + B[0] * Q22 + B[1] * Q21), R33, 0, -R13, 12);
    /* separating axis = u3 x (v1,v2,v3) */
    TST2(pp[1] * R11 - pp[0] * R21, (A[0] * Q21 + A[1] * Q11 + B[1] * Q33 + B[2] * Q32), -R21, R11, 0, 13);
    TST2(pp[1] * R12 - pp[0] * R22, (A[0] * Q22 + A[1] * Q12 + B[0] * Q33 + B[2] * Q31), -R22, R12, 0, 14);
    TST2(pp[1] * R13 - pp[0] * R23, (A[0] * Q23 + A[1] * Q13 + B[0] * Q32 + B[1] * Q31), -R23, R13, 0, 15);
#undef TST2

    if (!code) return 0;

    /* normal in global coordinates */
    if (normalR) { normal[0] = normalR[0]; normal[1] = normalR[4]; normal[2] = normalR[8]; }
    else orc_mul0_331(normal, R1, normalC);
    if (invert_normal) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
    real depth = -s;

    if (code > 6) {
        /* edge-edge: closest points of the two touching edges */
        real pa[3] = { p1[0], p1[1], p1[2] }, pb[3] = { p2[0], p2[1], p2[2] };
        for (int j = 0; j < 3; j++) {
            real sign = (orc_dot3_14(normal, R1 + j) > 0) ? R(1.0) : R(-1.0);
            for (int i = 0; i < 3; i++) pa[i] += sign * A[j] * R1[i * 4 + j];
        }
        for (int j = 0; j < 3; j++) {
            real sign = (orc_dot3_14(normal, R2 + j) > 0) ? R(-1.0) : R(1.0);
            for (int i = 0; i < 3; i++) pb[i] += sign * B[j] * R2[i * 4 + j];
        }
        real alpha, beta, ua[3], ub[3];
        for (int i = 0; i < 3; i++) ua[i] = R1[(code - 7) / 3 + i * 4];
        for (int i = 0; i < 3; i++) ub[i] = R2[(code - 7) % 3 + i * 4];
        line_closest_approach(pa, ua, pb, ub, &alpha, &beta);
        for (int i = 0; i < 3; i++) pa[i] += ua[i] * alpha;
        for (int i = 0; i < 3; i++) pb[i] += ub[i] * beta;
        for (int i = 0; i < 3; i++) out[0].pos[i] = R(0.5) * (pa[i] + pb[i]);
        out[0].depth = depth;
        return 1;
    }

    /* face-something: 'a' = reference face (normal axis), 'b' = incident box */
    const real *Ra, *Rb, *pa, *pb, *Sa, *Sb;
    if (code <= 3) { Ra = R1; Rb = R2; pa = p1; pb = p2; Sa = A; Sb = B; }
    else           { Ra = R2; Rb = R1; pa = p2; pb = p1; Sa = B; Sb = A; }
    real normal2[3], nr[3], anr[3];
    for (int i = 0; i < 3; i++) normal2[i] = (code <= 3) ? normal[i] : -normal[i];
    nr[0] = dot41(Rb + 0, normal2); nr[1] = dot41(Rb + 1, normal2); nr[2] = dot41(Rb + 2, normal2);
    anr[0] = orc_fabs(nr[0]); anr[1] = orc_fabs(nr[1]); anr[2] = orc_fabs(nr[2]);
    int lanr, a1, a2;
    if (anr[1] > anr[0]) {
        if (anr[1] > anr[2]) { a1 = 0; lanr = 1; a2 = 2; }
        else { a1 = 0; a2 = 1; lanr = 2; }
    } else {
        if (anr[0] > anr[2]) { lanr = 0; a1 = 1; a2 = 2; }
        else { a1 = 0; a2 = 1; lanr = 2; }
    }
    real center[3];
    if (nr[lanr] < 0)
        for (int i = 0; i < 3; i++) center[i] = pb[i] - pa[i] + Sb[lanr] * Rb[i * 4 + lanr];
    else
        for (int i = 0; i < 3; i++) center[i] = pb[i] - pa[i] - Sb[lanr] * Rb[i * 4 + lanr];
    int codeN = (code <= 3) ? code - 1 : code - 4, code1, code2;
    if (codeN == 0) { code1 = 1; code2 = 2; }
    else if (codeN == 1) { code1 = 0; code2 = 2; }
    else { code1 = 0; code2 = 1; }

    real quad[8], c1, c2, m11, m12, m21, m22;
    c1 = orc_dot3_14(center, Ra + code1);
    c2 = orc_dot3_14(center, Ra + code2);
    m11 = dot44(Ra + code1, Rb + a1);
    m12 = dot44(Ra + code1, Rb + a2);
    m21 = dot44(Ra + code2, Rb + a1);
    m22 = dot44(Ra + code2, Rb + a2);
    {
        real k1 = m11 * Sb[a1], k2 = m21 * Sb[a1], k3 = m12 * Sb[a2], k4 = m22 * Sb[a2];
        quad[0] = c1 - k1 - k3; quad[1] = c2 - k2 - k4;
        quad[2] = c1 - k1 + k3; quad[3] = c2 - k2 + k4;
        quad[4] = c1 + k1 + k3; quad[5] = c2 + k2 + k4;
        quad[6] = c1 + k1 - k3; quad[7] = c2 + k2 - k4;
    }
    real rect[2] = { Sa[code1], Sa[code2] };
    real ret[16];
    int n = intersect_rect_quad(rect, quad, ret);
    if (n < 1) return 0;

    real point[3 * 8], dep[8];
    real det1 = R(1.0) / (m11 * m22 - m12 * m21);
    m11 *= det1; m12 *= det1; m21 *= det1; m22 *= det1;
    int cnum = 0;
    for (int j = 0; j < n; j++) {
        real k1 = m22 * (ret[j * 2] - c1) - m12 * (ret[j * 2 + 1] - c2);
        real k2 = -m21 * (ret[j * 2] - c1) + m11 * (ret[j * 2 + 1] - c2);
        for (int i = 0; i < 3; i++)
            point[cnum * 3 + i] = center[i] + k1 * Rb[i * 4 + a1] + k2 * Rb[i * 4 + a2];
        dep[cnum] = Sa[codeN] - orc_dot3(normal2, point + cnum * 3);
        if (dep[cnum] >= 0) {
            ret[cnum * 2] = ret[j * 2];
            ret[cnum * 2 + 1] = ret[j * 2 + 1];
            cnum++;
        }
    }
    if (cnum < 1) return 0;

    int maxc = maxc_in;
    if (maxc > cnum) maxc = cnum;
    if (maxc < 1) maxc = 1;
    if (cnum <= maxc) {
        for (int j = 0; j < cnum; j++) {
            for (int i = 0; i < 3; i++) {
                out[j].pos[i] = point[j * 3 + i] + pa[i];
                if (code >= 4) out[j].pos[i] -= normal[i] * dep[j];
            }
            out[j].depth = dep[j];
        }
    } else {
        int i1 = 0;
        real maxdepth = dep[0];
        for (int i = 1; i < cnum; i++) if (dep[i] > maxdepth) { maxdepth = dep[i]; i1 = i; }
        int iret[8];
        cull_points(cnum, ret, maxc, i1, iret);
        for (int j = 0; j < maxc; j++) {
            for (int i = 0; i < 3; i++) out[j].pos[i] = point[iret[j] * 3 + i] + pa[i];
            out[j].depth = dep[iret[j]];
        }
        cnum = maxc;
    }
    return cnum;
}

/* dCollideBoxBox: contact normal = -(box1 -> box2 axis), i.e. it points into box 1 */
int orc_collide_box_box(const real *p1, const real *R1, const real *side1,
                        const real *p2, const real *R2, const real *side2,
                        int maxc, orc_contactgeom *out)
{
    real normal[3];
    int num = box_box(p1, R1, side1, p2, R2, side2, normal, maxc, out);
    for (int i = 0; i < num; i++) {
        out[i].normal[0] = -normal[0];
        out[i].normal[1] = -normal[1];
        out[i].normal[2] = -normal[2];
    }
    return num;
}
