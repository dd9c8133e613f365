/*
 * oracle/orc_math.h -- small vector / matrix / quaternion helpers (TEST
 * INFRASTRUCTURE, see orc.h).  Matrices are ODE-layout 3x4 row-major
 * (dMatrix3 = dReal[12], element (i,j) at [4*i+j]; the reference indexes
 * them that way at main.c:603-616); quaternions are (w,x,y,z).
 * Each helper restates the ODE routine named in its comment [ODE-recall].
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include "orc.h"

#ifdef ORC_SINGLE
#define orc_sqrt  sqrtf
#define orc_fabs  fabsf
#define ORC_INF   ((real)INFINITY)
#define ORC_SQRT1_2 0.70710678118654752440f
#define ORC_EPS   1.1920928955078125e-07f
#else
#define orc_sqrt  sqrt
#define orc_fabs  fabs
#define ORC_INF   ((real)INFINITY)
#define ORC_SQRT1_2 0.70710678118654752440
#define ORC_EPS   2.2204460492503131e-16
#endif

#define R(i)   ((real)(i))

/* The step is specified with fused multiply-adds in its dot products, cross products and a*x+y updates
 * (one rounding per fma, z-term last), in the oracle and in the HIP kernels alike: FMA(a,b,c) = a*b + c. */
#ifdef ORC_SINGLE
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#else
#define FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif

/* dCalcVectorDot3: a0*b0 + a1*b1 + a2*b2, summed left to right */
static inline real orc_dot3(const real *a, const real *b)
{
    return FMA(a[2], b[2], FMA(a[1], b[1], a[0] * b[0]));
}

/* dCalcVectorDot3_14: b strided by 4 (a column of a 3x4 matrix) */
static inline real orc_dot3_14(const real *a, const real *b)
{
    return FMA(a[2], b[8], FMA(a[1], b[4], a[0] * b[0]));
}

/* dCalcVectorCross3: a = b x c */
static inline void orc_cross3(real *a, const real *b, const real *c)
{
    a[0] = FMA(b[1], c[2], -(b[2] * c[1]));
    a[1] = FMA(b[2], c[0], -(b[0] * c[2]));
    a[2] = FMA(b[0], c[1], -(b[1] * c[0]));
}

/* dMultiply0_331: a = B * c, B 3x4 */
static inline void orc_mul0_331(real *a, const real *B, const real *c)
{
    real r0 = orc_dot3(B + 0, c);
    real r1 = orc_dot3(B + 4, c);
    real r2 = orc_dot3(B + 8, c);
    a[0] = r0; a[1] = r1; a[2] = r2;
}

/* dMultiplyAdd0_331: a += B * c */
static inline void orc_muladd0_331(real *a, const real *B, const real *c)
{
    real r0 = orc_dot3(B + 0, c);
    real r1 = orc_dot3(B + 4, c);
    real r2 = orc_dot3(B + 8, c);
    a[0] += r0; a[1] += r1; a[2] += r2;
}

/* dMultiply0_333: A = B * C */
static inline void orc_mul0_333(real *A, const real *B, const real *C)
{
    real T[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            T[4 * i + j] = orc_dot3_14(B + 4 * i, C + j);
        T[4 * i + 3] = 0;
    }
    for (int i = 0; i < 12; i++) A[i] = T[i];
}

/* dMultiply2_333: A = B * C^T */
static inline void orc_mul2_333(real *A, const real *B, const real *C)
{
    real T[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            T[4 * i + j] = orc_dot3(B + 4 * i, C + 4 * j);
        T[4 * i + 3] = 0;
    }
    for (int i = 0; i < 12; i++) A[i] = T[i];
}

/* dCalcMatrix3Det */
static inline real orc_det3(const real *m)
{
    real m0 = FMA(m[5], m[10], -(m[9] * m[6]));
    real m1 = FMA(m[4], m[10], -(m[8] * m[6]));
    real m2 = FMA(m[4], m[9], -(m[8] * m[5]));
    return FMA(m[2], m2, FMA(-m[1], m1, m[0] * m0));
}

/* dInvertMatrix3: closed-form adjugate / det; returns 0 if singular */
static inline int orc_invert3(real *dst, const real *ma)
{
    real det = orc_det3(ma);
    if (det == 0) return 0;
    real dr = R(1.0) / det;
    dst[0]  = FMA(ma[5], ma[10], -(ma[6] * ma[9])) * dr;
    dst[1]  = FMA(ma[9], ma[2], -(ma[1] * ma[10])) * dr;
    dst[2]  = FMA(ma[1], ma[6], -(ma[5] * ma[2])) * dr;
    dst[3]  = 0;
    dst[4]  = FMA(ma[6], ma[8], -(ma[4] * ma[10])) * dr;
    dst[5]  = FMA(ma[0], ma[10], -(ma[8] * ma[2])) * dr;
    dst[6]  = FMA(ma[4], ma[2], -(ma[0] * ma[6])) * dr;
    dst[7]  = 0;
    dst[8]  = FMA(ma[4], ma[9], -(ma[8] * ma[5])) * dr;
    dst[9]  = FMA(ma[8], ma[1], -(ma[0] * ma[9])) * dr;
    dst[10] = FMA(ma[0], ma[5], -(ma[1] * ma[4])) * dr;
    dst[11] = 0;
    return 1;
}

/* dQtoR */
static inline void orc_q_to_R(const real *q, real *Rm)
{
    real qq1 = 2 * q[1] * q[1];
    real qq2 = 2 * q[2] * q[2];
    real qq3 = 2 * q[3] * q[3];
    Rm[0]  = 1 - qq2 - qq3;
    Rm[1]  = 2 * FMA(q[1], q[2], -(q[0] * q[3]));
    Rm[2]  = 2 * FMA(q[1], q[3], q[0] * q[2]);
    Rm[3]  = 0;
    Rm[4]  = 2 * FMA(q[1], q[2], q[0] * q[3]);
    Rm[5]  = 1 - qq1 - qq3;
    Rm[6]  = 2 * FMA(q[2], q[3], -(q[0] * q[1]));
    Rm[7]  = 0;
    Rm[8]  = 2 * FMA(q[1], q[3], -(q[0] * q[2]));
    Rm[9]  = 2 * FMA(q[2], q[3], q[0] * q[1]);
    Rm[10] = 1 - qq1 - qq2;
    Rm[11] = 0;
}

/* dRtoQ (dQfromR): trace / largest-diagonal branches */
static inline void orc_R_to_q(const real *Rm, real *q)
{
#define _R(i, j) Rm[(i) * 4 + (j)]
    real tr = _R(0, 0) + _R(1, 1) + _R(2, 2), s;
    if (tr >= 0) {
        s = orc_sqrt(tr + 1);
        q[0] = R(0.5) * s;
        s = R(0.5) * (R(1.0) / s);
        q[1] = (_R(2, 1) - _R(1, 2)) * s;
        q[2] = (_R(0, 2) - _R(2, 0)) * s;
        q[3] = (_R(1, 0) - _R(0, 1)) * s;
    } else if (_R(1, 1) > _R(0, 0) && !(_R(2, 2) > _R(1, 1))) {
        s = orc_sqrt((_R(1, 1) - (_R(2, 2) + _R(0, 0))) + 1);
        q[2] = R(0.5) * s;
        s = R(0.5) * (R(1.0) / s);
        q[3] = (_R(1, 2) + _R(2, 1)) * s;
        q[1] = (_R(0, 1) + _R(1, 0)) * s;
        q[0] = (_R(0, 2) - _R(2, 0)) * s;
    } else if (_R(2, 2) > _R(0, 0) && _R(2, 2) > _R(1, 1)) {
        s = orc_sqrt((_R(2, 2) - (_R(0, 0) + _R(1, 1))) + 1);
        q[3] = R(0.5) * s;
        s = R(0.5) * (R(1.0) / s);
        q[1] = (_R(2, 0) + _R(0, 2)) * s;
        q[2] = (_R(1, 2) + _R(2, 1)) * s;
        q[0] = (_R(1, 0) - _R(0, 1)) * s;
    } else {
        s = orc_sqrt((_R(0, 0) - (_R(1, 1) + _R(2, 2))) + 1);
        q[1] = R(0.5) * s;
        s = R(0.5) * (R(1.0) / s);
        q[2] = (_R(0, 1) + _R(1, 0)) * s;
        q[3] = (_R(2, 0) + _R(0, 2)) * s;
        q[0] = (_R(2, 1) - _R(1, 2)) * s;
    }
#undef _R
}

/* dNormalize4 (_dSafeNormalize4 + fallback to identity) */
static inline void orc_normalize4(real *a)
{
    real l = FMA(a[3], a[3], FMA(a[2], a[2], FMA(a[1], a[1], a[0] * a[0])));
    if (l > 0) {
        l = R(1.0) / orc_sqrt(l);   /* dRecipSqrt */
        a[0] *= l; a[1] *= l; a[2] *= l; a[3] *= l;
    } else {
        a[0] = 1; a[1] = 0; a[2] = 0; a[3] = 0;
    }
}

/* dDQfromW / dWtoDQ: dq = 1/2 (0,w) (x) q */
static inline void orc_w_to_dq(const real *w, const real *q, real *dq)
{
    dq[0] = R(0.5) * FMA(-w[2], q[3], FMA(-w[1], q[2], -w[0] * q[1]));
    dq[1] = R(0.5) * FMA(-w[2], q[2], FMA( w[1], q[3],  w[0] * q[0]));
    dq[2] = R(0.5) * FMA( w[2], q[1], FMA( w[1], q[0], -w[0] * q[3]));
    dq[3] = R(0.5) * FMA( w[2], q[0], FMA(-w[1], q[1],  w[0] * q[2]));
}

/* dPlaneSpace: p,q orthonormal to unit n */
static inline void orc_plane_space(const real *n, real *p, real *q)
{
    if (orc_fabs(n[2]) > ORC_SQRT1_2) {
        real a = n[1] * n[1] + n[2] * n[2];
        real k = R(1.0) / orc_sqrt(a);
        p[0] = 0;
        p[1] = -n[2] * k;
        p[2] = n[1] * k;
        q[0] = a * k;
        q[1] = -n[0] * p[2];
        q[2] = n[0] * p[1];
    } else {
        real a = n[0] * n[0] + n[1] * n[1];
        real k = R(1.0) / orc_sqrt(a);
        p[0] = -n[1] * k;
        p[1] = n[0] * k;
        p[2] = 0;
        q[0] = -n[2] * p[1];
        q[1] = n[2] * p[0];
        q[2] = a * k;
    }
}

#endif
