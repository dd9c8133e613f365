// dmx_math.hpp -- per-body rigid-body arithmetic shared by the device kernels and
// the host side of the ODE-compatible API.  3x3 matrices are row-major without
// padding; quaternions are (w,x,y,z) as in ODE.  Every expression is written in
// the evaluation order the step is specified with (left-to-right sums, no FMA:
// the library is built with -ffp-contract=off; fused multiply-adds are explicit, fma_) so results are reproducible
// bit-for-bit across host and device.
#pragma once

#include <hip/hip_runtime.h>

#define DMX_HD __host__ __device__ __forceinline__

namespace dmx {

template <class T> struct Limits;
template <> struct Limits<float>  { static DMX_HD float  inf() { return __builtin_huge_valf(); } static DMX_HD float  nan() { return __builtin_nanf(""); } };
template <> struct Limits<double> { static DMX_HD double inf() { return __builtin_huge_val(); } static DMX_HD double nan() { return __builtin_nan(""); } };

template <class T> DMX_HD T tsqrt(T x);
template <> DMX_HD float  tsqrt<float>(float x)   { return __builtin_sqrtf(x); }
template <> DMX_HD double tsqrt<double>(double x) { return __builtin_sqrt(x); }
template <class T> DMX_HD T tabs(T x) { return x < T(0) ? -x : x; }
// The step is specified with fused multiply-adds in its dot products, cross products and a*x+y updates (one
// rounding per fma, z-term last); everything else is a plain rounded operation (-ffp-contract=off).
DMX_HD float  fma_(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
DMX_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <class T> struct V3 { T x, y, z; };
template <class T> struct Q4 { T w, x, y, z; };
template <class T> struct M3 { T m[3][3]; };

template <class T> DMX_HD T dot(const V3<T> &a, const V3<T> &b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
template <class T> DMX_HD V3<T> cross(const V3<T> &b, const V3<T> &c)
{
    return { fma_(b.y, c.z, -(b.z * c.y)), fma_(b.z, c.x, -(b.x * c.z)), fma_(b.x, c.y, -(b.y * c.x)) };
}
template <class T> DMX_HD V3<T> mulv(const M3<T> &B, const V3<T> &c)
{
    return { fma_(B.m[0][2], c.z, fma_(B.m[0][1], c.y, B.m[0][0] * c.x)),
             fma_(B.m[1][2], c.z, fma_(B.m[1][1], c.y, B.m[1][0] * c.x)),
             fma_(B.m[2][2], c.z, fma_(B.m[2][1], c.y, B.m[2][0] * c.x)) };
}
// A = B * C
template <class T> DMX_HD M3<T> mul(const M3<T> &B, const M3<T> &C)
{
    M3<T> A;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            A.m[i][j] = fma_(B.m[i][2], C.m[2][j], fma_(B.m[i][1], C.m[1][j], B.m[i][0] * C.m[0][j]));
    return A;
}

// rotation matrix of a unit quaternion
template <class T> DMX_HD M3<T> quat_to_R(const Q4<T> &q)
{
    T qq1 = 2 * q.x * q.x, qq2 = 2 * q.y * q.y, qq3 = 2 * q.z * q.z;
    M3<T> R;
    R.m[0][0] = 1 - qq2 - qq3;
    R.m[0][1] = 2 * fma_(q.x, q.y, -(q.w * q.z));
    R.m[0][2] = 2 * fma_(q.x, q.z, q.w * q.y);
    R.m[1][0] = 2 * fma_(q.x, q.y, q.w * q.z);
    R.m[1][1] = 1 - qq1 - qq3;
    R.m[1][2] = 2 * fma_(q.y, q.z, -(q.w * q.x));
    R.m[2][0] = 2 * fma_(q.x, q.z, -(q.w * q.y));
    R.m[2][1] = 2 * fma_(q.y, q.z, q.w * q.x);
    R.m[2][2] = 1 - qq1 - qq2;
    return R;
}

// quaternion of a rotation matrix (branch on trace / largest diagonal)
template <class T> DMX_HD Q4<T> R_to_quat(const M3<T> &R)
{
    Q4<T> q;
    T tr = R.m[0][0] + R.m[1][1] + R.m[2][2], s;
    if (tr >= 0) {
        s = tsqrt<T>(tr + 1);
        q.w = T(0.5) * s;
        s = T(0.5) * (T(1) / s);
        q.x = (R.m[2][1] - R.m[1][2]) * s;
        q.y = (R.m[0][2] - R.m[2][0]) * s;
        q.z = (R.m[1][0] - R.m[0][1]) * s;
    } else if (R.m[1][1] > R.m[0][0] && !(R.m[2][2] > R.m[1][1])) {
        s = tsqrt<T>((R.m[1][1] - (R.m[2][2] + R.m[0][0])) + 1);
        q.y = T(0.5) * s;
        s = T(0.5) * (T(1) / s);
        q.z = (R.m[1][2] + R.m[2][1]) * s;
        q.x = (R.m[0][1] + R.m[1][0]) * s;
        q.w = (R.m[0][2] - R.m[2][0]) * s;
    } else if (R.m[2][2] > R.m[0][0] && R.m[2][2] > R.m[1][1]) {
        s = tsqrt<T>((R.m[2][2] - (R.m[0][0] + R.m[1][1])) + 1);
        q.z = T(0.5) * s;
        s = T(0.5) * (T(1) / s);
        q.x = (R.m[2][0] + R.m[0][2]) * s;
        q.y = (R.m[1][2] + R.m[2][1]) * s;
        q.w = (R.m[1][0] - R.m[0][1]) * s;
    } else {
        s = tsqrt<T>((R.m[0][0] - (R.m[1][1] + R.m[2][2])) + 1);
        q.x = T(0.5) * s;
        s = T(0.5) * (T(1) / s);
        q.y = (R.m[0][1] + R.m[1][0]) * s;
        q.z = (R.m[2][0] + R.m[0][2]) * s;
        q.w = (R.m[2][1] - R.m[1][2]) * s;
    }
    return q;
}

template <class T> DMX_HD void normalize(Q4<T> &q)
{
    T l = fma_(q.z, q.z, fma_(q.y, q.y, fma_(q.x, q.x, q.w * q.w)));
    if (l > 0) {
        l = T(1) / tsqrt<T>(l);
        q.w *= l; q.x *= l; q.y *= l; q.z *= l;
    } else {
        q.w = 1; q.x = 0; q.y = 0; q.z = 0;
    }
}

// R diag(d) R^T, evaluated as R * (diag(d) * R^T)
template <class T> DMX_HD M3<T> rotate_diag(const M3<T> &R, const V3<T> &d)
{
    M3<T> t;
    const T dd[3] = { d.x, d.y, d.z };
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t.m[i][j] = dd[i] * R.m[j][i];
    return mul(R, t);
}

template <class T> DMX_HD T det3(const M3<T> &a)
{
    const T m0 = fma_(a.m[1][1], a.m[2][2], -(a.m[2][1] * a.m[1][2]));
    const T m1 = fma_(a.m[1][0], a.m[2][2], -(a.m[2][0] * a.m[1][2]));
    const T m2 = fma_(a.m[1][0], a.m[2][1], -(a.m[2][0] * a.m[1][1]));
    return fma_(a.m[0][2], m2, fma_(-a.m[0][1], m1, a.m[0][0] * m0));
}

// closed-form inverse (adjugate / determinant); false when singular
template <class T> DMX_HD bool invert3(M3<T> &d, const M3<T> &a)
{
    T det = det3(a);
    if (det == 0) return false;
    T r = T(1) / det;
    d.m[0][0] = fma_(a.m[1][1], a.m[2][2], -(a.m[1][2] * a.m[2][1])) * r;
    d.m[0][1] = fma_(a.m[2][1], a.m[0][2], -(a.m[0][1] * a.m[2][2])) * r;
    d.m[0][2] = fma_(a.m[0][1], a.m[1][2], -(a.m[1][1] * a.m[0][2])) * r;
    d.m[1][0] = fma_(a.m[1][2], a.m[2][0], -(a.m[1][0] * a.m[2][2])) * r;
    d.m[1][1] = fma_(a.m[0][0], a.m[2][2], -(a.m[2][0] * a.m[0][2])) * r;
    d.m[1][2] = fma_(a.m[1][0], a.m[0][2], -(a.m[0][0] * a.m[1][2])) * r;
    d.m[2][0] = fma_(a.m[1][0], a.m[2][1], -(a.m[2][0] * a.m[1][1])) * r;
    d.m[2][1] = fma_(a.m[2][0], a.m[0][1], -(a.m[0][0] * a.m[2][1])) * r;
    d.m[2][2] = fma_(a.m[0][0], a.m[1][1], -(a.m[0][1] * a.m[1][0])) * r;
    return true;
}

// Adds the gyroscopic torque for angular velocity w, world inertia Iw, step h to tacc.
//   explicit:  tacc -= w x (Iw w)
//   implicit (Lacoursiere 2006): Itild = Iw - h [L]x, tacc += (Iw Itild^-1 - 1) L / h, L = Iw w
// An isotropic inertia tensor has no gyroscopic torque (w x (I w) = I (w x w) = 0): the callers skip add_gyro_torque for it -- the
// reference's every body (m = 1, I = identity: AddBody leaves ODE's default mass, main.c:695-733), for which the formulas below
// would only add the rounding of R I R^T and of the 3 x 3 solve.  The oracle has the same early-out.
template <class T> DMX_HD bool isotropic(const V3<T> &Ib) { return Ib.x == Ib.y && Ib.y == Ib.z; }
template <class T> DMX_HD void add_gyro_torque(V3<T> &tacc, const M3<T> &Iw, const V3<T> &w, T h, int mode)
{
    V3<T> L = mulv(Iw, w);
    if (mode == 1) {
        V3<T> c = cross(w, L);
        tacc.x -= c.x; tacc.y -= c.y; tacc.z -= c.z;
        return;
    }
    M3<T> It;
    It.m[0][0] = 0;    It.m[0][1] = L.z;  It.m[0][2] = -L.y;
    It.m[1][0] = -L.z; It.m[1][1] = 0;    It.m[1][2] = L.x;
    It.m[2][0] = L.y;  It.m[2][1] = -L.x; It.m[2][2] = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) It.m[i][j] = fma_(It.m[i][j], h, Iw.m[i][j]);
    T hinv = T(1) / h;
    L.x *= hinv; L.y *= hinv; L.z *= hinv;
    M3<T> inv;
    if (!invert3(inv, It)) return;
    M3<T> P = mul(Iw, inv);
    P.m[0][0] -= 1; P.m[1][1] -= 1; P.m[2][2] -= 1;
    V3<T> tau = mulv(P, L);
    tacc.x += tau.x; tacc.y += tau.y; tacc.z += tau.z;
}

// q += h * 1/2 (0,w) (x) q ; renormalise
template <class T> DMX_HD void integrate_quat(Q4<T> &q, const V3<T> &w, T h)
{
    T d0 = T(0.5) * fma_(-w.z, q.z, fma_(-w.y, q.y, -w.x * q.x));
    T d1 = T(0.5) * fma_(-w.z, q.y, fma_( w.y, q.z,  w.x * q.w));
    T d2 = T(0.5) * fma_( w.z, q.x, fma_( w.y, q.w, -w.x * q.z));
    T d3 = T(0.5) * fma_( w.z, q.w, fma_(-w.y, q.x,  w.x * q.y));
    q.w = fma_(h, d0, q.w); q.x = fma_(h, d1, q.x); q.y = fma_(h, d2, q.y); q.z = fma_(h, d3, q.z);
    normalize(q);
}

// two unit vectors orthogonal to unit n and to each other
template <class T> DMX_HD void plane_space(const V3<T> &n, V3<T> &p, V3<T> &q)
{
    const T sqrt1_2 = T(0.70710678118654752440);
    if (tabs(n.z) > sqrt1_2) {
        T a = n.y * n.y + n.z * n.z;
        T k = T(1) / tsqrt<T>(a);
        p.x = 0; p.y = -n.z * k; p.z = n.y * k;
        q.x = a * k; q.y = -n.x * p.z; q.z = n.x * p.y;
    } else {
        T a = n.x * n.x + n.y * n.y;
        T k = T(1) / tsqrt<T>(a);
        p.x = -n.y * k; p.y = n.x * k; p.z = 0;
        q.x = -n.z * p.y; q.y = n.z * p.x; q.z = a * k;
    }
}

}  // namespace dmx
