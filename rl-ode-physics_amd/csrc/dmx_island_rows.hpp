// dmx_island_rows.hpp -- the per-body / per-contact / per-row phase functions of the general island step (gravity and
// world-frame inertia, the rows of a contact joint, rhs and M^-1 J^T, one SOR row update, the integration of a body).
// Shared by the SOR kernels (dmx_islands.hip) and the exact solve of dWorldStep (dmx_lcp.hip): both steppers build the same
// rows [ODE-recall: dxStepIsland / dxQuickStepIsland share getInfo1/getInfo2], /root/reference/src/main.c:213.
#pragma once
#include <hip/hip_runtime.h>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"

namespace dmx {

// per-row scratch layout (reals)
// (29 fields in ISLAND_ROW_REALS = 32 reals: a row is one aligned 128-byte line in f32 (two in f64) and solve_island_wg fetches it as
//  16-byte pieces -- eight requests that touch one line instead of 29 that touch two: a large island's sweeps are bound by how many
//  cache-line look-ups its lanes' scattered rows cost the compute unit's one L1, see row_load)
enum : int { RW_J = 0, RW_IMJ = 12, RW_RHS = 24, RW_AD = 25, RW_LO = 26, RW_HI = 27, RW_LAM = 28, RW_COUNT = ISLAND_ROW_REALS };
static_assert(RW_COUNT >= 29 && RW_COUNT % 4 == 0, "a row is read in 16-byte pieces");
// per island-body scratch layout (reals)
enum : int { BW_INVI = 0, BW_FACC = 9, BW_TACC = 12, BW_INVM = 15, BW_FC = 16, BW_TMP = 22, BW_COUNT = 28 };

template <class T> __device__ __forceinline__ V3<T> ld3(const T *p) { return { p[0], p[1], p[2] }; }
template <class T> __device__ __forceinline__ void st3(T *p, const V3<T> &v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
template <class T> __device__ __forceinline__ T dot3p(const T *a, const V3<T> &b) { return fma_(a[2], b.z, fma_(a[1], b.y, a[0] * b.x)); }
template <class T> __device__ __forceinline__ V3<T> ldS(const T *S, int64_t stride, int c0, int s)
{
    return { S[slab_ix(c0 + 0, s)], S[slab_ix(c0 + 1, s)], S[slab_ix(c0 + 2, s)] };
}

// ---- stage 0, body k of the island: gravity, world-frame inverse inertia, gyroscopic torque ---------------
template <class T>
__device__ __forceinline__ void stage_body(const T *S, const uint8_t *bflags, int64_t stride, const IslandSet<T> &I,
                                           const StepParams<T> &P, T *b, int s, int k)
{
    I.local[s] = k;
    const uint8_t fl = bflags[s];
    const Q4<T> q = { S[slab_ix(C_QUAT + 0, s)], S[slab_ix(C_QUAT + 1, s)],
                      S[slab_ix(C_QUAT + 2, s)], S[slab_ix(C_QUAT + 3, s)] };
    const V3<T> w = ldS(S, stride, C_AVEL, s);
    const T mass = S[slab_ix(C_MASS, s)];
    const V3<T> Ib = ldS(S, stride, C_INERTIA, s);
    V3<T> facc = ldS(S, stride, C_FORCE, s), tacc = ldS(S, stride, C_TORQUE, s);
    const bool kin = fl & BF_KINEMATIC;
    if (!kin && !(fl & BF_NOGRAVITY)) { facc.x = fma_(mass, P.g.x, facc.x); facc.y = fma_(mass, P.g.y, facc.y); facc.z = fma_(mass, P.g.z, facc.z); }
    M3<T> invIw;
    if (kin) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) invIw.m[i][j] = T(0);
        b[BW_INVM] = T(0);
    } else {
        const M3<T> R = quat_to_R(q);
        const V3<T> invIb = { T(1) / Ib.x, T(1) / Ib.y, T(1) / Ib.z };
        invIw = rotate_diag(R, invIb);
        if (P.gyro != 0 && !(fl & BF_NOGYRO) && !isotropic(Ib)) {
            const M3<T> Iw = rotate_diag(R, Ib);
            add_gyro_torque(tacc, Iw, w, P.h, P.gyro);
        }
        b[BW_INVM] = T(1) / mass;
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) b[BW_INVI + 3 * i + j] = invIw.m[i][j];
    st3(b + BW_FACC, facc);
    st3(b + BW_TACC, tacc);
    for (int j = 0; j < 6; j++) b[BW_FC + j] = T(0);
}

// per-contact surface arrays are optional: without them every contact carries the batch's surface (StepParams)
template <class T> __device__ __forceinline__ int contact_rpc(const IslandSet<T> &I, const StepParams<T> &P, int ci)
{
    return (I.cmu != nullptr ? I.cmu[ci] : P.mu) > 0 ? 3 : 1;
}

// ---- rows of contact ci (normal + 2 friction when mu > 0), written at island-relative row m -----------------
// RPCK = 3: the caller knows the contact has friction rows (constant trip counts: a caller that hands in thread-local
// arrays gets them in registers)
template <class T, int RPCK = 0>
__device__ __forceinline__ void contact_rows(const T *S, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P,
                                             T *rows, int *jb, int ci, int m, T hinv)
{
    const int s1 = I.cb1[ci], s2 = I.cb2[ci];
    const int l1 = I.local[s1], l2 = s2 >= 0 ? I.local[s2] : -1;
    const bool ind = I.csrc != nullptr;
    const size_t gi = ind ? (size_t)I.csrc[ci] : (size_t)ci;
    const V3<T> normal = ld3((ind ? I.gnormal : I.cnormal) + 3 * gi);
    const V3<T> cpos = ld3((ind ? I.gpos : I.cpos) + 3 * gi);
    const V3<T> x1 = ldS(S, stride, C_POS, s1);
    const V3<T> c1 = { cpos.x - x1.x, cpos.y - x1.y, cpos.z - x1.z };
    V3<T> c2 = { T(0), T(0), T(0) };
    if (s2 >= 0) {
        const V3<T> x2 = ldS(S, stride, C_POS, s2);
        c2 = { cpos.x - x2.x, cpos.y - x2.y, cpos.z - x2.z };
    }
    const bool own_surface = I.cmu != nullptr;
    const int mode = own_surface ? I.cmode[ci] : P.surf_mode;
    T mu = own_surface ? I.cmu[ci] : P.mu;
    if (mu < 0) mu = 0;
    const int rpc = RPCK ? RPCK : (mu > 0 ? 3 : 1);
    V3<T> dir[3];
    dir[0] = normal;
    if (rpc == 3) plane_space(normal, dir[1], dir[2]);
#pragma unroll
    for (int dnum = 0; dnum < rpc; dnum++) {
        T *row = rows + (size_t)(m + dnum) * RW_COUNT;
        jb[2 * (m + dnum)] = l1; jb[2 * (m + dnum) + 1] = l2;
        T *J = row + RW_J;
        st3(J, dir[dnum]);
        st3(J + 3, cross(c1, dir[dnum]));
        if (s2 >= 0) {
            J[6] = -dir[dnum].x; J[7] = -dir[dnum].y; J[8] = -dir[dnum].z;
            const V3<T> a = cross(c2, dir[dnum]);
            J[9] = -a.x; J[10] = -a.y; J[11] = -a.z;
        } else {
            for (int j = 6; j < 12; j++) J[j] = T(0);
        }
        T cval = T(0), cfm = P.cfm;
        if (dnum == 0) {
            T erp = P.erp;
            if (mode & SURF_SOFT_ERP) erp = own_surface ? I.csoft_erp[ci] : T(0);
            if (mode & SURF_SOFT_CFM) cfm = own_surface ? I.csoft_cfm[ci] : T(0);
            T depth = ind ? I.gdepth[gi] : I.cdepth[ci];
            if (depth < 0) depth = 0;
            cval = (hinv * erp) * depth;
            if (mode & SURF_BOUNCE) {
                T outgoing = dot3p(J, ldS(S, stride, C_LVEL, s1)) + dot3p(J + 3, ldS(S, stride, C_AVEL, s1));
                if (s2 >= 0) outgoing += dot3p(J + 6, ldS(S, stride, C_LVEL, s2)) + dot3p(J + 9, ldS(S, stride, C_AVEL, s2));
                const T bv = own_surface ? I.cbounce_vel[ci] : P.bounce_vel;
                if (bv >= 0 && (-outgoing) > bv) {
                    const T newc = -(own_surface ? I.cbounce[ci] : P.bounce) * outgoing;
                    if (newc > cval) cval = newc;
                }
            }
            row[RW_LO] = T(0); row[RW_HI] = Limits<T>::inf();
        } else {
            row[RW_LO] = -mu; row[RW_HI] = mu;
        }
        row[RW_RHS] = cval;     // c for now
        row[RW_AD] = cfm;       // cfm for now
        row[RW_LAM] = T(0);
    }
}

// ---- v/h + M^-1 f of body k ------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void body_tmp(const T *S, int64_t stride, T *b, int s, T hinv)
{
    const T im = b[BW_INVM];
    const V3<T> v = ldS(S, stride, C_LVEL, s), w = ldS(S, stride, C_AVEL, s);
    b[BW_TMP + 0] = fma_(b[BW_FACC + 0], im, v.x * hinv);
    b[BW_TMP + 1] = fma_(b[BW_FACC + 1], im, v.y * hinv);
    b[BW_TMP + 2] = fma_(b[BW_FACC + 2], im, v.z * hinv);
    const V3<T> tacc = ld3(b + BW_TACC);
    b[BW_TMP + 3] = dot3p(b + BW_INVI + 0, tacc);
    b[BW_TMP + 4] = dot3p(b + BW_INVI + 3, tacc);
    b[BW_TMP + 5] = dot3p(b + BW_INVI + 6, tacc);
    b[BW_TMP + 3] = fma_(w.x, hinv, b[BW_TMP + 3]); b[BW_TMP + 4] = fma_(w.y, hinv, b[BW_TMP + 4]);
    b[BW_TMP + 5] = fma_(w.z, hinv, b[BW_TMP + 5]);
}

// ---- row i: rhs = c/h - J (v/h + M^-1 f); cfm /= h; iMJ = M^-1 J^T; Ad = w/(J iMJ + cfm); J *= Ad; rhs *= Ad; Ad *= cfm
// SOR = false (the exact solve of dWorldStep): stop after iMJ -- J and rhs stay unscaled, row[RW_AD] = cfm / h
template <class T, bool SOR = true>
__device__ __forceinline__ void row_setup(T *rows, const int *jb, const T *bs, int i, T hinv, T sor_w)
{
    T *row = rows + (size_t)i * RW_COUNT;
    T *J = row + RW_J, *iMJ = row + RW_IMJ;
    const int l1 = jb[2 * i], l2 = jb[2 * i + 1];
    T sum = T(0);
    const T *in = bs + (size_t)l1 * BW_COUNT + BW_TMP;
    for (int j = 0; j < 6; j++) sum = fma_(J[j], in[j], sum);
    if (l2 >= 0) {
        in = bs + (size_t)l2 * BW_COUNT + BW_TMP;
        for (int j = 0; j < 6; j++) sum = fma_(J[6 + j], in[j], sum);
    }
    row[RW_RHS] = fma_(row[RW_RHS], hinv, -sum);
    row[RW_AD] *= hinv;

    const T *b1 = bs + (size_t)l1 * BW_COUNT;
    for (int j = 0; j < 3; j++) iMJ[j] = b1[BW_INVM] * J[j];
    const V3<T> ja1 = ld3(J + 3);
    iMJ[3] = dot3p(b1 + BW_INVI + 0, ja1); iMJ[4] = dot3p(b1 + BW_INVI + 3, ja1); iMJ[5] = dot3p(b1 + BW_INVI + 6, ja1);
    if (l2 >= 0) {
        const T *b2 = bs + (size_t)l2 * BW_COUNT;
        for (int j = 0; j < 3; j++) iMJ[6 + j] = b2[BW_INVM] * J[6 + j];
        const V3<T> ja2 = ld3(J + 9);
        iMJ[9] = dot3p(b2 + BW_INVI + 0, ja2); iMJ[10] = dot3p(b2 + BW_INVI + 3, ja2); iMJ[11] = dot3p(b2 + BW_INVI + 6, ja2);
    } else {
        for (int j = 6; j < 12; j++) iMJ[j] = T(0);
    }
    if (!SOR) return;
    T s2 = T(0);
    for (int j = 0; j < 6; j++) s2 = fma_(iMJ[j], J[j], s2);
    if (l2 >= 0) for (int j = 6; j < 12; j++) s2 = fma_(iMJ[j], J[j], s2);
    const T cfm = row[RW_AD];
    const T ad = sor_w / (s2 + cfm);
    for (int j = 0; j < 12; j++) J[j] *= ad;
    row[RW_RHS] *= ad;
    row[RW_AD] = ad * cfm;
}

// ---- one SOR row update; returns |delta lambda| -----------------------------------------------------------------
template <class T>
__device__ __forceinline__ T row_sor(T *rows, const int *jb, T *bs, int i)
{
    T *row = rows + (size_t)i * RW_COUNT;
    const T *J = row + RW_J, *iMJ = row + RW_IMJ;
    const int l1 = jb[2 * i], l2 = jb[2 * i + 1];
    T *fc1 = bs + (size_t)l1 * BW_COUNT + BW_FC;
    T *fc2 = l2 >= 0 ? bs + (size_t)l2 * BW_COUNT + BW_FC : nullptr;
    const T old = row[RW_LAM];
    T delta = fma_(-old, row[RW_AD], row[RW_RHS]);
    delta -= fma_(fc1[5], J[5], fma_(fc1[4], J[4], fma_(fc1[3], J[3], fma_(fc1[2], J[2], fma_(fc1[1], J[1], fc1[0] * J[0])))));
    if (fc2)
        delta -= fma_(fc2[5], J[11], fma_(fc2[4], J[10], fma_(fc2[3], J[9], fma_(fc2[2], J[8], fma_(fc2[1], J[7], fc2[0] * J[6])))));
    const T lo = row[RW_LO], hi = row[RW_HI];
    const T nl = old + delta;
    if (nl < lo) { delta = lo - old; row[RW_LAM] = lo; }
    else if (nl > hi) { delta = hi - old; row[RW_LAM] = hi; }
    else row[RW_LAM] = nl;
    for (int j = 0; j < 6; j++) fc1[j] = fma_(delta, iMJ[j], fc1[j]);
    if (fc2) for (int j = 0; j < 6; j++) fc2[j] = fma_(delta, iMJ[6 + j], fc2[j]);
    return tabs(delta);
}

// ---- the same row update with the row in registers and the bodies' constraint-force accumulators in LDS
//      (solve_island_wg): identical arithmetic, identical bits ---------------------------------------------------
// workgroup barrier that orders LDS traffic only (see solve_island_wg's level loop)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "lds_barrier() spells gfx9's s_waitcnt lgkmcnt(0) + s_barrier: this library is written for gfx950 (csrc/Makefile ARCH)"
#endif
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class T> struct RowRegs { T J[12], iMJ[12], rhs, ad, lo, hi, lam; int l1, l2, row; };

template <class T>
__device__ __forceinline__ void row_load(const T *rows, const int *jb, int i, RowRegs<T> &r)
{
    // the whole row as 16-byte pieces (rows are aligned to their own size: ISLAND_ROW_REALS reals from an aligned base)
    constexpr int PER = 16 / (int)sizeof(T), NP = RW_COUNT / PER;
    struct alignas(16) Piece { T v[PER]; };
    const Piece *row = reinterpret_cast<const Piece *>(rows + (size_t)i * RW_COUNT);
    T f[RW_COUNT];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const Piece q = row[p];
#pragma unroll
        for (int e = 0; e < PER; e++) f[p * PER + e] = q.v[e];
    }
#pragma unroll
    for (int j = 0; j < 12; j++) { r.J[j] = f[RW_J + j]; r.iMJ[j] = f[RW_IMJ + j]; }
    r.rhs = f[RW_RHS]; r.ad = f[RW_AD]; r.lo = f[RW_LO]; r.hi = f[RW_HI]; r.lam = f[RW_LAM];
    const int2 b = *reinterpret_cast<const int2 *>(jb + 2 * (size_t)i);
    r.l1 = b.x; r.l2 = b.y;
    r.row = i;
}

template <class T, bool STORE_LAM = true>
__device__ __forceinline__ T row_sor_lds(T *rows, RowRegs<T> &r, T *fc, bool eager = true)
{
    // both bodies' accumulators are fetched together, up front, and written back from registers: the row's two bodies differ,
    // so nothing needs re-reading in between -- one LDS round trip per row instead of three.  eager: a body-less second slot
    // fetches too (the first body's values, never used), which keeps the two fetches in one straight line of code; launches
    // of thousands of islands are bound by issue slots, not by latency, and fetch only what they use.
    T *fc1 = fc + 6 * r.l1;
    T *fc2 = fc + 6 * (r.l2 >= 0 ? r.l2 : r.l1);
    const bool two = r.l2 >= 0;
    T a[6], b[6];
#pragma unroll
    for (int j = 0; j < 6; j++) a[j] = fc1[j];
    if (two || eager) {
#pragma unroll
        for (int j = 0; j < 6; j++) b[j] = fc2[j];
    } else {
#pragma unroll
        for (int j = 0; j < 6; j++) b[j] = T(0);
    }
    const T *J = r.J;
    const T old = r.lam;
    T delta = fma_(-old, r.ad, r.rhs);
    delta -= fma_(a[5], J[5], fma_(a[4], J[4], fma_(a[3], J[3], fma_(a[2], J[2], fma_(a[1], J[1], a[0] * J[0])))));
    if (two)
        delta -= fma_(b[5], J[11], fma_(b[4], J[10], fma_(b[3], J[9], fma_(b[2], J[8], fma_(b[1], J[7], b[0] * J[6])))));
    const T nl = old + delta;
    if (nl < r.lo) { delta = r.lo - old; r.lam = r.lo; }
    else if (nl > r.hi) { delta = r.hi - old; r.lam = r.hi; }
    else r.lam = nl;
    if (STORE_LAM) rows[(size_t)r.row * RW_COUNT + RW_LAM] = r.lam;
#pragma unroll
    for (int j = 0; j < 6; j++) fc1[j] = fma_(delta, r.iMJ[j], a[j]);
    if (two) {
#pragma unroll
        for (int j = 0; j < 6; j++) fc2[j] = fma_(delta, r.iMJ[6 + j], b[j]);
    }
    return tabs(delta);
}

// ---- body k: v += h cforce ; v += h M^-1 f ; integrate ; clear accumulators -------------------------------------
template <class T>
__device__ __forceinline__ void finish_body(T *S, const uint8_t *bflags, int64_t stride, const T *b, int s, bool has_rows, T h)
{
    V3<T> x = ldS(S, stride, C_POS, s);
    Q4<T> q = { S[slab_ix(C_QUAT + 0, s)], S[slab_ix(C_QUAT + 1, s)],
                S[slab_ix(C_QUAT + 2, s)], S[slab_ix(C_QUAT + 3, s)] };
    V3<T> v = ldS(S, stride, C_LVEL, s), w = ldS(S, stride, C_AVEL, s);
    if (has_rows) {
        v.x = fma_(h, b[BW_FC + 0], v.x); v.y = fma_(h, b[BW_FC + 1], v.y); v.z = fma_(h, b[BW_FC + 2], v.z);
        w.x = fma_(h, b[BW_FC + 3], w.x); w.y = fma_(h, b[BW_FC + 4], w.y); w.z = fma_(h, b[BW_FC + 5], w.z);
    }
    if (!(bflags[s] & BF_KINEMATIC)) {
        const T hm = h * b[BW_INVM];
        v.x = fma_(hm, b[BW_FACC + 0], v.x); v.y = fma_(hm, b[BW_FACC + 1], v.y); v.z = fma_(hm, b[BW_FACC + 2], v.z);
        V3<T> tacc = ld3(b + BW_TACC);
        tacc.x *= h; tacc.y *= h; tacc.z *= h;
        w.x += dot3p(b + BW_INVI + 0, tacc); w.y += dot3p(b + BW_INVI + 3, tacc); w.z += dot3p(b + BW_INVI + 6, tacc);
    }
    x.x = fma_(h, v.x, x.x); x.y = fma_(h, v.y, x.y); x.z = fma_(h, v.z, x.z);
    integrate_quat(q, w, h);
    S[slab_ix(C_POS + 0, s)] = x.x; S[slab_ix(C_POS + 1, s)] = x.y; S[slab_ix(C_POS + 2, s)] = x.z;
    S[slab_ix(C_QUAT + 0, s)] = q.w; S[slab_ix(C_QUAT + 1, s)] = q.x;
    S[slab_ix(C_QUAT + 2, s)] = q.y; S[slab_ix(C_QUAT + 3, s)] = q.z;
    S[slab_ix(C_LVEL + 0, s)] = v.x; S[slab_ix(C_LVEL + 1, s)] = v.y; S[slab_ix(C_LVEL + 2, s)] = v.z;
    S[slab_ix(C_AVEL + 0, s)] = w.x; S[slab_ix(C_AVEL + 1, s)] = w.y; S[slab_ix(C_AVEL + 2, s)] = w.z;
    for (int j = 0; j < 6; j++) S[slab_ix(C_FORCE + j, s)] = T(0);
}

}  // namespace dmx
