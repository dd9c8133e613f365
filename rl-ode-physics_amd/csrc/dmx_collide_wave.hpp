// dmx_collide_wave.hpp -- the convex-hull colliders of dCollide (/root/reference/src/main.c:678), one WAVEFRONT per
// geom pair: lane l tests hull point / face 64 j + l, ballots give every hit its rank in array order, so the contacts kept
// are the ones a sequential walk keeps (the CPU restatement the tests compare with).  Shared by the exact tick's narrowphase (dmx_exact.hip)
// and the fused path of bodies at static geometry (dmx_narrow.hip).  The caller says where a contact goes:
// emit(rank, pos, normal, depth) runs on the one lane that holds contact `rank` of this geom pair.  The hull's points (3 reals a
// point) are read through `pts`: StepParams::hull itself, or a workgroup's copy of it in LDS.
#pragma once

#include <hip/hip_runtime.h>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"

namespace dmx {

template <class T> struct RealEps;
template <> struct RealEps<float>  { static __device__ __forceinline__ float  v() { return 1.1920929e-7f; } };
template <> struct RealEps<double> { static __device__ __forceinline__ double v() { return 2.220446049250313e-16; } };

// The filters below compute a point's coordinate q = c . (R p + x - x') as (R^T c) . p + c . (x - x') and compare it with a bound;
// the exact tests compute it the first way.  With u = eps / 2 the unit roundoff, r >= |p|, |c|_2 = 1 (so |c|_1 <= sqrt 3):
//   exact:  R p three fused steps (3 sqrt3 u r), + x (u (|x| + r)), - x' (u |d|), c . d three fused steps (3 sqrt3 u |d|), the
//           errors of d carried through c (x sqrt 3):                                      <= 1.8 u |x| + 11 u r + 7 u |d|
//   filter: R^T c (3 sqrt3 u a component, times |p|: 9 u r), the dot with p and the offset (4 u (r + |off|)), the offset itself
//           (7 u |x - x'|):                                                                <= 13 u r + 14 u |x - x'|
// with |d| <= |x - x'| + r the two differ by less than 1.8 u |x| + 31 u r + 21 u |x - x'| <= 32 u (|x|_1 + |x'|_1 + r): the slack
// is 16 eps of that sum (+ 1, + the bound itself for the rounding of the comparison's right-hand side).
template <class T> __device__ __forceinline__ T hull_filter_slack();
template <> __device__ __forceinline__ float  hull_filter_slack<float>()  { return 16.0f * 1.1920929e-7f; }
template <> __device__ __forceinline__ double hull_filter_slack<double>() { return 16.0 * 2.220446049250313e-16; }

// ---- convex hull against the ground plane (dCollideConvexPlane [ODE-recall]) ----------------------------------------
// ODE walks the hull's points in array order: a point on or below the plane becomes a contact (position = the point,
// depth = distance below) until max_contacts are taken, and the result counts only if the hull has points on both
// sides of the plane (or on it).  ODE's early exit (max_contacts reached and both signs seen) only skips points that can
// change neither the contact set nor the both-sides test, so the wave may stop at the same condition.
template <class T, class Emit>
__device__ __forceinline__ int wave_convex_plane(const V3<T> &x, const M3<T> &R, T hull_radius, const StepParams<T> &P, int maxc, int lane, Emit emit, const T *pts)
{
    int contacts = 0;
    bool any_le = false, any_ge = false;
    // A pass of 64 points is all arithmetic (the wavefront issues ~50 instructions for it) and nearly every point is far above the
    // plane.  So first the point's height by ONE composite dot product in the hull's frame, (R^T n) . p + (n . x - d): rounded
    // differently from the walk's own n . (R p + x) - d, but within `slack` of it; a point more than `slack` above the plane is
    // above it for the walk too (it sets any_ge and nothing else), and a pass with no other point is over.  The rest are evaluated
    // exactly as before, in array order: same contacts, same bits.
    const V3<T> u = { fma_(R.m[2][0], P.pn.z, fma_(R.m[1][0], P.pn.y, R.m[0][0] * P.pn.x)),
                      fma_(R.m[2][1], P.pn.z, fma_(R.m[1][1], P.pn.y, R.m[0][1] * P.pn.x)),
                      fma_(R.m[2][2], P.pn.z, fma_(R.m[1][2], P.pn.y, R.m[0][2] * P.pn.x)) };
    const T off = dot(P.pn, x) - P.pd;
    const T slack = (P.hull_nofilter & 1) ? Limits<T>::inf() : hull_filter_slack<T>() * (tabs(x.x) + tabs(x.y) + tabs(x.z) + tabs(P.pd) + hull_radius + T(1));
    for (int base = 0; base < P.hull_n; base += 64) {
        const int k = base + lane, kc = k < P.hull_n ? k : P.hull_n - 1;
        bool below = false, le = false;
        V3<T> v2 = { T(0), T(0), T(0) };
        T distance2 = T(0);
        const V3<T> p = { pts[3 * kc], pts[3 * kc + 1], pts[3 * kc + 2] };       // (spare lanes of the last pass: the last point again, told apart below)
        const bool near = k < P.hull_n && !(fma_(u.z, p.z, fma_(u.y, p.y, fma_(u.x, p.x, off))) > slack);
        bool ge = k < P.hull_n && !near;
        if (__ballot(near) != 0ull) {
            if (near) {
                v2 = mulv(R, p);
                v2.x += x.x; v2.y += x.y; v2.z += x.z;
                distance2 = dot(P.pn, v2) - P.pd;
                le = distance2 <= T(0);
                ge = distance2 >= T(0);
                below = le;
            }
            const unsigned long long mb = __ballot(below);
            any_le = any_le || mb != 0ull;
            if (below) {
                const int rank = contacts + __popcll(mb & ((1ull << lane) - 1ull));
                if (rank < maxc) emit(rank, v2, P.pn, -distance2);
            }
            contacts += __popcll(mb);
        }
        any_ge = any_ge || (__ballot(ge) != 0ull);
        if (contacts >= maxc && any_le && any_ge) break;
    }
    return (any_le && any_ge) ? (contacts < maxc ? contacts : maxc) : 0;
}

// ---- box against convex hull: this library's collider (ODE's dCollideConvexBox is an empty stub; include/dmx_batch.h,
// dmxBatchSetConvexHullFaces): (1) hull vertices inside the box, in array order, each along the box face it is nearest to;
// (2) box corners inside the hull (corner order = bits), each along the hull face it is nearest to.  The normal points
// into the box; `negate` flips it (hull first in dCollide's order, or a reversed joint).
// The walk's filter for one (box, hull) pair: plain per-lane arithmetic, so a caller with several pairs at hand can set them up
// one pair a lane (np_convex_static_tile) and hand each to the wavefront in turn.
// Along the box's THINNEST axis a (a floor's, a wall's, a plank's normal): the point's coordinate on it by one composite dot product,
// (Rh^T b_a) . p + b_a . (xh - xb); a point farther out than half_a + slack is outside the box for the exact test too, and a pass
// of 64 such points is over after five instructions instead of fifty.
template <class T> struct BoxFilter { V3<T> u; T off, bound, slack; };
template <class T>
__device__ __forceinline__ BoxFilter<T> box_filter(const V3<T> &xb, const M3<T> &Rb, const T (&half)[3], const V3<T> &xh, const M3<T> &Rh, T hull_radius)
{
    const int ax = half[0] <= half[1] ? (half[0] <= half[2] ? 0 : 2) : (half[1] <= half[2] ? 1 : 2);
    const V3<T> ba = { ax == 0 ? Rb.m[0][0] : (ax == 1 ? Rb.m[0][1] : Rb.m[0][2]),
                       ax == 0 ? Rb.m[1][0] : (ax == 1 ? Rb.m[1][1] : Rb.m[1][2]),
                       ax == 0 ? Rb.m[2][0] : (ax == 1 ? Rb.m[2][1] : Rb.m[2][2]) };
    const T ha = ax == 0 ? half[0] : (ax == 1 ? half[1] : half[2]);
    BoxFilter<T> F;
    F.u = { fma_(Rh.m[2][0], ba.z, fma_(Rh.m[1][0], ba.y, Rh.m[0][0] * ba.x)),
            fma_(Rh.m[2][1], ba.z, fma_(Rh.m[1][1], ba.y, Rh.m[0][1] * ba.x)),
            fma_(Rh.m[2][2], ba.z, fma_(Rh.m[1][2], ba.y, Rh.m[0][2] * ba.x)) };
    F.off = fma_(ba.z, xh.z - xb.z, fma_(ba.y, xh.y - xb.y, ba.x * (xh.x - xb.x)));
    F.slack = hull_filter_slack<T>() * (tabs(xh.x) + tabs(xh.y) + tabs(xh.z) + tabs(xb.x) + tabs(xb.y) + tabs(xb.z) + hull_radius + half[0] + half[1] + half[2] + T(1));
    F.bound = ha + F.slack;
    return F;
}
// a corner inside the hull is inside the hull's bounding sphere (slack for rounding): most corners of a floor-sized box are nowhere
// near it and skip the walk over the hull's faces.  Per-lane arithmetic again: corner cn of the box against one hull.
template <class T>
__device__ __forceinline__ bool box_corner_near(const V3<T> &xb, const M3<T> &Rb, const T (&half)[3], const V3<T> &xh, T hull_radius, int cn)
{
    const V3<T> l = { (cn & 1) ? half[0] : -half[0], (cn & 2) ? half[1] : -half[1], (cn & 4) ? half[2] : -half[2] };
    V3<T> cw = mulv(Rb, l);
    cw.x += xb.x; cw.y += xb.y; cw.z += xb.z;
    const V3<T> d = { cw.x - xh.x, cw.y - xh.y, cw.z - xh.z };
    return !(d.x * d.x + d.y * d.y + d.z * d.z > hull_radius * hull_radius * T(1.0001));
}

// the walk proper, for a pair whose filter F and near corners (bit cn) are at hand
template <class T, class Emit>
__device__ __forceinline__ int wave_box_convex_walk(const V3<T> &xb, const M3<T> &Rb, const T (&half)[3], const V3<T> &xh, const M3<T> &Rh,
                                                    const BoxFilter<T> &F, unsigned near_corners, const StepParams<T> &P, int maxc, bool negate,
                                                    int lane, Emit emit, const T *pts, const T *box_aabb = nullptr, bool *aabbs_meet = nullptr)
{
    // box_aabb (lo[3], hi[3]; may be null): the box's world AABB as dSpaceCollide tests it.  *aabbs_meet comes back true when some
    // hull vertex that became a contact lies inside it -- the hull's exact AABB, the bounds of these very vertex positions, then
    // overlaps it for certain; false says nothing (the caller that has not made dSpaceCollide's test yet makes it then).
    // (A margin on the depth instead of the comparison with the stored box -- one comparison on values the walk has anyway -- is
    //  too weak: a resting teapot's contacts are shallower than the rounding of positions 300 m from the origin.)
    // (no flag beside it: a per-lane bool carried through the loop is a lane mask that has to be merged on every pass, three scalar
    //  instructions in a pass of twenty-one; "not yet" is a NaN in the vertex itself, which also fails every comparison at the end)
    V3<T> wv = { Limits<T>::nan(), T(0), T(0) };
    int contacts = 0;
    const V3<T> u = F.u;
    const T off = F.off, bound = F.bound;
    // (one loop condition: the walk's end moves to 0 once the contacts are full -- decided after a pass that found something)
    int limit = maxc > 0 ? P.hull_n : 0;
    for (int base = 0; base < limit; base += 64) {
        const int k = base + lane, kc = k < P.hull_n ? k : P.hull_n - 1;
        bool inside = false;
        V3<T> v = { T(0), T(0), T(0) }, n = { T(0), T(0), T(0) };
        T dep = T(0);
        // (the last pass's spare lanes fetch the last point again and are told apart afterwards: no branch around the fetch, no
        //  registers to clear for it -- six of a pass's twenty-one instructions)
        const V3<T> p = { pts[3 * kc], pts[3 * kc + 1], pts[3 * kc + 2] };
        const bool near = k < P.hull_n && !(tabs(fma_(u.z, p.z, fma_(u.y, p.y, fma_(u.x, p.x, off)))) > bound);
        if (__ballot(near) == 0ull) continue;
        if (near) {
            v = mulv(Rh, p);
            v.x += xh.x; v.y += xh.y; v.z += xh.z;
            const V3<T> d = { v.x - xb.x, v.y - xb.y, v.z - xb.z };
            T q[3];
#pragma unroll
            for (int a = 0; a < 3; a++) q[a] = fma_(Rb.m[2][a], d.z, fma_(Rb.m[1][a], d.y, Rb.m[0][a] * d.x));     // box frame
            inside = !(tabs(q[0]) > half[0] || tabs(q[1]) > half[1] || tabs(q[2]) > half[2]);
            if (inside) {        // (the face it is nearest to: only the handful of points inside the box pay for this)
                int best = 0;
                dep = half[0] - tabs(q[0]);
#pragma unroll
                for (int a = 1; a < 3; a++) { const T e = half[a] - tabs(q[a]); if (e < dep) { dep = e; best = a; } }
                const T qb = best == 0 ? q[0] : (best == 1 ? q[1] : q[2]);
                const T sg = qb < T(0) ? T(-1) : T(1);
                const V3<T> col = { best == 0 ? Rb.m[0][0] : (best == 1 ? Rb.m[0][1] : Rb.m[0][2]),
                                    best == 0 ? Rb.m[1][0] : (best == 1 ? Rb.m[1][1] : Rb.m[1][2]),
                                    best == 0 ? Rb.m[2][0] : (best == 1 ? Rb.m[2][1] : Rb.m[2][2]) };
                n = { -(sg * col.x), -(sg * col.y), -(sg * col.z) };      // into the box
            }
        }
        const unsigned long long mb = __ballot(inside);
        if (inside) {
            const int rank = contacts + __popcll(mb & ((1ull << lane) - 1ull));
            if (rank < maxc) emit(rank, v, negate ? V3<T>{ -n.x, -n.y, -n.z } : n, dep);
        }
        // (a lane's first vertex inside the box is kept for the AABB question below: selects, nothing the loop branches on)
        {
            const bool first = inside && wv.x != wv.x;
            wv.y = first ? v.y : wv.y; wv.z = first ? v.z : wv.z; wv.x = first ? v.x : wv.x;
        }
        contacts += __popcll(mb);
        limit = contacts < maxc ? limit : 0;
    }
    if (aabbs_meet != nullptr && box_aabb != nullptr)
        *aabbs_meet = __ballot(wv.x >= box_aabb[0] && wv.x <= box_aabb[3] && wv.y >= box_aabb[1] && wv.y <= box_aabb[4] && wv.z >= box_aabb[2] &&
                               wv.z <= box_aabb[5]) != 0ull;
    if (contacts > maxc) contacts = maxc;
    if (contacts >= maxc || P.hull_nf <= 0) return contacts;
    for (int cn = 0; cn < 8 && contacts < maxc; cn++) {
        if (!((near_corners >> cn) & 1u)) continue;
        const V3<T> l = { (cn & 1) ? half[0] : -half[0], (cn & 2) ? half[1] : -half[1], (cn & 4) ? half[2] : -half[2] };
        V3<T> cw = mulv(Rb, l);
        cw.x += xb.x; cw.y += xb.y; cw.z += xb.z;
        const V3<T> d = { cw.x - xh.x, cw.y - xh.y, cw.z - xh.z };
        V3<T> r;
        r.x = fma_(Rh.m[2][0], d.z, fma_(Rh.m[1][0], d.y, Rh.m[0][0] * d.x));
        r.y = fma_(Rh.m[2][1], d.z, fma_(Rh.m[1][1], d.y, Rh.m[0][1] * d.x));
        r.z = fma_(Rh.m[2][2], d.z, fma_(Rh.m[1][2], d.y, Rh.m[0][2] * d.x));
        T dep = Limits<T>::inf();
        int fbest = 0x7fffffff;
        bool neg = false;
        for (int f = lane; f < P.hull_nf; f += 64) {
            const T *pl = P.hull_planes + 4 * f;
            const T e = pl[3] - dot(V3<T>{ pl[0], pl[1], pl[2] }, r);
            if (e < T(0)) neg = true;
            if (e < dep) { dep = e; fbest = f; }             // (f ascends within a lane: the first minimum is kept)
        }
        if (__ballot(neg) != 0ull) continue;                 // outside some face
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {                   // lexicographic (depth, face) minimum over the wave
            const T od = __shfl_xor(dep, o, 64);
            const int of = __shfl_xor(fbest, o, 64);
            if (od < dep || (od == dep && of < fbest)) { dep = od; fbest = of; }
        }
        if (fbest == 0x7fffffff) continue;
        if (lane == 0) {
            const T *pl = P.hull_planes + 4 * fbest;
            const V3<T> nw = mulv(Rh, V3<T>{ pl[0], pl[1], pl[2] });     // the hull's outward normal points into the box
            emit(contacts, cw, negate ? V3<T>{ -nw.x, -nw.y, -nw.z } : nw, dep);
        }
        contacts++;
    }
    return contacts;
}

// one pair, set up here: every lane computes the same filter, lane cn asks for corner cn
template <class T, class Emit>
__device__ __forceinline__ int wave_box_convex(const V3<T> &xb, const M3<T> &Rb, const T *side, const V3<T> &xh, const M3<T> &Rh,
                                               T hull_radius, const StepParams<T> &P, int maxc, bool negate, int lane, Emit emit, const T *pts,
                                               const T *box_aabb = nullptr, bool *aabbs_meet = nullptr)
{
    const T half[3] = { T(0.5) * side[0], T(0.5) * side[1], T(0.5) * side[2] };
    BoxFilter<T> F = box_filter<T>(xb, Rb, half, xh, Rh, hull_radius);
    if (P.hull_nofilter & 1) F.bound = Limits<T>::inf();
    const unsigned near_corners = (unsigned)(__ballot(box_corner_near<T>(xb, Rb, half, xh, hull_radius, lane & 7)) & 0xffull);
    return wave_box_convex_walk<T>(xb, Rb, half, xh, Rh, F, near_corners, P, maxc, negate, lane, emit, pts, box_aabb, aabbs_meet);
}

// ---- a point of the world in a hull's frame: R^T (v - x), the oracle's to_hull_frame ------------------------------------------
template <class T> __device__ __forceinline__ V3<T> to_hull_frame(const M3<T> &Rh, const V3<T> &xh, const V3<T> &v)
{
    const V3<T> d = { v.x - xh.x, v.y - xh.y, v.z - xh.z };
    return { fma_(Rh.m[2][0], d.z, fma_(Rh.m[1][0], d.y, Rh.m[0][0] * d.x)),
             fma_(Rh.m[2][1], d.z, fma_(Rh.m[1][1], d.y, Rh.m[0][1] * d.x)),
             fma_(Rh.m[2][2], d.z, fma_(Rh.m[1][2], d.y, Rh.m[0][2] * d.x)) };
}

// ---- the hull's exact world AABB (dxConvex::computeAABB [ODE-recall]: the bounds of its transformed points), by the wave ----
// The exact tick's pair search uses it for every hull (bp_convex_aabb, dmx_broadphase.hip: one wavefront per hull per exact tick);
// the fused paths keep the bounding sphere's box (conservative, and free), which changes nothing for colliders that are exact
// geometry (vertex in box / in hull: no contact unless the true AABBs overlap).  The sphere collider below is not -- beside an
// edge it answers before the sphere arrives -- so ITS call site applies dSpaceCollide's own test on the exact AABB itself.
template <class T>
__device__ __forceinline__ void wave_hull_aabb(const V3<T> &x, const M3<T> &R, const T *hull, int hull_n, int lane, T lo[3], T hi[3])
{
    T l[3] = { Limits<T>::inf(), Limits<T>::inf(), Limits<T>::inf() }, h[3] = { -Limits<T>::inf(), -Limits<T>::inf(), -Limits<T>::inf() };
    for (int k = lane; k < hull_n; k += 64) {
        V3<T> v = mulv(R, V3<T>{ hull[3 * k], hull[3 * k + 1], hull[3 * k + 2] });
        v.x += x.x; v.y += x.y; v.z += x.z;
        l[0] = v.x < l[0] ? v.x : l[0]; h[0] = v.x > h[0] ? v.x : h[0];
        l[1] = v.y < l[1] ? v.y : l[1]; h[1] = v.y > h[1] ? v.y : h[1];
        l[2] = v.z < l[2] ? v.z : l[2]; h[2] = v.z > h[2] ? v.z : h[2];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const T ol = __shfl_xor(l[a], o, 64), oh = __shfl_xor(h[a], o, 64);
            l[a] = ol < l[a] ? ol : l[a]; h[a] = oh > h[a] ? oh : h[a];
        }
        lo[a] = l[a]; hi[a] = h[a];
    }
}

// ---- sphere (geom 1) against convex hull (geom 2): this library's collider (include/dmx_batch.h): the hull's face planes'
// largest signed distance to the sphere's centre, first face on ties; one contact along that face, normal into the sphere.
// Lane l walks faces l, l + 64, ...; a lexicographic (largest distance, lowest face) wave reduction picks the face.
template <class T, class Emit>
__device__ __forceinline__ int wave_sphere_convex(const V3<T> &cs, T radius, const V3<T> &xh, const M3<T> &Rh, const StepParams<T> &P,
                                                  bool negate, int lane, Emit emit)
{
    if (P.hull_nf <= 0) return 0;
    {   // dSpaceCollide's AABB test on the hull's exact box (see wave_hull_aabb)
        T lo[3], hi[3];
        wave_hull_aabb<T>(xh, Rh, P.hull, P.hull_n, lane, lo, hi);
        if (cs.x - radius > hi[0] || lo[0] > cs.x + radius || cs.y - radius > hi[1] || lo[1] > cs.y + radius ||
            cs.z - radius > hi[2] || lo[2] > cs.z + radius) return 0;
    }
    const V3<T> r = to_hull_frame(Rh, xh, cs);
    T smax = -Limits<T>::inf();
    int fbest = 0x7fffffff;
    for (int f = lane; f < P.hull_nf; f += 64) {
        const T *pl = P.hull_planes + 4 * f;
        const T sd = dot(V3<T>{ pl[0], pl[1], pl[2] }, r) - pl[3];
        if (sd > smax) { smax = sd; fbest = f; }              // (f ascends within a lane: the first maximum is kept)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T os = __shfl_xor(smax, o, 64);
        const int of = __shfl_xor(fbest, o, 64);
        if (os > smax || (os == smax && of < fbest)) { smax = os; fbest = of; }
    }
    if (fbest == 0x7fffffff || smax > radius) return 0;
    if (lane == 0) {
        const T *pl = P.hull_planes + 4 * fbest;
        const V3<T> nw = mulv(Rh, V3<T>{ pl[0], pl[1], pl[2] });
        const V3<T> pos = { cs.x - nw.x * radius, cs.y - nw.y * radius, cs.z - nw.z * radius };
        emit(0, pos, negate ? V3<T>{ -nw.x, -nw.y, -nw.z } : nw, radius - smax);
    }
    return 1;
}

// ---- convex hull A (geom 1) against convex hull B (geom 2), both the batch's one hull shape: (1) B's vertices inside A, in
// array order, each along A's face it is nearest to, normal into A; (2) A's vertices inside B, along B's nearest face.  Lane l
// transforms vertex 64 j + l; the few that survive the culls are then tested one after the other, each by the whole wave.
// Culling (it decides nothing, it only spares walks): a vertex inside a hull is inside that hull's bounding sphere and inside its
// world AABB (boxA / boxB: lo3 hi3, the exact AABBs the pair search already holds), both taken with slack far above rounding.  Two
// teapots that touch overlap in a sliver; nearly every chunk of 64 vertices then has no candidate and costs a dozen instructions.
// Is the point rr (hull frame) inside the hull?  The WAVE's walk over the faces: lane l tests faces l, l + 64, ... -- a point outside
// leaves at the first group of 64 faces that holds a face it is outside of; before that, at `fhint`, the face that sent the last
// candidate away (wave-uniform, updated here: neighbouring vertices tend to fail the same face).  Inside: (dep, fbest) = the
// nearest face, lowest index on ties, on every lane.
template <class T>
__device__ __forceinline__ bool wave_point_in_hull(const V3<T> &rr, const StepParams<T> &P, int lane, int &fhint, T &dep, int &fbest)
{
    if (fhint >= 0) {
        const T *ph = P.hull_planes + 4 * fhint;
        if (ph[3] - dot(V3<T>{ ph[0], ph[1], ph[2] }, rr) < T(0)) return false;
    }
    dep = Limits<T>::inf();
    fbest = 0x7fffffff;
    for (int f0 = 0; f0 < P.hull_nf; f0 += 64) {
        const int f = f0 + lane;
        bool neg = false;
        if (f < P.hull_nf) {
            const T *pl = P.hull_planes + 4 * f;
            const T e = pl[3] - dot(V3<T>{ pl[0], pl[1], pl[2] }, rr);
            if (e < T(0)) neg = true;
            else if (e < dep) { dep = e; fbest = f; }          // (f ascends within a lane: the first minimum is kept)
        }
        const unsigned long long nb = __ballot(neg);
        if (nb != 0ull) { fhint = f0 + __builtin_ctzll(nb); return false; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                         // lexicographic (depth, face) minimum over the wave
        const T od = __shfl_xor(dep, o, 64);
        const int of = __shfl_xor(fbest, o, 64);
        if (od < dep || (od == dep && of < fbest)) { dep = od; fbest = of; }
    }
    return fbest != 0x7fffffff;
}

template <class T, class Emit>
__device__ __forceinline__ int wave_convex_convex(const V3<T> &xa, const M3<T> &Ra, const V3<T> &xb, const M3<T> &Rb, T hull_radius,
                                                  const T *boxA, const T *boxB, const StepParams<T> &P, int maxc, bool negate, int lane,
                                                  Emit emit)
{
    if (P.hull_nf <= 0) return 0;
    int contacts = 0;
    const T slack = hull_radius * T(1e-4);
    for (int pass = 0; pass < 2; pass++) {
        const V3<T> &xv = pass == 0 ? xb : xa; const M3<T> &Rv = pass == 0 ? Rb : Ra;      // the hull whose vertices are walked
        const V3<T> &xh = pass == 0 ? xa : xb; const M3<T> &Rh = pass == 0 ? Ra : Rb;      // the hull they are tested against
        const T *box = pass == 0 ? boxA : boxB;
        int fhint = -1;                    // the face that sent the last candidate away (wave-uniform)
        for (int base = 0; base < P.hull_n && contacts < maxc; base += 64) {
            const int k = base + lane;
            V3<T> v = { T(0), T(0), T(0) }, r = { T(0), T(0), T(0) };
            bool alive = false;
            if (k < P.hull_n) {
                v = mulv(Rv, V3<T>{ P.hull[3 * k], P.hull[3 * k + 1], P.hull[3 * k + 2] });
                v.x += xv.x; v.y += xv.y; v.z += xv.z;
                r = to_hull_frame(Rh, xh, v);
                // inside the hull means inside its bounding sphere (slack for rounding): most vertices skip the walk
                alive = !(r.x * r.x + r.y * r.y + r.z * r.z > hull_radius * hull_radius * T(1.0001)) &&
                        !(v.x < box[0] - slack || v.x > box[3] + slack || v.y < box[1] - slack || v.y > box[4] + slack ||
                          v.z < box[2] - slack || v.z > box[5] + slack);
            }
            // the candidates of this chunk, in array order; each one's walk over the faces is the WAVE's: lane l tests faces l, l + 64,
            // ... (a vertex outside the hull leaves at the first group of 64 faces that holds a face it is outside of; before that,
            // at the face that sent the last candidate away: neighbouring vertices tend to fail the same face)
            unsigned long long cand = __ballot(alive);
            while (cand != 0ull && contacts < maxc) {
                const int l = __builtin_ctzll(cand);
                cand &= cand - 1ull;
                const V3<T> rr = { __shfl(r.x, l, 64), __shfl(r.y, l, 64), __shfl(r.z, l, 64) };
                T dep;
                int fbest;
                if (!wave_point_in_hull<T>(rr, P, lane, fhint, dep, fbest)) continue;
                if (lane == l) {
                    const T *pl = P.hull_planes + 4 * fbest;
                    const V3<T> nw = mulv(Rh, V3<T>{ pl[0], pl[1], pl[2] });
                    const bool flip = (pass == 0) != negate;           // pass 0: against A's outward normal (into A); `negate` flips all
                    emit(contacts, v, flip ? V3<T>{ -nw.x, -nw.y, -nw.z } : nw, dep);
                }
                contacts++;
            }
        }
        if (contacts > maxc) contacts = maxc;
    }
    return contacts;
}

// ---- the same collider by a WORKGROUP of NW wavefronts: one deep pair of hulls is a hundred candidate vertices, each a walk over
// 2 500 faces by a whole wavefront -- 150-180 us for one wavefront, and the duration of the narrowphase launch whatever the
// number of pairs.  Here the vertices are taken 64 NW at a time: every thread tests one against the other hull's bounding
// sphere and box, the survivors are listed in array order (ballots + a prefix over the waves), the waves take them in turn
// (wave w: candidates w, w + NW, ...), and the first maxc that are inside -- in array order, as the one-wavefront walk keeps
// them -- become contacts.  Same arithmetic per vertex, same contacts, same bits.  Called by every thread of the workgroup;
// returns the number of contacts on every thread; emit(rank, ...) runs on one thread per contact.
template <class T, int NW, class Emit>
__device__ __forceinline__ int wg_convex_convex(const V3<T> &xa, const M3<T> &Ra, const V3<T> &xb, const M3<T> &Rb, T hull_radius,
                                                const T *boxA, const T *boxB, const StepParams<T> &P, int maxc, bool negate, Emit emit)
{
    constexpr int CH = 64 * NW;
    __shared__ int s_k[CH], s_face[CH], s_rank[CH], s_wcount[NW], s_contacts;
    __shared__ T s_r[3 * CH], s_dep[CH];
    if (P.hull_nf <= 0) return 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T slack = hull_radius * T(1e-4);
    if (tid == 0) s_contacts = 0;
    __syncthreads();
    int contacts = 0;
    for (int pass = 0; pass < 2; pass++) {
        const V3<T> &xv = pass == 0 ? xb : xa; const M3<T> &Rv = pass == 0 ? Rb : Ra;      // the hull whose vertices are walked
        const V3<T> &xh = pass == 0 ? xa : xb; const M3<T> &Rh = pass == 0 ? Ra : Rb;      // the hull they are tested against
        const T *box = pass == 0 ? boxA : boxB;
        int fhint = -1;
        for (int base = 0; base < P.hull_n && contacts < maxc; base += CH) {
            const int k = base + tid;
            V3<T> r = { T(0), T(0), T(0) };
            bool alive = false;
            if (k < P.hull_n) {
                V3<T> v = mulv(Rv, V3<T>{ P.hull[3 * k], P.hull[3 * k + 1], P.hull[3 * k + 2] });
                v.x += xv.x; v.y += xv.y; v.z += xv.z;
                r = to_hull_frame(Rh, xh, v);
                alive = !(r.x * r.x + r.y * r.y + r.z * r.z > hull_radius * hull_radius * T(1.0001)) &&
                        !(v.x < box[0] - slack || v.x > box[3] + slack || v.y < box[1] - slack || v.y > box[4] + slack ||
                          v.z < box[2] - slack || v.z > box[5] + slack);
            }
            const unsigned long long mb = __ballot(alive);
            if (lane == 0) s_wcount[wave] = __popcll(mb);
            __syncthreads();
            int before = 0, ncand = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) { const int c = s_wcount[w]; if (w < wave) before += c; ncand += c; }
            if (alive) {
                const int c = before + __popcll(mb & ((1ull << lane) - 1ull));
                s_k[c] = k; s_r[3 * c] = r.x; s_r[3 * c + 1] = r.y; s_r[3 * c + 2] = r.z;
            }
            __syncthreads();
            for (int c = wave; c < ncand; c += NW) {                      // wave-uniform
                const V3<T> rr = { s_r[3 * c], s_r[3 * c + 1], s_r[3 * c + 2] };
                T dep;
                int fbest;
                const bool in = wave_point_in_hull<T>(rr, P, lane, fhint, dep, fbest);
                if (lane == 0) { s_face[c] = in ? fbest : -1; s_dep[c] = dep; }
            }
            __syncthreads();
            if (tid == 0) {
                int n = s_contacts;                                       // (the one-wavefront walk stops at maxc: so does the count)
                for (int c = 0; c < ncand; c++) {
                    const bool take = s_face[c] >= 0 && n < maxc;
                    s_rank[c] = take ? n : -1;
                    if (take) n++;
                }
                s_contacts = n;
            }
            __syncthreads();
            if (tid < ncand && s_rank[tid] >= 0) {
                const int kk = s_k[tid];
                V3<T> v = mulv(Rv, V3<T>{ P.hull[3 * kk], P.hull[3 * kk + 1], P.hull[3 * kk + 2] });
                v.x += xv.x; v.y += xv.y; v.z += xv.z;
                const T *pl = P.hull_planes + 4 * s_face[tid];
                const V3<T> nw = mulv(Rh, V3<T>{ pl[0], pl[1], pl[2] });
                const bool flip = (pass == 0) != negate;
                emit(s_rank[tid], v, flip ? V3<T>{ -nw.x, -nw.y, -nw.z } : nw, s_dep[tid]);
            }
            contacts = s_contacts;
            __syncthreads();
        }
    }
    return contacts;
}

}  // namespace dmx
