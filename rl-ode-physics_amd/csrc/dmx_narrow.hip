// dmx_narrow.hip -- device narrowphase of the exact (pair-bearing) tick: dCollide (/root/reference/src/main.c:678)
// for the bodies the broadphase found in body-body pairs.  Same __host__ __device__ colliders as the host side of
// the ODE API (dmx_collide.hpp); contact geometry stays in device memory and is consumed by solve_islands through
// an index, so the host only ever sees integer counts.
#include <hip/hip_runtime.h>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"
#include "dmx_collide.hpp"

namespace dmx {

template <class T> struct BodyGeom { V3<T> x; M3<T> R; T side[3]; int gt; };

template <class T>
__device__ __forceinline__ BodyGeom<T> load_geom(const T *S, const uint8_t *gtype, int64_t stride, int64_t i)
{
    BodyGeom<T> g;
    g.x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
    g.R = quat_to_R(Q4<T>{ S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)],
                           S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] });
    for (int a = 0; a < 3; a++) g.side[a] = S[slab_ix(C_SIDES + a, i)];
    g.gt = gtype[i];
    return g;
}

template <class T> __device__ __forceinline__ void put(T *gpos, T *gnormal, T *gdepth, int slot, const V3<T> &p,
                                                       const V3<T> &n, T d)
{
    gpos[3 * slot] = p.x; gpos[3 * slot + 1] = p.y; gpos[3 * slot + 2] = p.z;
    gnormal[3 * slot] = n.x; gnormal[3 * slot + 1] = n.y; gnormal[3 * slot + 2] = n.z;
    gdepth[slot] = d;
}

// ground-plane contacts of the listed bodies: 4 slots per body (slot = 4*k + c)
template <class T>
__global__ __launch_bounds__(64) void np_plane(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t stride,
                                               const int32_t *__restrict__ bodies, int nb, StepParams<T> P,
                                               T *__restrict__ gpos, T *__restrict__ gnormal, T *__restrict__ gdepth,
                                               int32_t *__restrict__ count)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nb) return;
    int nc = 0;
    if (P.plane_on) {
        const BodyGeom<T> g = load_geom<T>(S, gtype, stride, bodies[k]);
        V3<T> cp[4]; T cd[4];
        if (g.gt == GEOM_BOX) nc = box_plane(g.x, g.R, g.side, P.pn, P.pd, P.max_contacts, cp, cd);
        else if (g.gt == GEOM_SPHERE) nc = sphere_plane(g.x, g.side[0], P.pn, P.pd, cp, cd);
        for (int c = 0; c < nc; c++) put(gpos, gnormal, gdepth, 4 * k + c, cp[c], P.pn, cd[c]);
    }
    count[k] = nc;
}

// contacts of the listed (sorted) body pairs: 8 slots per pair (slot = base + 8*p + c); normal points into body i
template <class T>
__global__ __launch_bounds__(64) void np_pairs(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t stride,
                                               const int32_t *__restrict__ pairs, int np, int maxc, int base,
                                               T *__restrict__ gpos, T *__restrict__ gnormal, T *__restrict__ gdepth,
                                               int32_t *__restrict__ count)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    const BodyGeom<T> A = load_geom<T>(S, gtype, stride, pairs[2 * p]);
    const BodyGeom<T> B = load_geom<T>(S, gtype, stride, pairs[2 * p + 1]);
    ContactPoint<T> c[8];
    int nc = 0;
    bool flip = false;      // a collider exists only for the swapped class order: swap, then negate the normal
    const int mc = maxc > 8 ? 8 : maxc;
    if (A.gt == GEOM_BOX && B.gt == GEOM_BOX) nc = box_box(A.x, A.R, A.side, B.x, B.R, B.side, mc, c);
    else if (A.gt == GEOM_SPHERE && B.gt == GEOM_SPHERE) nc = sphere_sphere(A.x, A.side[0], B.x, B.side[0], c);
    else if (A.gt == GEOM_SPHERE && B.gt == GEOM_BOX) nc = sphere_box(A.x, A.side[0], B.x, B.R, B.side, c);
    else if (A.gt == GEOM_BOX && B.gt == GEOM_SPHERE) { nc = sphere_box(B.x, B.side[0], A.x, A.R, A.side, c); flip = true; }
    if (nc > mc) nc = mc;
    for (int k = 0; k < nc; k++) {
        const V3<T> n = flip ? V3<T>{ -c[k].normal.x, -c[k].normal.y, -c[k].normal.z } : c[k].normal;
        put(gpos, gnormal, gdepth, base + 8 * p + k, c[k].pos, n, c[k].depth);
    }
    count[p] = nc;
}

// ---- convex hull against the ground plane (dCollideConvexPlane): one wavefront per convex body ------------------
// ODE walks the hull's points in array order: a point on or below the plane becomes a contact (position = the point,
// depth = distance below) until max_contacts are taken, and the result counts only if the hull has points on both
// sides of the plane (or on it).  Taken in parallel: lane l tests point 64*j + l; a ballot gives every penetrating
// point its rank in array order (contacts so far + penetrating lanes below it); ranks < max_contacts are the contacts
// ODE would have kept.  ODE's early exit (max_contacts reached and both signs seen) only skips points that can change
// neither the contact set nor the both-sides test, so the wave may stop at the same condition.
// Output: cbuf[i][k] = (x, y, z, depth) for k < ccount[i]; bodies of other classes are left alone.
template <class T>
__global__ __launch_bounds__(256) void np_convex_plane(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t n,
                                                       StepParams<T> P)
{
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // 4 wavefronts per workgroup
    if (i >= n || gtype[i] != GEOM_CONVEX) return;                         // wave-uniform
    const int lane = threadIdx.x & 63;
    const V3<T> x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
    const M3<T> R = quat_to_R(Q4<T>{ S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)],
                                     S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] });
    const int maxc = P.max_contacts < CONVEX_MAXC ? P.max_contacts : CONVEX_MAXC;
    T *out = P.cbuf + (size_t)i * CONVEX_MAXC * 4;
    int contacts = 0;
    bool any_le = false, any_ge = false;
    for (int base = 0; base < P.hull_n; base += 64) {
        const int k = base + lane;
        bool below = false, le = false, ge = false;
        V3<T> v2 = { T(0), T(0), T(0) };
        T distance2 = T(0);
        if (k < P.hull_n) {
            const V3<T> p = { P.hull[3 * k], P.hull[3 * k + 1], P.hull[3 * k + 2] };
            v2 = mulv(R, p);
            v2.x += x.x; v2.y += x.y; v2.z += x.z;
            distance2 = dot(P.pn, v2) - P.pd;
            le = distance2 <= T(0);
            ge = distance2 >= T(0);
            below = le;
        }
        const unsigned long long mb = __ballot(below);
        any_le = any_le || (__ballot(le) != 0ull);
        any_ge = any_ge || (__ballot(ge) != 0ull);
        if (below) {
            const int rank = contacts + __popcll(mb & ((1ull << lane) - 1ull));
            if (rank < maxc) {
                out[4 * rank + 0] = v2.x; out[4 * rank + 1] = v2.y; out[4 * rank + 2] = v2.z;
                out[4 * rank + 3] = -distance2;
            }
        }
        contacts += __popcll(mb);
        if (contacts >= maxc && any_le && any_ge) break;
    }
    if (lane == 0) P.ccount[i] = (any_le && any_ge) ? (contacts < maxc ? contacts : maxc) : 0;
}

template <class T>
hipError_t launch_np_convex_plane(const T *S, const uint8_t *gtype, int64_t n, const StepParams<T> &P, hipStream_t st)
{
    if (n <= 0 || P.hull_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((np_convex_plane<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, S, gtype, n, P);
    return hipGetLastError();
}

template <class T>
hipError_t launch_np_plane(const T *S, const uint8_t *gtype, int64_t stride, const int32_t *bodies, int nb,
                           const StepParams<T> &P, T *gpos, T *gnormal, T *gdepth, int32_t *count, hipStream_t st)
{
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL((np_plane<T>), dim3((nb + 63) / 64), dim3(64), 0, st, S, gtype, stride, bodies, nb, P, gpos, gnormal, gdepth, count);
    return hipGetLastError();
}
template <class T>
hipError_t launch_np_pairs(const T *S, const uint8_t *gtype, int64_t stride, const int32_t *pairs, int np, int maxc,
                           int base_slot, T *gpos, T *gnormal, T *gdepth, int32_t *count, hipStream_t st)
{
    if (np <= 0) return hipSuccess;
    hipLaunchKernelGGL((np_pairs<T>), dim3((np + 63) / 64), dim3(64), 0, st, S, gtype, stride, pairs, np, maxc, base_slot, gpos, gnormal, gdepth, count);
    return hipGetLastError();
}

#define DMX_NP_INST(T)                                                                                                        \
    template hipError_t launch_np_plane<T>(const T *, const uint8_t *, int64_t, const int32_t *, int, const StepParams<T> &,  \
                                           T *, T *, T *, int32_t *, hipStream_t);                                            \
    template hipError_t launch_np_pairs<T>(const T *, const uint8_t *, int64_t, const int32_t *, int, int, int, T *, T *, T *, \
                                           int32_t *, hipStream_t);                                                           \
    template hipError_t launch_np_convex_plane<T>(const T *, const uint8_t *, int64_t, const StepParams<T> &, hipStream_t);
DMX_NP_INST(float)
DMX_NP_INST(double)

}  // namespace dmx
