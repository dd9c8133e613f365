// dmx_narrow.hip -- dCollide (/root/reference/src/main.c:678) for convex hulls against the ground plane on the FUSED
// path: one wavefront per hull leaves the hull's contacts where step_plane<.., 8> picks them up.  (The narrowphase of
// the exact tick -- boxes / spheres / hulls against plane, static boxes and one another -- is in dmx_exact.hip.)
#include <hip/hip_runtime.h>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"
#include "dmx_collide.hpp"

namespace dmx {

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

#define DMX_NP_INST(T)                                                                                                        \
    template hipError_t launch_np_convex_plane<T>(const T *, const uint8_t *, int64_t, const StepParams<T> &, hipStream_t);
DMX_NP_INST(float)
DMX_NP_INST(double)

}  // namespace dmx
