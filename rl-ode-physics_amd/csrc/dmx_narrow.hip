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
                                           int32_t *, hipStream_t);
DMX_NP_INST(float)
DMX_NP_INST(double)

}  // namespace dmx
