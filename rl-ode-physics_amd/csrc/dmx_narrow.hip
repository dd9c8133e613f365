// dmx_narrow.hip -- dCollide (/root/reference/src/main.c:678) on the FUSED paths: convex hulls against the ground plane
// (one wavefront per hull leaves the hull's contacts where step_plane<.., 8> picks them up), and every body class against
// the ground plane and the static boxes (np_static / np_convex_static, read by step_contacts).  (The narrowphase of
// the exact tick -- boxes / spheres / hulls against plane, static boxes and one another -- is in dmx_exact.hip.)
#include <hip/hip_runtime.h>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"
#include "dmx_collide.hpp"
#include "dmx_collide_wave.hpp"

namespace dmx {

// ---- convex hull against the ground plane (dCollideConvexPlane, wave_convex_plane): one wavefront per convex body --
// Output: cbuf[i][k] = (x, y, z, depth) for k < ccount[i]; bodies of other classes are left alone.
template <class T>
__global__ __launch_bounds__(256) void np_convex_plane(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t n,
                                                       StepParams<T> P)
{
    // a chunk in which a body has left its zone is rolled back whole, and a speculative launch the device's record refuses does
    // nothing: the step kernel behind this one returns at once in both cases (step_plane / step_contacts), so its contacts need not be made
    if ((P.bp_check && P.bp_flags[BPF_VIOLATION] != 0u) || (P.gate != nullptr && *P.gate == 0u)) return;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // 4 wavefronts per workgroup
    if (i >= n || gtype[i] != GEOM_CONVEX) return;                         // wave-uniform
    const int lane = threadIdx.x & 63;
    const V3<T> x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
    const M3<T> R = quat_to_R(Q4<T>{ S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)],
                                     S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] });
    const int maxc = P.max_contacts < CONVEX_MAXC ? P.max_contacts : CONVEX_MAXC;
    T *out = P.cbuf + (size_t)i * CONVEX_MAXC * 4;
    const int nc = wave_convex_plane<T>(x, R, P, maxc, lane, [&](int rank, const V3<T> &p, const V3<T> &, T dep) {
        out[4 * rank + 0] = p.x; out[4 * rank + 1] = p.y; out[4 * rank + 2] = p.z; out[4 * rank + 3] = dep; });
    if (lane == 0) P.ccount[i] = nc;
}

// =====================================================================================================================
// The fused path of bodies at static geometry: dSpaceCollide + NearCallback (main.c:212, 674-693) for the pairs
// (ground plane, body) and (static box, body) -- every pair a single-body island can have.  Contacts go to P.sbuf in
// joint creation order (the plane's, then the static boxes' in their order: the reference creates its map before any body,
// main.c:115-121) and in canonical form: the body is body 1, the normal points into it.  A static box is geom 1 of its
// dCollide call and its joint is attached (0, body), i.e. reversed [ODE-recall: dJointAttach swaps and sets dJOINT_REVERSE,
// contact getInfo2 negates the normal]: box_box's normal is negated, sphere_box -- called in its own (sphere, box) order,
// which already flips -- is not.  Same colliders, same AABB test (body_aabb against the static box's) as the exact tick's
// ex_narrow, so both paths keep the same contacts.  step_contacts (dmx_kernels.hip) solves and integrates.
// =====================================================================================================================
template <class T> __device__ __forceinline__ void put_sc(T *sbuf, int64_t i, int k, const V3<T> &p, const V3<T> &n, T d)
{
    sbuf[sc_ix(k, SC_POS + 0, i)] = p.x; sbuf[sc_ix(k, SC_POS + 1, i)] = p.y; sbuf[sc_ix(k, SC_POS + 2, i)] = p.z;
    sbuf[sc_ix(k, SC_NORMAL + 0, i)] = n.x; sbuf[sc_ix(k, SC_NORMAL + 1, i)] = n.y; sbuf[sc_ix(k, SC_NORMAL + 2, i)] = n.z;
    sbuf[sc_ix(k, SC_DEPTH, i)] = d;
}
template <class T> __device__ __forceinline__ bool aabb_meets_static(const T lo[3], const T hi[3], const T *b)
{
    return !(lo[0] > b[SBOX_HI + 0] || b[SBOX_LO + 0] > hi[0] || lo[1] > b[SBOX_HI + 1] || b[SBOX_LO + 1] > hi[1] ||
             lo[2] > b[SBOX_HI + 2] || b[SBOX_LO + 2] > hi[2]);
}

// boxes and spheres: one lane per body
template <class T>
__global__ __launch_bounds__(256) void np_static(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t n, StepParams<T> P)
{
    // a chunk in which a body has left its zone is rolled back whole, and a speculative launch the device's record refuses does
    // nothing: the step kernel behind this one returns at once in both cases (step_plane / step_contacts), so its contacts need not be made
    if ((P.bp_check && P.bp_flags[BPF_VIOLATION] != 0u) || (P.gate != nullptr && *P.gate == 0u)) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (P.skip != nullptr && P.skip[i])) return;
    const int gt = gtype[i];
    if (gt == GEOM_CONVEX) return;                          // np_convex_static
    int nc = 0;
    if (gt == GEOM_BOX || gt == GEOM_SPHERE) {
        const V3<T> x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
        const T side[3] = { S[slab_ix(C_SIDES + 0, i)], S[slab_ix(C_SIDES + 1, i)], S[slab_ix(C_SIDES + 2, i)] };
        M3<T> R;
        T r[3];
        if (gt == GEOM_BOX) {
            R = quat_to_R(Q4<T>{ S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)], S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] });
            for (int a = 0; a < 3; a++)       // (body_aabb, dmx_grid.hpp)
                r[a] = T(0.5) * (tabs(R.m[a][0] * side[0]) + tabs(R.m[a][1] * side[1]) + tabs(R.m[a][2] * side[2]));
        } else {
            for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) R.m[a][c] = a == c ? T(1) : T(0);
            r[0] = r[1] = r[2] = side[0];
        }
        const T lo[3] = { x.x - r[0], x.y - r[1], x.z - r[2] }, hi[3] = { x.x + r[0], x.y + r[1], x.z + r[2] };
        if (P.plane_on) {
            V3<T> cp[4]; T cd[4];
            const int k = gt == GEOM_BOX ? box_plane(x, R, side, P.pn, P.pd, P.max_contacts, cp, cd) : sphere_plane(x, side[0], P.pn, P.pd, cp, cd);
            for (int c = 0; c < k; c++) put_sc(P.sbuf, i, c, cp[c], P.pn, cd[c]);
            nc = k;
        }
        const int mc = P.max_contacts > 8 ? 8 : P.max_contacts;
        for (int s = 0; s < P.n_static; s++) {
            const T *sb = P.sbox + s * SBOX_REALS;
            if (!aabb_meets_static(lo, hi, sb)) continue;
            const V3<T> sx = { sb[SBOX_POS], sb[SBOX_POS + 1], sb[SBOX_POS + 2] };
            M3<T> sR;
            for (int a = 0; a < 3; a++) for (int c2 = 0; c2 < 3; c2++) sR.m[a][c2] = sb[SBOX_R + 3 * a + c2];
            const T sside[3] = { sb[SBOX_SIDE], sb[SBOX_SIDE + 1], sb[SBOX_SIDE + 2] };
            ContactPoint<T> c[8];
            int k;
            bool negate = true;
            if (gt == GEOM_BOX) k = box_box(sx, sR, sside, x, R, side, mc, c);
            else { k = sphere_box(x, side[0], sx, sR, sside, c); negate = false; }
            if (k > mc) k = mc;
#pragma unroll
            for (int q = 0; q < 8; q++) {               // (static indices: the contacts stay in registers)
                if (q < k && nc + q < SC_MAXC) {
                    const V3<T> nn = negate ? V3<T>{ -c[q].normal.x, -c[q].normal.y, -c[q].normal.z } : c[q].normal;
                    put_sc(P.sbuf, i, nc + q, c[q].pos, nn, c[q].depth);
                }
            }
            nc += k;
        }
    }
    P.scount[i] = nc > SC_MAXC ? SC_MAXC + 1 : nc;
}

// convex hulls: one wavefront per body (bodies of other classes leave at once)
template <class T>
__global__ __launch_bounds__(256) void np_convex_static(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t n, StepParams<T> P)
{
    // (Tried on 16 384 teapots on a floor box, 0.071 ms/tick as it is: the hull's points staged in LDS once per workgroup, 0.081 --
    //  the walk is bound by instruction issue, the points come from L1 anyway; a cull in the hull's frame first -- a point's box
    //  coordinates by nine multiply-adds, a pass of 64 points skipped when none is near -- 0.083: a pass of 64 consecutive points
    //  spans the whole teapot, so nearly every pass holds a point near the floor.)
    // a chunk in which a body has left its zone is rolled back whole, and a speculative launch the device's record refuses does
    // nothing: the step kernel behind this one returns at once in both cases (step_plane / step_contacts), so its contacts need not be made
    if ((P.bp_check && P.bp_flags[BPF_VIOLATION] != 0u) || (P.gate != nullptr && *P.gate == 0u)) return;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n || gtype[i] != GEOM_CONVEX || (P.skip != nullptr && P.skip[i])) return;      // wave-uniform
    const int lane = threadIdx.x & 63;
    const V3<T> x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
    const M3<T> R = quat_to_R(Q4<T>{ S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)],
                                     S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] });
    const T radius = S[slab_ix(C_SIDES + 0, i)];          // the hull's bounding radius; its AABB is that sphere's box (body_aabb)
    const T lo[3] = { x.x - radius, x.y - radius, x.z - radius }, hi[3] = { x.x + radius, x.y + radius, x.z + radius };
    const int maxc = P.max_contacts < CONVEX_MAXC ? P.max_contacts : CONVEX_MAXC;
    int nc = 0;
    if (P.hull_n > 0) {
        if (P.plane_on)
            nc = wave_convex_plane<T>(x, R, P, maxc, lane, [&](int rank, const V3<T> &p, const V3<T> &nn, T dep) { put_sc(P.sbuf, i, rank, p, nn, dep); });
        for (int s = 0; s < P.n_static; s++) {
            const T *sb = P.sbox + s * SBOX_REALS;
            if (!aabb_meets_static(lo, hi, sb)) continue;
            const V3<T> sx = { sb[SBOX_POS], sb[SBOX_POS + 1], sb[SBOX_POS + 2] };
            M3<T> sR;
            for (int a = 0; a < 3; a++) for (int c2 = 0; c2 < 3; c2++) sR.m[a][c2] = sb[SBOX_R + 3 * a + c2];
            const T sside[3] = { sb[SBOX_SIDE], sb[SBOX_SIDE + 1], sb[SBOX_SIDE + 2] };
            const int base = nc;
            nc += wave_box_convex<T>(sx, sR, sside, x, R, radius, P, maxc, true, lane, [&](int rank, const V3<T> &p, const V3<T> &nn, T dep) {
                if (base + rank < SC_MAXC) put_sc(P.sbuf, i, base + rank, p, nn, dep); });
        }
    }
    if (lane == 0) P.scount[i] = nc > SC_MAXC ? SC_MAXC + 1 : nc;
}

template <class T>
hipError_t launch_np_convex_plane(const T *S, const uint8_t *gtype, int64_t n, const StepParams<T> &P, hipStream_t st)
{
    if (n <= 0 || P.hull_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((np_convex_plane<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, S, gtype, n, P);
    return hipGetLastError();
}

template <class T>
hipError_t launch_np_static(const T *S, const uint8_t *gtype, int64_t n, const StepParams<T> &P, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    // (np_static serves boxes and spheres and writes a zero count for slots of no class; a batch of hulls only does not need the
    //  launch: slots of no class keep the zero count dmxBatchSetStaticBoxes / dmxBatchUploadGeomType left there)
    if (P.has_simple || P.hull_n <= 0) hipLaunchKernelGGL((np_static<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, gtype, n, P);
    if (P.hull_n > 0) hipLaunchKernelGGL((np_convex_static<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, S, gtype, n, P);
    return hipGetLastError();
}

#define DMX_NP_INST(T)                                                                                                        \
    template hipError_t launch_np_static<T>(const T *, const uint8_t *, int64_t, const StepParams<T> &, hipStream_t);        \
    template hipError_t launch_np_convex_plane<T>(const T *, const uint8_t *, int64_t, const StepParams<T> &, hipStream_t);
DMX_NP_INST(float)
DMX_NP_INST(double)

// HIP loads a translation unit's code object at the first launch of one of its kernels -- a couple of milliseconds each, which an
// interactive caller would meet as a hitch at the first tick that needs the exact pipeline.  dmxBatchCreate asks for one
// kernel's attributes per unit instead (dmx_preload_code, dmx_batch.cpp): the load happens there.
hipError_t dmx_touch_narrow(int real_bytes)
{
    // (the unit's code object, and -- what costs more -- each kernel's own first-use set-up: every kernel an exact tick or a fused
    //  tick may launch, in the batch's precision)
    hipFuncAttributes a;
    hipError_t e = hipSuccess;
    auto touch = [&](const void *k) { const hipError_t r = hipFuncGetAttributes(&a, k); if (r != hipSuccess) e = r; };
    if (real_bytes == 4) {
        touch((const void *)&np_static<float>);
    } else {
        touch((const void *)&np_static<double>);
    }
    return e;
}

}  // namespace dmx
