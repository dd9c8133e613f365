// dmx_grid.hpp -- the hashed (x,z)-column grid of the body-body broadphase: device helpers shared by the kernels that
// fill it (dmx_broadphase.hip) and the exact tick's pair search (dmx_exact.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"

namespace dmx {

// Cell (ix, iz) -> bucket.  Torus form: entry (iz mod rows) * 2^xbits + (ix mod 2^xbits), so a wavefront of bodies
// adjacent in space looks up adjacent entries (a handful of cache lines instead of one per lookup); cells a whole
// torus period apart share a bucket, which callers tell apart by the bodies' true cells.  Scenes far longer than
// wide wrap too often for that; they use the scrambled form (xbits = 0).
__device__ __forceinline__ uint32_t cell_hash(int ix, int iz, uint32_t mask, int xbits)
{
    if (xbits > 0) return ((((uint32_t)iz) << xbits) | ((uint32_t)ix & ((1u << xbits) - 1u))) & mask;
    return ((uint32_t)ix * 73856093u ^ (uint32_t)iz * 19349663u) & mask;
}

template <class T> __device__ __forceinline__ T bound_radius(int gt, const T *S, int64_t i)
{
    const T sx = S[slab_ix(C_SIDES + 0, i)];
    if (gt == GEOM_SPHERE || gt == GEOM_CONVEX) return sx;       // convex: hull bounding radius
    const T sy = S[slab_ix(C_SIDES + 1, i)], sz = S[slab_ix(C_SIDES + 2, i)];
    return T(0.5) * tsqrt<T>(sx * sx + sy * sy + sz * sz);
}

template <class T> __device__ __forceinline__ void body_aabb(const T *S, const uint8_t *gtype, int64_t i, T lo[3], T hi[3])
{
    const T p[3] = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
    T r[3];
    if (gtype[i] == GEOM_SPHERE || gtype[i] == GEOM_CONVEX) {      // convex: the bounding sphere's box (conservative)
        r[0] = r[1] = r[2] = S[slab_ix(C_SIDES + 0, i)];
    } else {
        const Q4<T> q = { S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)],
                          S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] };
        const M3<T> R = quat_to_R(q);
        const T s[3] = { S[slab_ix(C_SIDES + 0, i)], S[slab_ix(C_SIDES + 1, i)], S[slab_ix(C_SIDES + 2, i)] };
        for (int a = 0; a < 3; a++)
            r[a] = T(0.5) * (tabs(R.m[a][0] * s[0]) + tabs(R.m[a][1] * s[1]) + tabs(R.m[a][2] * s[2]));
    }
    for (int a = 0; a < 3; a++) { lo[a] = p[a] - r[a]; hi[a] = p[a] + r[a]; }
}

// body i -> the bucket of its (x,z) column; leaves its bounding radius in the slab and (when the grid carries the array)
// its AABB for the exact pair search, which tests every candidate's
template <class T> __device__ __forceinline__ void grid_insert(T *S, const uint8_t *gtype, int64_t i, const GridParams<T> &G)
{
    // (convex bodies take part with their bounding sphere's box: conservative, so only candidates are added)
    if (gtype[i] == GEOM_NONE) return;
    S[slab_ix(C_BPR, i)] = bound_radius<T>(gtype[i], S, i);      // neighbours read this instead of 3 sides + sqrt
    const int ix = (int)floor((double)(S[slab_ix(C_POS + 0, i)] * G.inv_cell));
    const int iz = (int)floor((double)(S[slab_ix(C_POS + 2, i)] * G.inv_cell));
    if (G.rec != nullptr) {
        GridRec<T> r;
        body_aabb<T>(S, gtype, i, r.lo, r.hi);
        r.ix = ix; r.iz = iz;
        G.rec[i] = r;
    }
    const uint32_t h = cell_hash(ix, iz, G.mask, G.xbits);
    const uint32_t slot = atomicAdd(&G.count[h], 1u);
    if (slot < (uint32_t)G.cap) G.items[(size_t)h * G.cap + slot] = (int32_t)i;
    else atomicOr(&G.flags[BPF_OVERFLOW], 1u);
}

}  // namespace dmx
