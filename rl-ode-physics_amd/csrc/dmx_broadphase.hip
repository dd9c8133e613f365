// dmx_broadphase.hip -- device side of dSpaceCollide for body-body pairs (/root/reference/src/main.c:212).
//
// ODE's hash space bins geom AABBs into cells and tests geoms of the same / adjacent cells.  Here bodies are
// binned by the (x,z) column of their centre into a hashed table of fixed-capacity buckets; the cell size is
// at least the largest bounding-sphere diameter, so two bodies whose AABBs overlap always sit in adjacent
// columns (3x3 neighbourhood).  Two consumers:
//   * the exact pair search (dmx_exact.hip): every (i<j) whose AABBs overlap (what reaches NearCallback);
//   * bp_safe_zone -- per body, half the horizontal gap to its nearest neighbour's bounding sphere.  While every
//                     body stays inside its safe zone (a 3-real check fused into the step kernels) no two
//                     bounding spheres can touch, so no collider can return a CONTACT (each body lies inside its
//                     sphere) and the fused single-body kernels are exact.  AABB pairs may still exist -- the AABBs of
//                     two tilted boxes placed diagonally overlap while their spheres are apart -- they would all come
//                     back from dCollide empty, and are not enumerated on this path.
// Integer / index work: coalesced loads of positions, hashed bucket atomics in L2, no MFMA, no LDS reuse to stage.
#include <hip/hip_runtime.h>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"
#include "dmx_grid.hpp"
#include "dmx_collide_wave.hpp"

namespace dmx {

// bodies [0,n) -> buckets of their (x,z) column
template <class T>
__global__ __launch_bounds__(256) void bp_insert(T *__restrict__ S, const uint8_t *__restrict__ gtype,
                                                 int64_t stride, int64_t n, GridParams<T> G)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) grid_insert<T>(S, gtype, i, G);
}

// Convex bodies: the exact world AABB -- the bounds of the hull's transformed points, what ODE's dxConvex::computeAABB keeps
// [ODE-recall] and what the oracle's pair search tests -- over the bounding sphere's box bp_insert wrote: one wavefront per hull.
// The sphere's box is up to twice as wide as a teapot; with it, neighbours on a 3 m grid that have rocked half a metre towards one
// another are "pairs", each costing two walks over 1 265 vertices x 2 526 faces to find nothing.
template <class T>
__global__ __launch_bounds__(256) void bp_convex_aabb(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t n, GridParams<T> G)
{
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n || gtype[i] != GEOM_CONVEX) return;             // wave-uniform
    const int lane = threadIdx.x & 63;
    const V3<T> x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
    const M3<T> R = quat_to_R(Q4<T>{ S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)], S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] });
    T lo[3], hi[3];
    wave_hull_aabb<T>(x, R, G.hull, G.hull_n, lane, lo, hi);
    if (lane < 3) {        // (selects, not lo[lane]: a runtime index would send the arrays to scratch memory)
        G.rec[i].lo[lane] = lane == 0 ? lo[0] : (lane == 1 ? lo[1] : lo[2]);
        G.rec[i].hi[lane] = lane == 0 ? hi[0] : (lane == 1 ? hi[1] : hi[2]);
    }
}

// per active body: build position (x,z) and safe radius = half the horizontal gap to the nearest bounding sphere
template <class T>
__global__ __launch_bounds__(256) void bp_safe_zone(T *__restrict__ S, const uint8_t *__restrict__ gtype,
                                                    int64_t stride, int64_t n_active, GridParams<T> G)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_active) return;
    const int gt = gtype[i];
    const T x = S[slab_ix(C_POS + 0, i)], z = S[slab_ix(C_POS + 2, i)];
    T safe = Limits<T>::inf();
    if (gt != GEOM_NONE) {
        const T ri = bound_radius<T>(gt, S, i);
        const int ix = (int)floor((double)(x * G.inv_cell)), iz = (int)floor((double)(z * G.inv_cell));
        // nothing outside the 3x3 block is closer than one cell: gap >= cell - r_i - (largest radius of a class i collides with)
        T rm = T(0);                 // the largest radius among the classes this body's class collides with (0: none of them here)
#pragma unroll
        for (int c = 1; c < 4; c++) if (classes_collide(gt, c, G.class_pairs) && G.r_cls[c] > rm) rm = G.r_cls[c];
        T gap = rm > T(0) ? G.cell - ri - rm : Limits<T>::inf();
        for (int dz = -1; dz <= 1; dz++)
            for (int dx = -1; dx <= 1; dx++) {
                const uint32_t h = cell_hash(ix + dx, iz + dz, G.mask, G.xbits);
                uint32_t cnt = G.count[h];
                if (cnt > (uint32_t)G.cap) cnt = (uint32_t)G.cap;
                for (uint32_t s = 0; s < cnt; s++) {
                    const int64_t j = G.items[(size_t)h * G.cap + s];
                    if (j == i || !classes_collide(gt, gtype[j], G.class_pairs)) continue;
                    const T ddx = S[slab_ix(C_POS + 0, j)] - x, ddz = S[slab_ix(C_POS + 2, j)] - z;
                    const T g = tsqrt<T>(ddx * ddx + ddz * ddz) - ri - S[slab_ix(C_BPR, j)];
                    if (g < gap) gap = g;
                }
            }
        safe = T(0.5) * gap;
        // bounding sphere reaches into a static box's AABB: without the static fused path (G.static_fast) only the exact path
        // can step it; with it, step_contacts does, and says so itself (BPF_NOFAST) when a body's contacts overflow its buffer
        bool near_static = false;
        const T y = S[slab_ix(C_POS + 1, i)];
        for (int s = 0; s < (G.static_fast ? 0 : G.n_static); s++) {
            const T *b = G.sbox + s * SBOX_REALS;
            if (!(x - ri > b[SBOX_HI + 0] || x + ri < b[SBOX_LO + 0] || y - ri > b[SBOX_HI + 1] || y + ri < b[SBOX_LO + 1] ||
                  z - ri > b[SBOX_HI + 2] || z + ri < b[SBOX_LO + 2])) near_static = true;
        }
        if (!(safe > 0) || near_static) atomicAdd(&G.flags[BPF_CROWDED], 1u);
    }
    S[slab_ix(C_BPX, i)] = x;
    S[slab_ix(C_BPZ, i)] = z;
    S[slab_ix(C_BPSAFE, i)] = safe;
}

// the grid's bucket counts and flags to zero, and up to two small records with them (the exact tick's count record and the
// island solve's diagnostics): one launch where four memsets went
__global__ __launch_bounds__(256) void bp_clear(uint32_t *__restrict__ count, size_t n_count, uint32_t *__restrict__ flags,
                                                uint32_t *__restrict__ rec_a, int words_a, uint32_t *__restrict__ rec_b, int words_b)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t k = t; k < n_count; k += (size_t)gridDim.x * blockDim.x) count[k] = 0u;
    if (t < (size_t)BPF_COUNT) flags[t] = 0u;
    if (rec_a != nullptr && t < (size_t)words_a) rec_a[t] = 0u;
    if (rec_b != nullptr && t < (size_t)words_b) rec_b[t] = 0u;
}

hipError_t launch_bp_clear(uint32_t *count, size_t n_count, uint32_t *flags, void *rec_a, size_t bytes_a, void *rec_b, size_t bytes_b,
                           hipStream_t st)
{
    const size_t g = (n_count + 255) / 256;
    hipLaunchKernelGGL(bp_clear, dim3((unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g))), dim3(256), 0, st, count, n_count, flags,
                       (uint32_t *)rec_a, (int)(bytes_a / 4), (uint32_t *)rec_b, (int)(bytes_b / 4));
    return hipGetLastError();
}

static inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

template <class T>
hipError_t launch_bp_insert(T *S, const uint8_t *gtype, int64_t stride, int64_t n, const GridParams<T> &G, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL((bp_insert<T>), dim3(nblk(n)), dim3(256), 0, st, S, gtype, stride, n, G);
    return hipGetLastError();
}
template <class T>
hipError_t launch_bp_convex_aabb(const T *S, const uint8_t *gtype, int64_t n, const GridParams<T> &G, hipStream_t st)
{
    if (n <= 0 || G.hull_n <= 0 || G.rec == nullptr) return hipSuccess;
    hipLaunchKernelGGL((bp_convex_aabb<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, S, gtype, n, G);
    return hipGetLastError();
}
template <class T>
hipError_t launch_bp_safe_zone(T *S, const uint8_t *gtype, int64_t stride, int64_t n_active, const GridParams<T> &G, hipStream_t st)
{
    if (n_active <= 0) return hipSuccess;
    hipLaunchKernelGGL((bp_safe_zone<T>), dim3(nblk(n_active)), dim3(256), 0, st, S, gtype, stride, n_active, G);
    return hipGetLastError();
}
#define DMX_BP_INST(T)                                                                                              \
    template hipError_t launch_bp_convex_aabb<T>(const T *, const uint8_t *, int64_t, const GridParams<T> &, hipStream_t); \
    template hipError_t launch_bp_insert<T>(T *, const uint8_t *, int64_t, int64_t, const GridParams<T> &, hipStream_t); \
    template hipError_t launch_bp_safe_zone<T>(T *, const uint8_t *, int64_t, int64_t, const GridParams<T> &, hipStream_t);
DMX_BP_INST(float)
DMX_BP_INST(double)

// HIP loads a translation unit's code object at the first launch of one of its kernels -- a couple of milliseconds each, which an
// interactive caller would meet as a hitch at the first tick that needs the exact pipeline.  dmxBatchCreate asks for one
// kernel's attributes per unit instead (dmx_preload_code, dmx_batch.cpp): the load happens there.
hipError_t dmx_touch_broadphase(int real_bytes)
{
    // (the unit's code object, and -- what costs more -- each kernel's own first-use set-up: every kernel an exact tick or a fused
    //  tick may launch, in the batch's precision)
    hipFuncAttributes a;
    hipError_t e = hipSuccess;
    auto touch = [&](const void *k) { const hipError_t r = hipFuncGetAttributes(&a, k); if (r != hipSuccess) e = r; };
    touch((const void *)&bp_clear);
    if (real_bytes == 4) {
        touch((const void *)&bp_insert<float>);
        touch((const void *)&bp_safe_zone<float>);
    } else {
        touch((const void *)&bp_insert<double>);
        touch((const void *)&bp_safe_zone<double>);
    }
    return e;
}

}  // namespace dmx
