// dmx_kernels.hip -- gfx950 kernels of the rigid-body step hot path.
//
// Replaces, per tick, the three ODE calls of the reference's physics loop
// (/root/reference/src/main.c:211-215): dSpaceCollide + NearCallback
// (main.c:674-693), dWorldStep stepped with QuickStep semantics, and
// dJointGroupEmpty -- for scenes whose dynamics islands are single bodies
// (free bodies, and bodies in contact with the static ground plane only).
//
// Data layout: one slab `S` of `real` in tiles of 64 bodies x 30 components
// (dmx_internal.hpp: component c of body i at S[slab_ix(c, i)]), so that a
// wavefront's 30 component accesses fall in one contiguous 7.5 KiB (f32) run.
// A lane owns V consecutive bodies (V divides the tile).
//
// No MFMA: there is no dense contraction on this path.  integrate_free is
// HBM-bound (30 reals per body-step); step_plane is VALU/latency bound
// (20 SOR sweeps over <= 12 rows held in registers).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"
#include "dmx_collide.hpp"
#include "dmx_step_fused.hpp"

namespace dmx {

template <class T, int V> struct alignas(sizeof(T) * V) Pack { T v[V]; };

template <class T, int V>
__device__ __forceinline__ Pack<T, V> ldv(const T *__restrict__ base, int64_t stride, int comp, int64_t i)
{
    return *reinterpret_cast<const Pack<T, V> *>(base + slab_ix(comp, i));
}
template <class T, int V>
__device__ __forceinline__ void stv(T *__restrict__ base, int64_t stride, int comp, int64_t i, const Pack<T, V> &p)
{
    *reinterpret_cast<Pack<T, V> *>(base + slab_ix(comp, i)) = p;
}
// the same with the non-temporal hint (streamed once: do not keep the line in cache); launch tuning, StepParams::nt
template <class T, int V>
__device__ __forceinline__ Pack<T, V> ldv_nt(const T *base, int64_t stride, int comp, int64_t i)
{
    Pack<T, V> p;
    const T *a = base + slab_ix(comp, i);
#pragma unroll
    for (int b = 0; b < V; b++) p.v[b] = __builtin_nontemporal_load(a + b);
    return p;
}
template <class T, int V>
__device__ __forceinline__ void stv_nt(T *base, int64_t stride, int comp, int64_t i, const Pack<T, V> &p)
{
    T *a = base + slab_ix(comp, i);
#pragma unroll
    for (int b = 0; b < V; b++) __builtin_nontemporal_store(p.v[b], a + b);
}

// ---------------------------------------------------------------------------------------------
// integrate_free: contact-free tick (BASELINE config 2/4).  Algorithmic traffic per body-step:
// read 13 state + 4 constant reals, write 13 state reals = 30 reals (120 B f32 / 240 B f64).
// ---------------------------------------------------------------------------------------------
template <class T, int V, bool EXT, int MINW, bool MULTI>
__global__ __launch_bounds__(256, MINW) void integrate_free(T *S, T *So, int64_t stride, int64_t nvec,
                                                      StepParams<T> P)
{
    // So = where the new state goes: S itself (in place) or the batch's other slab (the first launch of a
    // collision-proof chunk, which thereby leaves the chunk's start state behind as the rollback snapshot)
    if (P.gate != nullptr && *P.gate == 0u) return;
    const int nticks = MULTI ? P.ticks : 1;         // MULTI = false: the one-tick kernel, no loop
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < nvec;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t * V;
        if (V == 1 && P.skip != nullptr && P.skip[i]) continue;      // this body belongs to the island path this tick
        Pack<T, V> c[C_SIDES];
        if (P.nt & 2) {
#pragma unroll
            for (int k = 0; k < C_SIDES; k++) c[k] = ldv_nt<T, V>(S, stride, k, i);
        } else {
#pragma unroll
            for (int k = 0; k < C_SIDES; k++) c[k] = ldv<T, V>(S, stride, k, i);
        }
        Pack<T, V> bx, bz, bs;
        if (P.bp_check) {
            bx = ldv<T, V>(S, stride, C_BPX, i); bz = ldv<T, V>(S, stride, C_BPZ, i); bs = ldv<T, V>(S, stride, C_BPSAFE, i);
        }
        Pack<T, V> f[6];
        if (EXT) {
#pragma unroll
            for (int k = 0; k < 6; k++) f[k] = ldv<T, V>(S, stride, C_FORCE + k, i);
        }
        // P.ticks ticks with the state in registers: one read and one write of the state per launch, not per tick
        for (int s = 0; s < nticks; s++) {
            if (P.bp_check && ((P.bp_check & BPC_ALL) || (s == 0 && (P.bp_check & BPC_FIRST)) ||
                               (s == nticks - 1 && (P.bp_check & BPC_LAST)))) {
                // dSpaceCollide for body-body pairs, by proof: a body inside its safe zone cannot touch any other
                int zs = 0;
#pragma unroll
                for (int b = 0; b < V; b++) {
                    int z = zone_state(c[C_POS].v[b] - bx.v[b], c[C_POS + 2].v[b] - bz.v[b], bs.v[b]);
                    if (P.n_static > 0) {
                        const int z2 = static_state(P, c[C_POS].v[b], c[C_POS + 1].v[b], c[C_POS + 2].v[b], S[slab_ix(C_BPR, i + b)]);
                        z = z2 > z ? z2 : z;
                    }
                    zs = z > zs ? z : zs;
                }
                report_zone(zs, P.bp_flags);
            }
#pragma unroll
            for (int b = 0; b < V; b++) {
                V3<T> x = { c[C_POS].v[b], c[C_POS + 1].v[b], c[C_POS + 2].v[b] };
                Q4<T> q = { c[C_QUAT].v[b], c[C_QUAT + 1].v[b], c[C_QUAT + 2].v[b], c[C_QUAT + 3].v[b] };
                V3<T> v = { c[C_LVEL].v[b], c[C_LVEL + 1].v[b], c[C_LVEL + 2].v[b] };
                V3<T> w = { c[C_AVEL].v[b], c[C_AVEL + 1].v[b], c[C_AVEL + 2].v[b] };
                const V3<T> Ib = { c[C_INERTIA].v[b], c[C_INERTIA + 1].v[b], c[C_INERTIA + 2].v[b] };
                V3<T> facc = { T(0), T(0), T(0) }, tacc = { T(0), T(0), T(0) };
                if (EXT && s == 0) {                       // the accumulators act in the first tick and are cleared by it
                    facc = { f[0].v[b], f[1].v[b], f[2].v[b] };
                    tacc = { f[3].v[b], f[4].v[b], f[5].v[b] };
                }
                free_body_step(x, q, v, w, c[C_MASS].v[b], Ib, facc, tacc, P.g, P.h, P.gyro);
                if (s == nticks - 1) pack_boundary(P, i + b, x, q, v, w);
                c[C_POS].v[b] = x.x; c[C_POS + 1].v[b] = x.y; c[C_POS + 2].v[b] = x.z;
                c[C_QUAT].v[b] = q.w; c[C_QUAT + 1].v[b] = q.x; c[C_QUAT + 2].v[b] = q.y; c[C_QUAT + 3].v[b] = q.z;
                c[C_LVEL].v[b] = v.x; c[C_LVEL + 1].v[b] = v.y; c[C_LVEL + 2].v[b] = v.z;
                c[C_AVEL].v[b] = w.x; c[C_AVEL + 1].v[b] = w.y; c[C_AVEL + 2].v[b] = w.z;
            }
        }
        if (P.nt & 1) {
#pragma unroll
            for (int k = 0; k < C_MASS; k++) stv_nt<T, V>(So, stride, k, i, c[k]);
        } else {
#pragma unroll
            for (int k = 0; k < C_MASS; k++) stv<T, V>(So, stride, k, i, c[k]);
        }
        if (EXT) {
            Pack<T, V> z;
#pragma unroll
            for (int b = 0; b < V; b++) z.v[b] = T(0);
#pragma unroll
            for (int k = 0; k < 6; k++) stv<T, V>(S, stride, C_FORCE + k, i, z);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// integrate_free_wide: the same contact-free tick with the tile moved in 16-byte pieces.  integrate_free's lanes load and store
// one real each (a wavefront's request is 256 B); here a wavefront streams its tile's 17 input components as ONE flat run in
// dwordx4 pieces (1 KiB per request) into LDS, every lane picks its body's components out of LDS, steps it, and the 13 new
// state components go back through LDS and out as dwordx4 stores.  Same arithmetic (free_body_step), same bits; an experiment
// on the memory system's request granularity (DMX_WIDE=1; DESIGN.md section 4, HBM-resident sizes).
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void integrate_free_wide(T *S, T *So, int64_t ntiles, StepParams<T> P)
{
    constexpr int W16 = 16 / sizeof(T);                       // reals per 16-byte piece
    constexpr int IN = C_SIDES * SLAB_TILE, OUT = C_MASS * SLAB_TILE;      // reals in, reals out per tile
    __shared__ __align__(16) T stage[4][IN];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    const bool live = tile < ntiles;
    T *ls = stage[wv];
    const int64_t base = tile * (int64_t)(C_COUNT * SLAB_TILE);
    if (live) {
#pragma unroll
        for (int k = 0; k * 64 * W16 < IN; k++) {
            const int f = (k * 64 + lane) * W16;
            if (f < IN) *reinterpret_cast<Pack<T, W16> *>(ls + f) = *reinterpret_cast<const Pack<T, W16> *>(S + base + f);
        }
    }
    __syncthreads();
    T c[C_SIDES];
#pragma unroll
    for (int k = 0; k < C_SIDES; k++) c[k] = ls[k * SLAB_TILE + lane];
    __syncthreads();
    const int64_t i = tile * SLAB_TILE + lane;
    if (live) {
        if (P.bp_check) {
            int z = zone_state(c[C_POS] - S[slab_ix(C_BPX, i)], c[C_POS + 2] - S[slab_ix(C_BPZ, i)], S[slab_ix(C_BPSAFE, i)]);
            if (P.n_static > 0) {
                const int z2 = static_state(P, c[C_POS], c[C_POS + 1], c[C_POS + 2], S[slab_ix(C_BPR, i)]);
                z = z2 > z ? z2 : z;
            }
            report_zone(z, P.bp_flags);
        }
        V3<T> x = { c[C_POS], c[C_POS + 1], c[C_POS + 2] };
        Q4<T> q = { c[C_QUAT], c[C_QUAT + 1], c[C_QUAT + 2], c[C_QUAT + 3] };
        V3<T> v = { c[C_LVEL], c[C_LVEL + 1], c[C_LVEL + 2] };
        V3<T> w = { c[C_AVEL], c[C_AVEL + 1], c[C_AVEL + 2] };
        const V3<T> Ib = { c[C_INERTIA], c[C_INERTIA + 1], c[C_INERTIA + 2] };
        free_body_step(x, q, v, w, c[C_MASS], Ib, V3<T>{ T(0), T(0), T(0) }, V3<T>{ T(0), T(0), T(0) }, P.g, P.h, P.gyro);
        pack_boundary(P, i, x, q, v, w);
        ls[(C_POS + 0) * SLAB_TILE + lane] = x.x; ls[(C_POS + 1) * SLAB_TILE + lane] = x.y; ls[(C_POS + 2) * SLAB_TILE + lane] = x.z;
        ls[(C_QUAT + 0) * SLAB_TILE + lane] = q.w; ls[(C_QUAT + 1) * SLAB_TILE + lane] = q.x;
        ls[(C_QUAT + 2) * SLAB_TILE + lane] = q.y; ls[(C_QUAT + 3) * SLAB_TILE + lane] = q.z;
        ls[(C_LVEL + 0) * SLAB_TILE + lane] = v.x; ls[(C_LVEL + 1) * SLAB_TILE + lane] = v.y; ls[(C_LVEL + 2) * SLAB_TILE + lane] = v.z;
        ls[(C_AVEL + 0) * SLAB_TILE + lane] = w.x; ls[(C_AVEL + 1) * SLAB_TILE + lane] = w.y; ls[(C_AVEL + 2) * SLAB_TILE + lane] = w.z;
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 0; k * 64 * W16 < OUT; k++) {
            const int f = (k * 64 + lane) * W16;
            if (f < OUT) *reinterpret_cast<Pack<T, W16> *>(So + base + f) = *reinterpret_cast<const Pack<T, W16> *>(ls + f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// integrate_free_dma: the contact-free tick with its READS taken off the register path.  A persistent grid; every wavefront walks
// tiles, and a tile's 17 input components -- one flat run of the slab -- arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB per
// request, no VGPR held while in flight) into one of the wave's two LDS buffers while the wave steps the tile before it: two
// tiles of reads in flight per wave.  The counters behind this (profiles/r03_tcc_counters_16Mi.txt): at HBM-resident sizes the
// pass is bound by how many read requests a compute unit keeps in flight (about 110-125 of 128 B through the vector L1), not by
// DRAM credits or by request size.  Same arithmetic (free_body_step), same bits (DMX_WIDE=2).
// ---------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ void dma_tile_in(const T *src, T *lds, int lane)
{
    constexpr int BYTES = C_SIDES * SLAB_TILE * (int)sizeof(T), FULL = BYTES / 1024, REM4 = (BYTES - FULL * 1024) / 256;
    const char *g = reinterpret_cast<const char *>(src);
    char *l = reinterpret_cast<char *>(lds);
#pragma unroll
    for (int k = 0; k < FULL; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + k * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(l + k * 1024), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < REM4; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + FULL * 1024 + k * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void *)(l + FULL * 1024 + k * 256), 4, 0, 0);
}
template <class T> constexpr int dma_loads_per_tile()
{
    return (C_SIDES * SLAB_TILE * (int)sizeof(T)) / 1024 + ((C_SIDES * SLAB_TILE * (int)sizeof(T)) % 1024) / 256;
}

template <class T>
__global__ __launch_bounds__(256) void integrate_free_dma(T *S, T *So, int64_t ntiles, StepParams<T> P)
{
    constexpr int IN = C_SIDES * SLAB_TILE;
    extern __shared__ __align__(16) unsigned char dma_raw[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    T *buf0 = reinterpret_cast<T *>(dma_raw) + (size_t)(2 * wv) * IN, *buf1 = buf0 + IN;
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    if (tile >= ntiles) return;
    constexpr int TILE_REALS = C_COUNT * SLAB_TILE;
    // counted waits need every vector-memory operation between two waits to be known: the 13 state stores and the next tile's
    // loads.  Ticks that also test zones or pack boundary rows (chunk ends, exchange ticks) wait for everything instead.
    const bool counted = !P.bp_check && P.pack_out == nullptr;
    dma_tile_in<T>(S + tile * TILE_REALS, buf0, lane);
    int cur = 0;
    bool first = true;
    for (; tile < ntiles; tile += step, cur ^= 1) {
        T *ls = cur ? buf1 : buf0;
        const int64_t next = tile + step;
        if (next < ntiles) dma_tile_in<T>(S + next * TILE_REALS, cur ? buf0 : buf1, lane);
        // this tile's loads are older than: the tile before's 13 stores and the next tile's loads (vmcnt counts them all, in order)
        // (the wave's first tile has no stores before it)
        if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (first) {
            if (next >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (sizeof(T) == 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else if (next < ntiles) {
            if (sizeof(T) == 4) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");      // 13 + 5
            else                asm volatile("s_waitcnt vmcnt(23)" ::: "memory");      // 13 + 10
        } else asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        first = false;
        T c[C_SIDES];
#pragma unroll
        for (int k = 0; k < C_SIDES; k++) c[k] = ls[k * SLAB_TILE + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // in registers before the buffer is handed to the tile after next
        const int64_t i = tile * SLAB_TILE + lane;
        if (P.bp_check) {
            int z = zone_state(c[C_POS] - S[slab_ix(C_BPX, i)], c[C_POS + 2] - S[slab_ix(C_BPZ, i)], S[slab_ix(C_BPSAFE, i)]);
            if (P.n_static > 0) {
                const int z2 = static_state(P, c[C_POS], c[C_POS + 1], c[C_POS + 2], S[slab_ix(C_BPR, i)]);
                z = z2 > z ? z2 : z;
            }
            report_zone(z, P.bp_flags);
        }
        V3<T> x = { c[C_POS], c[C_POS + 1], c[C_POS + 2] };
        Q4<T> q = { c[C_QUAT], c[C_QUAT + 1], c[C_QUAT + 2], c[C_QUAT + 3] };
        V3<T> v = { c[C_LVEL], c[C_LVEL + 1], c[C_LVEL + 2] };
        V3<T> w = { c[C_AVEL], c[C_AVEL + 1], c[C_AVEL + 2] };
        const V3<T> Ib = { c[C_INERTIA], c[C_INERTIA + 1], c[C_INERTIA + 2] };
        free_body_step(x, q, v, w, c[C_MASS], Ib, V3<T>{ T(0), T(0), T(0) }, V3<T>{ T(0), T(0), T(0) }, P.g, P.h, P.gyro);
        pack_boundary(P, i, x, q, v, w);
        So[slab_ix(C_POS + 0, i)] = x.x; So[slab_ix(C_POS + 1, i)] = x.y; So[slab_ix(C_POS + 2, i)] = x.z;
        So[slab_ix(C_QUAT + 0, i)] = q.w; So[slab_ix(C_QUAT + 1, i)] = q.x;
        So[slab_ix(C_QUAT + 2, i)] = q.y; So[slab_ix(C_QUAT + 3, i)] = q.z;
        So[slab_ix(C_LVEL + 0, i)] = v.x; So[slab_ix(C_LVEL + 1, i)] = v.y; So[slab_ix(C_LVEL + 2, i)] = v.z;
        So[slab_ix(C_AVEL + 0, i)] = w.x; So[slab_ix(C_AVEL + 1, i)] = w.y; So[slab_ix(C_AVEL + 2, i)] = w.z;
    }
}


// ---------------------------------------------------------------------------------------------
// step_plane: fused tick for single-body islands resting on / falling onto the ground plane
// (BASELINE config 1/3): narrowphase -> contact rows (normal + 2 friction per contact) ->
// SOR-PGS sweeps -> velocity update -> integrate.  One lane per body; rows live in registers.
// Algorithmic traffic per body-step: 13 + 4 + 3 (sides) read, 13 written = 33 reals.
// ---------------------------------------------------------------------------------------------
// NC = contact slots per body: 4 (box-plane yields at most 4 contacts) or CONVEX_MAXC when the batch has convex bodies,
// whose plane contacts np_convex_plane left in P.cbuf.
template <class T, bool EXT, int MINW, int NC>
__global__ __launch_bounds__(256, MINW) void step_plane(T *S, T *So, const uint8_t *__restrict__ gtype,
                                                  int64_t stride, int64_t n, StepParams<T> P,
                                                  StepDiag *__restrict__ diag)
{
    step_plane_body<T, EXT, NC>(S, So, gtype, stride, n, P, diag, blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// step_contacts: fused tick for single-body islands at STATIC GEOMETRY -- the ground plane and the static boxes
// (AddBodyMap, main.c:735-761; the reference's floor is one, main.c:115).  The contacts come from np_static /
// np_convex_static (dmx_narrow.hip) in joint creation order and canonical form; here: rows (normal + 2 friction per
// contact, every contact with its own normal) -> SOR-PGS sweeps -> velocity update -> integrate, one lane per body, rows in
// registers, the arithmetic of step_plane / solve_singles operation for operation (same bits as the oracle).
// Two instantiations share a tick: NC = 4 steps the bodies with 0..4 contacts (free bodies included), NC = 8 -- launched
// behind it when the batch may have such bodies (P.have8) -- those with 5..8; each leaves the other's lanes alone.  A body
// with more contacts than the buffer holds raises BPF_NOFAST (the exact path steps it: the chunk is rolled back).
// Algorithmic traffic per body-step: 13 + 4 read, 13 written, + 7 reals per contact written and read.
// ---------------------------------------------------------------------------------------------
// ALL (with NC = 8): this one launch steps every body, whatever its contact count -- a scene of a few ten thousand bodies is one
// wave per SIMD or less, where the tick costs the longest lane's chain once per LAUNCH: one launch beats two.
template <class T, bool EXT, int MINW, int NC, bool ALL = false>
__global__ __launch_bounds__(256, MINW) void step_contacts(T *S, T *So, int64_t n, StepParams<T> P, StepDiag *__restrict__ diag)
{
    static_assert(!ALL || NC == SC_MAXC, "the one-launch form holds every contact count");
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int my_contacts = 0;
    double my_resid = 0.0;
    if (P.bp_check && P.bp_flags[BPF_VIOLATION] != 0u) return;      // the chunk will be rolled back whole
    if (P.gate != nullptr && *P.gate == 0u) return;
    const int cnt = (i < n && !(P.skip != nullptr && P.skip[i])) ? P.scount[i] : -1;
    // (a body with more contacts than the buffer holds is flagged, and stepped with the first SC_MAXC of them: what a caller
    //  who has switched the collision proof off -- nobody reads the flag then -- gets, and says so in include/dmx_batch.h)
    const bool mine = ALL ? cnt >= 0 : NC == 4 ? (cnt >= 0 && cnt <= 4) : (cnt > 4);
    if (NC == 4 || ALL) {
        const unsigned long long over = __ballot(cnt > SC_MAXC), need8 = __ballot(!ALL && cnt > 4 && cnt <= SC_MAXC && !P.have8);
        if ((over | need8) != 0ull && (threadIdx.x & 63) == 0 && P.bp_flags != nullptr) {
            if (over != 0ull) atomicOr(&P.bp_flags[BPF_NOFAST], 1u);
            if (need8 != 0ull) atomicOr(&P.bp_flags[BPF_NEED8], 1u);
            atomicOr(&P.bp_flags[BPF_VIOLATION], 1u);
        }
    } else if (__ballot(mine) == 0ull) return;                      // nobody here has 5..8 contacts
    if (mine) {
        V3<T> x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
        if (P.bp_check) report_zone(zone_state(x.x - S[slab_ix(C_BPX, i)], x.z - S[slab_ix(C_BPZ, i)], S[slab_ix(C_BPSAFE, i)]), P.bp_flags);
        Q4<T> q = { S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)],
                    S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] };
        V3<T> v = { S[slab_ix(C_LVEL + 0, i)], S[slab_ix(C_LVEL + 1, i)], S[slab_ix(C_LVEL + 2, i)] };
        V3<T> w = { S[slab_ix(C_AVEL + 0, i)], S[slab_ix(C_AVEL + 1, i)], S[slab_ix(C_AVEL + 2, i)] };
        const T mass = S[slab_ix(C_MASS, i)];
        const V3<T> Ib = { S[slab_ix(C_INERTIA + 0, i)], S[slab_ix(C_INERTIA + 1, i)],
                           S[slab_ix(C_INERTIA + 2, i)] };
        V3<T> facc = { T(0), T(0), T(0) }, tacc = { T(0), T(0), T(0) };
        if (EXT) {
            facc = { S[slab_ix(C_FORCE + 0, i)], S[slab_ix(C_FORCE + 1, i)], S[slab_ix(C_FORCE + 2, i)] };
            tacc = { S[slab_ix(C_TORQUE + 0, i)], S[slab_ix(C_TORQUE + 1, i)], S[slab_ix(C_TORQUE + 2, i)] };
        }

        const T h = P.h;
        const M3<T> R = quat_to_R(q);
        const T invMass = T(1) / mass;
        const V3<T> invIb = { T(1) / Ib.x, T(1) / Ib.y, T(1) / Ib.z };
        facc.x = fma_(mass, P.g.x, facc.x); facc.y = fma_(mass, P.g.y, facc.y); facc.z = fma_(mass, P.g.z, facc.z);
        const M3<T> invIw = rotate_diag(R, invIb);
        if (P.gyro != 0 && !isotropic(Ib)) {
            const M3<T> Iw = rotate_diag(R, Ib);
            add_gyro_torque(tacc, Iw, w, h, P.gyro);
        }
        const int nc = cnt > SC_MAXC ? SC_MAXC : cnt;
        my_contacts = nc;
        int ncu = 0;     // largest contact count among the wave's lanes stepped here (wave-uniform by construction)
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (__ballot(nc > k) != 0ull) ncu = k + 1;

        if (ncu > 0) {
            constexpr int MAXR = 3 * NC;
            const int rpc = P.mu > 0 ? 3 : 1;
            const T hinv = T(1) / h;
            // v/h + M^-1 f
            const V3<T> tl = { fma_(facc.x, invMass, v.x * hinv), fma_(facc.y, invMass, v.y * hinv),
                               fma_(facc.z, invMass, v.z * hinv) };
            V3<T> ta = mulv(invIw, tacc);
            ta.x = fma_(w.x, hinv, ta.x); ta.y = fma_(w.y, hinv, ta.y); ta.z = fma_(w.z, hinv, ta.z);
            const T cfm = P.cfm * hinv;
            T rhs[MAXR], adcfm[MAXR], lam[MAXR];
            const T lo_f = -P.mu, hi_f = P.mu, hi_n = Limits<T>::inf();
            V3<T> Ja[MAXR], iMa[MAXR], Jl[MAXR], iMl[MAXR];      // J (angular, linear) times Ad, as J *= Ad leaves it; M^-1 J^T
#pragma unroll
            for (int r = 0; r < MAXR; r++) {      // rows of absent contacts stay zero
                rhs[r] = adcfm[r] = lam[r] = T(0);
                Ja[r] = { T(0), T(0), T(0) }; iMa[r] = { T(0), T(0), T(0) }; Jl[r] = { T(0), T(0), T(0) }; iMl[r] = { T(0), T(0), T(0) };
            }
#pragma unroll
            for (int k = 0; k < NC; k++) {
                if (k < ncu) {                    // (scalar branch; lanes with fewer contacts keep zero rows)
                    if (k < nc) {
                        const V3<T> cp = { P.sbuf[sc_ix(k, SC_POS + 0, i)], P.sbuf[sc_ix(k, SC_POS + 1, i)], P.sbuf[sc_ix(k, SC_POS + 2, i)] };
                        V3<T> dir[3];
                        dir[0] = { P.sbuf[sc_ix(k, SC_NORMAL + 0, i)], P.sbuf[sc_ix(k, SC_NORMAL + 1, i)], P.sbuf[sc_ix(k, SC_NORMAL + 2, i)] };
                        dir[1] = dir[2] = { T(0), T(0), T(0) };
                        if (rpc == 3) plane_space(dir[0], dir[1], dir[2]);
                        const V3<T> c1 = { cp.x - x.x, cp.y - x.y, cp.z - x.z };
#pragma unroll
                        for (int dnum = 0; dnum < 3; dnum++) {
                            const int r = 3 * k + dnum;
                            if (dnum < rpc) {
                                const V3<T> ja = cross(c1, dir[dnum]);
                                T c = T(0);
                                if (dnum == 0) {
                                    T depth = P.sbuf[sc_ix(k, SC_DEPTH, i)];
                                    if (depth < 0) depth = 0;
                                    c = (hinv * P.erp) * depth;
                                    if (P.surf_mode & SURF_BOUNCE) {
                                        const T outgoing = dot(dir[0], v) + dot(ja, w);
                                        if (P.bounce_vel >= 0 && (-outgoing) > P.bounce_vel) {
                                            const T newc = -P.bounce * outgoing;
                                            if (newc > c) c = newc;
                                        }
                                    }
                                }
                                T sum = dir[dnum].x * tl.x;
                                sum = fma_(dir[dnum].y, tl.y, sum); sum = fma_(dir[dnum].z, tl.z, sum);
                                sum = fma_(ja.x, ta.x, sum); sum = fma_(ja.y, ta.y, sum); sum = fma_(ja.z, ta.z, sum);
                                const T b = fma_(c, hinv, -sum);
                                const V3<T> iml = { invMass * dir[dnum].x, invMass * dir[dnum].y, invMass * dir[dnum].z };
                                const V3<T> ima = mulv(invIw, ja);
                                T s2 = iml.x * dir[dnum].x;
                                s2 = fma_(iml.y, dir[dnum].y, s2); s2 = fma_(iml.z, dir[dnum].z, s2);
                                s2 = fma_(ima.x, ja.x, s2); s2 = fma_(ima.y, ja.y, s2); s2 = fma_(ima.z, ja.z, s2);
                                const T ad = P.sor_w / (s2 + cfm);
                                Ja[r] = { ja.x * ad, ja.y * ad, ja.z * ad };
                                Jl[r] = { dir[dnum].x * ad, dir[dnum].y * ad, dir[dnum].z * ad };
                                iMa[r] = ima; iMl[r] = iml;
                                rhs[r] = b * ad;
                                adcfm[r] = ad * cfm;
                            }
                        }
                    }
                }
            }

            // ---- SOR-PGS: lambda = 0 start, rows in creation order; branch-free row update as in step_plane ----
            V3<T> fl = { T(0), T(0), T(0) }, fa = { T(0), T(0), T(0) };
            T rsum = T(0);
            // (FULL: as in step_plane -- every contact slot taken in every lane stepped here, friction rows present)
            auto sweep = [&](auto FAST, auto LAST, auto FULL) {
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    if (decltype(FULL)::value || k < ncu) {
#pragma unroll
                        for (int dnum = 0; dnum < 3; dnum++) {
                            const int r = 3 * k + dnum;
                            if (decltype(FULL)::value || dnum < rpc) {
                                const T old = lam[r];
                                T delta = fma_(-old, adcfm[r], rhs[r]);
                                delta -= fma_(fa.z, Ja[r].z, fma_(fa.y, Ja[r].y, fma_(fa.x, Ja[r].x,
                                         fma_(fl.z, Jl[r].z, fma_(fl.y, Jl[r].y, fl.x * Jl[r].x)))));
                                const T nl = old + delta;
                                T nlam = nl;
                                if (dnum == 0 || !decltype(FAST)::value) {
                                    const T lo = dnum == 0 ? T(0) : lo_f, hi = dnum == 0 ? hi_n : hi_f;
                                    const bool below = nl < lo, above = nl > hi;
                                    nlam = below ? lo : (above ? hi : nl);
                                    delta = below ? lo - old : (above ? hi - old : delta);
                                }
                                lam[r] = nlam;
                                fl.x = fma_(delta, iMl[r].x, fl.x); fl.y = fma_(delta, iMl[r].y, fl.y);
                                fl.z = fma_(delta, iMl[r].z, fl.z);
                                fa.x = fma_(delta, iMa[r].x, fa.x); fa.y = fma_(delta, iMa[r].y, fa.y);
                                fa.z = fma_(delta, iMa[r].z, fa.z);
                                if (decltype(LAST)::value) rsum += tabs(delta);
                            }
                        }
                    }
                }
            };
            using std::true_type;
            using std::false_type;
            // FAST: friction rows are unbounded (mu = inf, the reference's surface, main.c:687): no friction clamp.  Lanes with
            // fewer contacts than the wave's count need no masking either way: their surplus rows are all zero, a zero row's
            // delta is exactly zero and leaves lambda and the accumulators as they are.
            const bool fast = !(P.mu < Limits<T>::inf());   // wave-uniform
            if (fast && ncu == NC && rpc == 3) {
                for (int it = 0; it + 1 < P.iters; it++) sweep(true_type{}, false_type{}, true_type{});
                if (P.iters > 0) sweep(true_type{}, true_type{}, true_type{});
            } else if (fast) {
                for (int it = 0; it + 1 < P.iters; it++) sweep(true_type{}, false_type{}, false_type{});
                if (P.iters > 0) sweep(true_type{}, true_type{}, false_type{});
            } else {
                for (int it = 0; it + 1 < P.iters; it++) sweep(false_type{}, false_type{}, false_type{});
                if (P.iters > 0) sweep(false_type{}, true_type{}, false_type{});
            }
            my_resid = (double)rsum;
            if (nc > 0) {        // v += h * (M^-1 J^T lambda)
                v.x = fma_(h, fl.x, v.x); v.y = fma_(h, fl.y, v.y); v.z = fma_(h, fl.z, v.z);
                w.x = fma_(h, fa.x, w.x); w.y = fma_(h, fa.y, w.y); w.z = fma_(h, fa.z, w.z);
            }
        }

        // ---- v += h M^-1 f_ext ; integrate ----------------------------------------------------
        const T hm = h * invMass;
        v.x = fma_(hm, facc.x, v.x); v.y = fma_(hm, facc.y, v.y); v.z = fma_(hm, facc.z, v.z);
        tacc.x *= h; tacc.y *= h; tacc.z *= h;
        const V3<T> dw = mulv(invIw, tacc);
        w.x += dw.x; w.y += dw.y; w.z += dw.z;
        x.x = fma_(h, v.x, x.x); x.y = fma_(h, v.y, x.y); x.z = fma_(h, v.z, x.z);
        integrate_quat(q, w, h);
        pack_boundary(P, i, x, q, v, w);

        So[slab_ix(C_POS + 0, i)] = x.x; So[slab_ix(C_POS + 1, i)] = x.y; So[slab_ix(C_POS + 2, i)] = x.z;
        So[slab_ix(C_QUAT + 0, i)] = q.w; So[slab_ix(C_QUAT + 1, i)] = q.x;
        So[slab_ix(C_QUAT + 2, i)] = q.y; So[slab_ix(C_QUAT + 3, i)] = q.z;
        So[slab_ix(C_LVEL + 0, i)] = v.x; So[slab_ix(C_LVEL + 1, i)] = v.y; So[slab_ix(C_LVEL + 2, i)] = v.z;
        So[slab_ix(C_AVEL + 0, i)] = w.x; So[slab_ix(C_AVEL + 1, i)] = w.y; So[slab_ix(C_AVEL + 2, i)] = w.z;
        if (EXT) {
#pragma unroll
            for (int k = 0; k < 6; k++) S[slab_ix(C_FORCE + k, i)] = T(0);
        }
    }
    // ---- diagnostics: one slot per wave; the NC = 4 launch writes it, the NC = 8 launch behind it adds its lanes' share
    const int wc = wave_sum<int>(my_contacts);
    const double wr = wave_sum<double>(my_resid);
    if ((threadIdx.x & 63) == 0 && i < n) {
        StepDiag *d = &diag[(blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6];
        if (NC == 4 || ALL) { d->contacts = (unsigned long long)wc; d->residual = wr; }
        else { d->contacts += (unsigned long long)wc; d->residual += wr; }
    }
}

// ---------------------------------------------------------------------------------------------
// pack_transforms: GetTransformMat (main.c:602-622) per body: column-major 4x4 from pos + R(q).
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void pack_transforms(const T *__restrict__ S, int64_t stride, int64_t first,
                                                       int64_t count, T *__restrict__ out)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int64_t i = first + t;
    const Q4<T> q = { S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)],
                      S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] };
    const M3<T> R = quat_to_R(q);
    T *o = out + 16 * t;
    o[0] = R.m[0][0]; o[1] = R.m[1][0]; o[2] = R.m[2][0]; o[3] = T(0);
    o[4] = R.m[0][1]; o[5] = R.m[1][1]; o[6] = R.m[2][1]; o[7] = T(0);
    o[8] = R.m[0][2]; o[9] = R.m[1][2]; o[10] = R.m[2][2]; o[11] = T(0);
    o[12] = S[slab_ix(C_POS + 0, i)]; o[13] = S[slab_ix(C_POS + 1, i)];
    o[14] = S[slab_ix(C_POS + 2, i)]; o[15] = T(1);
}

// gather / scatter the 13 state reals of listed bodies (boundary exchange between GPUs)
// safe-zone test only, for slots the step kernels do not own (ghosts of a neighbour rank's boundary bodies)
template <class T>
__global__ __launch_bounds__(256) void check_zones(const T *__restrict__ S, int64_t first, int64_t count, uint32_t *flags)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int zs = 0;
    if (t < count) {
        const int64_t i = first + t;
        zs = zone_state(S[slab_ix(C_POS + 0, i)] - S[slab_ix(C_BPX, i)], S[slab_ix(C_POS + 2, i)] - S[slab_ix(C_BPZ, i)],
                        S[slab_ix(C_BPSAFE, i)]);
    }
    report_zone(zs, flags);
}

// Ghost refresh of the multi-GPU exchange, one launch per tick: the lower neighbour's rows (AoS, 13 reals per body) go
// to slots [first, first+count_lo), the upper neighbour's to the count_hi slots behind them; a null source leaves its
// range alone (no neighbour on that side).  With `check` the new (x,z) is tested against the slot's safe zone.
template <class T>
__global__ __launch_bounds__(256) void refresh_ghosts(T *__restrict__ S, int64_t first, int64_t count_lo,
                                                      const T *__restrict__ src_lo, int64_t count_hi,
                                                      const T *__restrict__ src_hi, int check, uint32_t *flags)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int zs = 0;
    if (t < count_lo + count_hi) {
        const T *src = t < count_lo ? src_lo : src_hi;
        if (src != nullptr) {
            const T *p = src + (t < count_lo ? t : t - count_lo) * C_MASS;
            const int64_t i = first + t;
#pragma unroll
            for (int c = 0; c < C_MASS; c++) S[slab_ix(c, i)] = p[c];
            if (check) zs = zone_state(p[0] - S[slab_ix(C_BPX, i)], p[2] - S[slab_ix(C_BPZ, i)], S[slab_ix(C_BPSAFE, i)]);
        }
    }
    if (check) report_zone(zs, flags);
}

// fill component c of every allocated body (pad included) with one value
template <class T>
__global__ __launch_bounds__(256) void fill_component(T *__restrict__ S, int c, T value, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) S[slab_ix(c, i)] = value;
}

// Rollback snapshot of the 13 state components (the first C_MASS reals x SLAB_TILE of every tile are one
// contiguous run): `packed` holds them tile after tile.  save: slab -> packed, else packed -> slab.
template <class T>
__global__ __launch_bounds__(256) void copy_state(T *__restrict__ S, T *__restrict__ packed, int64_t n_elems, bool save)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_elems) return;
    constexpr int64_t run = (int64_t)C_MASS * SLAB_TILE;
    const int64_t s = (t / run) * (int64_t)(C_COUNT * SLAB_TILE) + t % run;
    if (save) packed[t] = S[s];
    else      S[s] = packed[t];
}

// components [c0, c0+k) of bodies [first, first+count), slab to slab (the batch keeps two slabs of one layout)
template <class T>
__global__ __launch_bounds__(256) void copy_components(const T *__restrict__ from, T *__restrict__ to, int c0, int k,
                                                       int64_t first, int64_t count)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= count * k) return;
    const int c = (int)(t / count);
    const int64_t ix = slab_ix(c0 + c, first + (t - (int64_t)c * count));
    to[ix] = from[ix];
}

template <class T>
__global__ __launch_bounds__(256) void gather_bodies(const T *__restrict__ S, int64_t stride,
                                                     const int32_t *__restrict__ idx, int64_t count,
                                                     T *__restrict__ out)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= count * C_MASS) return;
    const int64_t b = t / C_MASS;
    const int c = (int)(t - b * C_MASS);
    out[t] = S[slab_ix(c, idx[b])];
}
template <class T>
__global__ __launch_bounds__(256) void scatter_bodies(T *__restrict__ S, int64_t stride,
                                                      const int32_t *__restrict__ idx, int64_t count,
                                                      const T *__restrict__ in)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= count * C_MASS) return;
    const int64_t b = t / C_MASS;
    const int c = (int)(t - b * C_MASS);
    S[slab_ix(c, idx[b])] = in[t];
}

// host-order rows (n x k, array of structs) <-> the slab's component tiles, used by upload/download through a staging buffer
template <class T>
__global__ __launch_bounds__(256) void aos_to_soa(T *__restrict__ S, int64_t stride, int comp0, int k,
                                                  int64_t first, int64_t count, const T *__restrict__ aos)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= count * k) return;
    const int64_t b = t / k;
    const int c = (int)(t - b * k);
    S[slab_ix(comp0 + c, first + b)] = aos[t];
}
template <class T>
__global__ __launch_bounds__(256) void soa_to_aos(const T *__restrict__ S, int64_t stride, int comp0, int k,
                                                  int64_t first, int64_t count, T *__restrict__ aos)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= count * k) return;
    const int64_t b = t / k;
    const int c = (int)(t - b * k);
    aos[t] = S[slab_ix(comp0 + c, first + b)];
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }
constexpr int64_t kOneLaunchBodies = 256 * 4 * 64;

template <class T>
hipError_t launch_step(T *S, T *So, const uint8_t *gtype, int64_t stride, int64_t n, const StepParams<T> &P, bool ext,
                       StepDiag *diag, hipStream_t st)
{
    if (P.n_static > 0 && P.sbuf != nullptr) {
        // bodies at static geometry: narrowphase against the plane and the static boxes, then the fused solve + integrate
        const hipError_t e = launch_np_static<T>(S, gtype, n, P, st);
        if (e != hipSuccess) return e;
        const unsigned grid = blocks_for(n, 256);
        // up to one wave per SIMD of the chip (256 CUs x 4 SIMDs x 64 lanes): the tick is the longest lane's chain per launch
        if (P.have8 && n <= kOneLaunchBodies && sizeof(T) == 4) {
            if (ext) hipLaunchKernelGGL((step_contacts<T, true, 1, 8, true>), dim3(grid), dim3(256), 0, st, S, So, n, P, diag);
            else     hipLaunchKernelGGL((step_contacts<T, false, 1, 8, true>), dim3(grid), dim3(256), 0, st, S, So, n, P, diag);
            return hipGetLastError();
        }
        if (ext) hipLaunchKernelGGL((step_contacts<T, true, 1, 4>), dim3(grid), dim3(256), 0, st, S, So, n, P, diag);
        else if (sizeof(T) == 4) hipLaunchKernelGGL((step_contacts<T, false, 2, 4>), dim3(grid), dim3(256), 0, st, S, So, n, P, diag);
        else hipLaunchKernelGGL((step_contacts<T, false, 1, 4>), dim3(grid), dim3(256), 0, st, S, So, n, P, diag);
        if (P.have8) {
            if (ext) hipLaunchKernelGGL((step_contacts<T, true, 1, 8>), dim3(grid), dim3(256), 0, st, S, So, n, P, diag);
            else     hipLaunchKernelGGL((step_contacts<T, false, 1, 8>), dim3(grid), dim3(256), 0, st, S, So, n, P, diag);
        }
    } else if (!P.plane_on) {
        constexpr int VMAX = 16 / sizeof(T);
        // default: one body per lane.  On the tiled slab a wave's 17 loads already cover one contiguous run, so wider
        // per-lane loads buy nothing, and V = 1 keeps the kernel at 67 VGPRs (7 waves/SIMD): measured 21.1 / 22.2 / 22.4 us
        // per tick for V = 1 / 2 / 4 at 1 Mi f32 bodies (profiles/r01_integrate_free_tiled_sweep.txt).
        const int VDEF = 1;
        const int V = P.skip != nullptr ? 1 : (P.vec == 1 || P.vec == 2 || P.vec == VMAX) ? P.vec : VDEF;
        const int64_t nvec = (n + V - 1) / V;     // pad bodies up to `stride` are valid memory
        const unsigned grid = blocks_for(nvec, 256);
        if (P.vec <= -32 && P.ticks == 1 && !ext && P.skip == nullptr) {      // DMX_WIDE=2[:blocks per CU]: reads by LDS-DMA, persistent grid
            const int64_t ntiles = (n + SLAB_TILE - 1) / SLAB_TILE;
            const int per_cu = (-P.vec) / 32 > 0 ? ((-P.vec) % 32 == 0 ? 4 : (-P.vec) % 32) : 4;
            const size_t lds = (size_t)4 * 2 * C_SIDES * SLAB_TILE * sizeof(T);
            const hipError_t ea = hipFuncSetAttribute((const void *)integrate_free_dma<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) return ea;
            int64_t blocks = (ntiles + 3) / 4;
            if (blocks > (int64_t)256 * per_cu) blocks = (int64_t)256 * per_cu;
            hipLaunchKernelGGL((integrate_free_dma<T>), dim3((unsigned)blocks), dim3(256), lds, st, S, So, ntiles, P);
            return hipGetLastError();
        }
        if (P.vec == -16 && P.ticks == 1 && !ext && P.skip == nullptr) {      // DMX_WIDE=1: the tile in 16-byte pieces through LDS
            const int64_t ntiles = (n + SLAB_TILE - 1) / SLAB_TILE;
            hipLaunchKernelGGL((integrate_free_wide<T>), dim3(blocks_for(ntiles, 4)), dim3(256), 0, st, S, So, ntiles, P);
            return hipGetLastError();
        }
#define DMX_LAUNCH_FREE(VV, MW)                                                                                      \
    do {                                                                                                             \
        if (P.ticks > 1) hipLaunchKernelGGL((integrate_free<T, VV, false, MW, true>), dim3(grid), dim3(256), 0, st, S, So, stride, nvec, P);   \
        else if (ext) hipLaunchKernelGGL((integrate_free<T, VV, true, MW, false>), dim3(grid), dim3(256), 0, st, S, So, stride, nvec, P);  \
        else     hipLaunchKernelGGL((integrate_free<T, VV, false, MW, false>), dim3(grid), dim3(256), 0, st, S, So, stride, nvec, P); \
    } while (0)
        const int mw = P.min_waves;   // launch tuning: minimum waves per SIMD the register allocator must leave room for
        if (V == 1) { if (mw == 8) DMX_LAUNCH_FREE(1, 8); else if (mw == 6) DMX_LAUNCH_FREE(1, 6); else DMX_LAUNCH_FREE(1, 1); }
        else if (V == 2) { if (mw == 4) DMX_LAUNCH_FREE(2, 4); else if (mw == 5) DMX_LAUNCH_FREE(2, 5); else if (mw == 6) DMX_LAUNCH_FREE(2, 6); else DMX_LAUNCH_FREE(2, 1); }
        else { if (mw == 4) DMX_LAUNCH_FREE(VMAX, 4); else if (mw == 3) DMX_LAUNCH_FREE(VMAX, 3); else DMX_LAUNCH_FREE(VMAX, 1); }
#undef DMX_LAUNCH_FREE
    } else {
        const unsigned grid = blocks_for(n, 256);
#define DMX_LAUNCH_PLANE(MW)                                                                                           \
    do {                                                                                                                   \
        if (convex) {                                                                                                      \
            if (ext) hipLaunchKernelGGL((step_plane<T, true, 1, CONVEX_MAXC>), dim3(grid), dim3(256), 0, st, S, So, gtype, stride, n, P, diag);  \
            else     hipLaunchKernelGGL((step_plane<T, false, 1, CONVEX_MAXC>), dim3(grid), dim3(256), 0, st, S, So, gtype, stride, n, P, diag); \
        } else if (ext) hipLaunchKernelGGL((step_plane<T, true, MW, 4>), dim3(grid), dim3(256), 0, st, S, So, gtype, stride, n, P, diag);  \
        else     hipLaunchKernelGGL((step_plane<T, false, MW, 4>), dim3(grid), dim3(256), 0, st, S, So, gtype, stride, n, P, diag); \
    } while (0)
        // convex bodies: their plane contacts first (one wavefront per body), then the fused step with 8 contact slots
        const bool convex = P.hull_n > 0 && P.cbuf != nullptr;
        if (convex) {
            const hipError_t e = launch_np_convex_plane<T>(S, gtype, n, P, st);
            if (e != hipSuccess) return e;
        }
        switch (P.min_waves) {
        case 1: DMX_LAUNCH_PLANE(1); break;
        case 2: DMX_LAUNCH_PLANE(2); break;
        default:
            if (sizeof(T) == 4) DMX_LAUNCH_PLANE(2); else DMX_LAUNCH_PLANE(1);
        }
#undef DMX_LAUNCH_PLANE
    }
    return hipGetLastError();
}

template <class T>
hipError_t launch_pack_transforms(const T *S, int64_t stride, int64_t first, int64_t count, T *out, hipStream_t st)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL((pack_transforms<T>), dim3(blocks_for(count, 256)), dim3(256), 0, st, S, stride, first, count, out);
    return hipGetLastError();
}
template <class T>
hipError_t launch_check_zones(const T *S, int64_t first, int64_t count, uint32_t *flags, hipStream_t st)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL((check_zones<T>), dim3(blocks_for(count, 256)), dim3(256), 0, st, S, first, count, flags);
    return hipGetLastError();
}

template <class T>
hipError_t launch_refresh_ghosts(T *S, int64_t first, int64_t count_lo, const T *src_lo, int64_t count_hi, const T *src_hi,
                                 int check, uint32_t *flags, hipStream_t st)
{
    if (count_lo + count_hi <= 0) return hipSuccess;
    hipLaunchKernelGGL((refresh_ghosts<T>), dim3(blocks_for(count_lo + count_hi, 256)), dim3(256), 0, st, S, first, count_lo,
                       src_lo, count_hi, src_hi, check, flags);
    return hipGetLastError();
}

template <class T>
hipError_t launch_fill_component(T *S, int c, T value, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL((fill_component<T>), dim3(blocks_for(n, 256)), dim3(256), 0, st, S, c, value, n);
    return hipGetLastError();
}

template <class T>
hipError_t launch_copy_state(T *S, T *packed, int64_t n_bodies, bool save, hipStream_t st)
{
    const int64_t n_elems = (n_bodies + SLAB_TILE - 1) / SLAB_TILE * C_MASS * SLAB_TILE;
    hipLaunchKernelGGL((copy_state<T>), dim3(blocks_for(n_elems, 256)), dim3(256), 0, st, S, packed, n_elems, save);
    return hipGetLastError();
}

template <class T>
hipError_t launch_copy_components(const T *from, T *to, int c0, int k, int64_t first, int64_t count, hipStream_t st)
{
    if (count <= 0 || k <= 0) return hipSuccess;
    hipLaunchKernelGGL((copy_components<T>), dim3(blocks_for(count * k, 256)), dim3(256), 0, st, from, to, c0, k, first, count);
    return hipGetLastError();
}

template <class T>
hipError_t launch_gather(const T *S, int64_t stride, const int32_t *idx, int64_t count, T *out, hipStream_t st)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL((gather_bodies<T>), dim3(blocks_for(count * C_MASS, 256)), dim3(256), 0, st, S, stride, idx, count, out);
    return hipGetLastError();
}
template <class T>
hipError_t launch_scatter(T *S, int64_t stride, const int32_t *idx, int64_t count, const T *in, hipStream_t st)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL((scatter_bodies<T>), dim3(blocks_for(count * C_MASS, 256)), dim3(256), 0, st, S, stride, idx, count, in);
    return hipGetLastError();
}
template <class T>
hipError_t launch_aos_to_soa(T *S, int64_t stride, int comp0, int k, int64_t first, int64_t count, const T *aos,
                             hipStream_t st)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL((aos_to_soa<T>), dim3(blocks_for(count * k, 256)), dim3(256), 0, st, S, stride, comp0, k, first, count, aos);
    return hipGetLastError();
}
template <class T>
hipError_t launch_soa_to_aos(const T *S, int64_t stride, int comp0, int k, int64_t first, int64_t count, T *aos,
                             hipStream_t st)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL((soa_to_aos<T>), dim3(blocks_for(count * k, 256)), dim3(256), 0, st, S, stride, comp0, k, first, count, aos);
    return hipGetLastError();
}

#define DMX_INSTANTIATE(T)                                                                                         \
    template hipError_t launch_step<T>(T *, T *, const uint8_t *, int64_t, int64_t, const StepParams<T> &, bool,   \
                                       StepDiag *, hipStream_t);                                                   \
    template hipError_t launch_pack_transforms<T>(const T *, int64_t, int64_t, int64_t, T *, hipStream_t);         \
    template hipError_t launch_gather<T>(const T *, int64_t, const int32_t *, int64_t, T *, hipStream_t);          \
    template hipError_t launch_scatter<T>(T *, int64_t, const int32_t *, int64_t, const T *, hipStream_t);         \
    template hipError_t launch_aos_to_soa<T>(T *, int64_t, int, int, int64_t, int64_t, const T *, hipStream_t);    \
    template hipError_t launch_soa_to_aos<T>(const T *, int64_t, int, int, int64_t, int64_t, T *, hipStream_t);     \
    template hipError_t launch_fill_component<T>(T *, int, T, int64_t, hipStream_t);                                \
    template hipError_t launch_copy_state<T>(T *, T *, int64_t, bool, hipStream_t);                                 \
    template hipError_t launch_copy_components<T>(const T *, T *, int, int, int64_t, int64_t, hipStream_t);         \
    template hipError_t launch_check_zones<T>(const T *, int64_t, int64_t, uint32_t *, hipStream_t);               \
    template hipError_t launch_refresh_ghosts<T>(T *, int64_t, int64_t, const T *, int64_t, const T *, int, uint32_t *, hipStream_t);
DMX_INSTANTIATE(float)
DMX_INSTANTIATE(double)

// HIP loads a translation unit's code object at the first launch of one of its kernels -- a couple of milliseconds each, which an
// interactive caller would meet as a hitch at the first tick that needs the exact pipeline.  dmxBatchCreate asks for one
// kernel's attributes per unit instead (dmx_preload_code, dmx_batch.cpp): the load happens there.
hipError_t dmx_touch_kernels(int real_bytes)
{
    // (the unit's code object, and -- what costs more -- each kernel's own first-use set-up: every kernel an exact tick or a fused
    //  tick may launch, in the batch's precision)
    hipFuncAttributes a;
    hipError_t e = hipSuccess;
    auto touch = [&](const void *k) { const hipError_t r = hipFuncGetAttributes(&a, k); if (r != hipSuccess) e = r; };
    if (real_bytes == 4) {
        touch((const void *)&integrate_free<float, 1, false, 1, false>);
        touch((const void *)&step_plane<float, false, 1, 4>);
        touch((const void *)&step_plane<float, false, 2, 4>);
        touch((const void *)&step_contacts<float, false, 1, 8, true>);
        touch((const void *)&check_zones<float>);
        touch((const void *)&copy_components<float>);
    } else {
        touch((const void *)&integrate_free<double, 1, false, 1, false>);
        touch((const void *)&step_plane<double, false, 1, 4>);
        touch((const void *)&step_plane<double, false, 2, 4>);
        touch((const void *)&step_contacts<double, false, 1, 8, true>);
        touch((const void *)&check_zones<double>);
        touch((const void *)&copy_components<double>);
    }
    return e;
}

}  // namespace dmx
