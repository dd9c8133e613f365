// dmx_exact.hip -- the exact (pair-bearing) tick's bookkeeping, on the device.
//
// What dSpaceCollide + NearCallback + the island builder of dWorldStep do between them
// (/root/reference/src/main.c:211-215, 674-693) for the bodies that are in body-body pairs this tick:
//   body pairs in canonical order (ascending i, then j) -> the bodies involved, ascending -> connected components
//   (dynamics islands, numbered by their lowest slot) -> narrowphase contacts -> the tick's contact joints in creation
//   order (ground-plane contacts by body, then pair contacts by pair), grouped by island -> for islands that get a
//   workgroup, the level schedule of their rows (row level = 1 + latest level of an earlier row sharing a body).
// Integer / index work over a few arrays: coalesced loads, lock-free union-find (atomicCAS hooking, larger root under
// smaller), rocPRIM scans and one stable radix sort of (island, entry) keys; no MFMA, nothing to stage in LDS.
//
// Sizes live on the device (ExactCounts); every kernel loops over the device-side count with a grid sized from the
// host's capacity estimate, and arrays are padded up to that capacity with neutral entries, so the host never needs a
// count to enqueue the pipeline.  It reads ExactCounts back ONCE per tick (capacity check + launch shape of the island
// solve); a count above its capacity makes the host grow the estimate and run the pipeline again -- nothing has
// touched the state by then.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/block/block_radix_sort.hpp>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"
#include "dmx_collide.hpp"
#include "dmx_collide_wave.hpp"
#include "dmx_grid.hpp"
#include "dmx_exact.hpp"

namespace dmx {

namespace {

__device__ __forceinline__ uint32_t lo32(uint64_t v) { return (uint32_t)v; }
__device__ __forceinline__ uint32_t hi32(uint64_t v) { return (uint32_t)(v >> 32); }

// does the body's AABB overlap static box b's?
template <class T> __device__ __forceinline__ bool rec_meets_static(const GridRec<T> &r, const T *b)
{
    return !(r.lo[0] > b[SBOX_HI + 0] || b[SBOX_LO + 0] > r.hi[0] || r.lo[1] > b[SBOX_HI + 1] || b[SBOX_LO + 1] > r.hi[1] ||
             r.lo[2] > b[SBOX_HI + 2] || b[SBOX_LO + 2] > r.hi[2]);
}

// Walk the 3x3 columns around body i and call f(j) for every other body whose AABB overlaps i's (each once).
// Dependent accesses are what this costs (a thread's chain is the kernel's duration in a small scene): a candidate's column
// and AABB come as one record, a bucket's items four at a time, and the next column's count and first four items are on
// their way while this column's candidates are tested.  The column loop stays rolled: straight-line code that runs once per
// launch is paid for in instruction fetches.
template <class T, class F>
__device__ __forceinline__ void for_each_partner(const T *S, const uint8_t *gtype, int64_t i, const GridParams<T> &G, F f)
{
    const GridRec<T> me = G.rec[i];
    const int gti = gtype[i];
    auto test4 = [&](const int4 &it, uint32_t s0, uint32_t n, int cx, int cz) {
        const int32_t js[4] = { it.x, it.y, it.z, it.w };
        GridRec<T> o[4];
        int gtj[4];
        bool live[4];
        // (fetching all four records unconditionally was tried: in a sparse scene three of four slots are dead, and one
        //  compute unit's memory pipeline -- which is all a one-workgroup launch has -- is the bound, not the latency)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            live[q] = s0 + q < n && js[q] != (int32_t)i;
            if (live[q]) { o[q] = G.rec[js[q]]; gtj[q] = gtype[js[q]]; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (!live[q]) continue;
            const int64_t j = js[q];
            // hashed buckets can mix columns: keep only true 3x3 neighbours so (i,j) is met in one cell only
            if (o[q].ix != cx || o[q].iz != cz) continue;
            if (!classes_collide(gti, gtj[q], G.class_pairs)) continue;      // switched off by the caller (dmxBatchSetClassPairs): not a pair
            if (o[q].lo[0] > me.hi[0] || me.lo[0] > o[q].hi[0] || o[q].lo[1] > me.hi[1] || me.lo[1] > o[q].hi[1] ||
                o[q].lo[2] > me.hi[2] || me.lo[2] > o[q].hi[2])
                continue;
            f(j);
        }
    };
    // (bucket capacities are multiples of 4, so a bucket's items are 16-byte aligned and its first four always exist)
    uint32_t h = cell_hash(me.ix - 1, me.iz - 1, G.mask, G.xbits);
    uint32_t cnt = G.count[h];
    int4 it = *reinterpret_cast<const int4 *>(G.items + (size_t)h * G.cap);
#pragma unroll 1
    for (int c = 0; c < 9; c++) {
        const int cx = me.ix + (c % 3) - 1, cz = me.iz + (c / 3) - 1;
        const uint32_t h0 = h, n = cnt > (uint32_t)G.cap ? (uint32_t)G.cap : cnt;
        const int4 it0 = it;
        if (c < 8) {
            h = cell_hash(me.ix + ((c + 1) % 3) - 1, me.iz + ((c + 1) / 3) - 1, G.mask, G.xbits);
            cnt = G.count[h];
            it = *reinterpret_cast<const int4 *>(G.items + (size_t)h * G.cap);
        }
        if (n > 0) test4(it0, 0, n, cx, cz);
        for (uint32_t s0 = 4; s0 < n; s0 += 4)
            test4(*reinterpret_cast<const int4 *>(G.items + (size_t)h0 * G.cap + s0), s0, n, cx, cz);
    }
}

// the walk above as a callable, for the stage functions below
template <class T> struct GridWalk {
    const T *S; const uint8_t *gtype; const GridParams<T> &G;
    template <class F> __device__ __forceinline__ void operator()(int64_t i, F f) const { for_each_partner<T>(S, gtype, i, G, f); }
};
// ... and the same walk over a grid that lives in LDS (ex_small_front: the scene of a one-workgroup launch fits there).  Cells
// are runs of ONE array of records sorted by cell (start[h] .. start[h + 1]); a record carries the body's AABB, its slot and class
// and the low halves of its column -- everything a candidate is tested on -- so the walk's reads do not depend on one another:
// the nine cells' bounds are fetched together, then the candidates two at a time.  (The bucket walk in device memory is a chain of
// dependent L2 round trips -- count, items, record -- and was the kernel's longest stage.)  The order in which a body's partners
// come up differs from the bucket walk's; both callers are indifferent to it (a count; a run that is sorted).
template <class T> struct alignas(16) CellRec { T lo[3], hi[3]; uint32_t id, key; };      // id = slot | class << 16; key = (ix & 0xffff) | iz << 16
template <class T> struct LdsGridWalk {
    const GridRec<T> *rec_g;         // the body's own record comes from device memory (one read, before the walk)
    const CellRec<T> *cr; const uint32_t *start; const uint8_t *gtype;
    uint32_t mask; int xbits; uint32_t class_pairs;
    // (sub, K): this caller is lane `sub` of K that share body i -- it takes every K-th pair of candidates of every run
    template <class F> __device__ __forceinline__ void operator()(int64_t i, F f, uint32_t sub = 0, uint32_t K = 1) const
    {
        const GridRec<T> me = rec_g[i];
        const int gti = gtype[i];
        uint32_t a[9], e[9];
#pragma unroll
        for (int c = 0; c < 9; c++) {
            const uint32_t h = cell_hash(me.ix + (c % 3) - 1, me.iz + (c / 3) - 1, mask, xbits);
            a[c] = start[h]; e[c] = start[h + 1];
        }
        auto test = [&](const CellRec<T> &o, uint32_t key) {
            const int64_t j = (int64_t)(o.id & 0xffffu);
            if (j == i || o.key != key) return;                  // buckets can mix columns: (i, j) is met in one cell only
            if (!classes_collide(gti, (int)(o.id >> 16), class_pairs)) return;
            if (o.lo[0] > me.hi[0] || me.lo[0] > o.hi[0] || o.lo[1] > me.hi[1] || me.lo[1] > o.hi[1] ||
                o.lo[2] > me.hi[2] || me.lo[2] > o.hi[2])
                return;
            f(j);
        };
#pragma unroll
        for (int c = 0; c < 9; c++) {
            const uint32_t key = ((uint32_t)(me.ix + (c % 3) - 1) & 0xffffu) | ((uint32_t)(me.iz + (c / 3) - 1) << 16);
            for (uint32_t k = a[c] + 2 * sub; k < e[c]; k += 2 * K) {
                const bool two = k + 1 < e[c];
                const CellRec<T> o0 = cr[k], o1 = cr[two ? k + 1 : k];
                test(o0, key);
                if (two) test(o1, key);
            }
        }
    }
};
constexpr int EXS_PARTNERS = 8;        // partners above it that a body's count pass leaves in LDS for the write pass (more: that body walks again)

// ---- 1. per active body: partners above it (the pairs it owns) and whether it is in any pair at all ---------------
// pc[i] = (owned pairs << 32) | in-any-pair; inpair[i] = in-any-pair (the fused kernel's skip mask).  A partner in a ghost
// slot means an island spanning two ranks.
template <class T, class W, class ST = uint16_t, int NST = EXS_PARTNERS>
__device__ __forceinline__ void st_pair_count(const W &walk, const uint8_t *gtype, int64_t n_active, const GridParams<T> &G, uint64_t *pc,
                                              uint8_t *inpair, ExactCounts *C, int32_t *cross_list, int64_t first, int64_t step,
                                              ST *staged = nullptr)
{
    for (int64_t i = first; i < n_active; i += step) {
        uint32_t owned = 0, any = 0;
        if (gtype[i] != GEOM_NONE) {
            walk(i, [&](int64_t j) {
                any = 1;
                if (j >= n_active) {
                    if (atomicOr(&C->cross, 1u) == 0u) { C->cross_a = (uint32_t)i; C->cross_b = (uint32_t)j; }
                    const uint32_t at = atomicAdd(&C->ncross, 1u);
                    if (at < EX_CROSS_CAP) { cross_list[2 * at] = (int32_t)i; cross_list[2 * at + 1] = (int32_t)j; }
                }
                else if (j > i) {
                    // (staged: the first EXS_PARTNERS of them stay in LDS, so the write pass need not walk again)
                    if (staged != nullptr && owned < (uint32_t)NST) staged[(size_t)i * NST + owned] = (ST)j;
                    owned++;
                }
            });
            // static box geoms are "big geoms against everyone".  A body whose AABB overlaps static boxes but no other body's
            // is a one-body island: the fused path (np_static -> step_contacts) steps it -- unless its contacts might not fit
            // that path's buffer of SC_MAXC: AABB over two or more static boxes, or over one with a ground plane present
            // (<= 8 contacts per geom pair).  Those, and every such body when the fused path is off, are involved here.
            if (G.n_static > 0) {
                const GridRec<T> me = G.rec[i];
                int ns = 0;
                for (int s = 0; s < G.n_static; s++)
                    if (rec_meets_static(me, G.sbox + s * SBOX_REALS)) ns++;
                if (G.static_fast ? (ns >= 2 || (ns >= 1 && G.plane_on)) : ns >= 1) any = 1;
            }
        }
        pc[i] = ((uint64_t)owned << 32) | any;
        inpair[i] = (uint8_t)any;
    }
}
template <class T>
__global__ __launch_bounds__(256) void ex_pair_count(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t n_active,
                                                     GridParams<T> G, uint64_t *__restrict__ pc, uint8_t *__restrict__ inpair,
                                                     ExactCounts *__restrict__ C, int32_t *__restrict__ cross_list, int32_t *__restrict__ stage)
{
    st_pair_count<T, GridWalk<T>, int32_t, EX_STAGE_PARTNERS>(GridWalk<T>{ S, gtype, G }, gtype, n_active, G, pc, inpair, C, cross_list,
                                                               blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, stage);
}

// ---- 1b. the same count by a WAVEFRONT per body, for launches of a few thousand bodies: a lane per candidate instead of a lane
// per body.  In a crowded pen every body's nine columns hold every other body -- a lane walking 400 candidates through the
// buckets is a chain of a hundred dependent L2 round trips (94 us for 400 bodies) while the chip idles; here the nine columns'
// counts are fetched together, the candidates are numbered across the columns and dealt to the lanes (two dependent loads each:
// the item, its record), hits are ranked by ballots.  Same pc / inpair / stage as st_pair_count (the staged partners in another
// order: the write pass sorts them).
template <class T>
__global__ __launch_bounds__(256) void ex_pair_count_wave(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t n_active,
                                                          GridParams<T> G, uint64_t *__restrict__ pc, uint8_t *__restrict__ inpair,
                                                          ExactCounts *__restrict__ C, int32_t *__restrict__ cross_list,
                                                          int32_t *__restrict__ stage)
{
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n_active; i += (int64_t)gridDim.x * 4) {      // wave-uniform
        uint32_t owned = 0, any = 0;
        const int gti = gtype[i];
        if (gti != GEOM_NONE) {
            const GridRec<T> me = G.rec[i];
            // lanes 0..8: their column's bucket and its count; everyone: the running totals
            uint32_t myh = 0, mycnt = 0;
            if (lane < 9) {
                myh = cell_hash(me.ix + (lane % 3) - 1, me.iz + (lane / 3) - 1, G.mask, G.xbits);
                mycnt = G.count[myh];
                if (mycnt > (uint32_t)G.cap) mycnt = (uint32_t)G.cap;
            }
            uint32_t start[10], hh[9];
            start[0] = 0;
#pragma unroll
            for (int c = 0; c < 9; c++) { hh[c] = __shfl(myh, c, 64); start[c + 1] = start[c] + __shfl(mycnt, c, 64); }
            const uint32_t total = start[9];
            for (uint32_t u0 = 0; u0 < total; u0 += 64) {
                const uint32_t u = u0 + lane;
                bool hit = false;
                int64_t j = -1;
                if (u < total) {
                    int c = 0;
#pragma unroll
                    for (int q = 1; q < 9; q++) c += u >= start[q] ? 1 : 0;
                    uint32_t hb = hh[0], sb = start[0];
#pragma unroll
                    for (int q = 1; q < 9; q++) if (c == q) { hb = hh[q]; sb = start[q]; }
                    j = G.items[(size_t)hb * G.cap + (u - sb)];
                    if (j != i) {
                        const GridRec<T> o = G.rec[j];
                        const int cx = me.ix + (c % 3) - 1, cz = me.iz + (c / 3) - 1;
                        hit = o.ix == cx && o.iz == cz && classes_collide(gti, gtype[j], G.class_pairs) &&
                              !(o.lo[0] > me.hi[0] || me.lo[0] > o.hi[0] || o.lo[1] > me.hi[1] || me.lo[1] > o.hi[1] ||
                                o.lo[2] > me.hi[2] || me.lo[2] > o.hi[2]);
                    }
                }
                if (__ballot(hit) != 0ull) any = 1;
                if (hit && j >= n_active) {
                    if (atomicOr(&C->cross, 1u) == 0u) { C->cross_a = (uint32_t)i; C->cross_b = (uint32_t)j; }
                    const uint32_t at = atomicAdd(&C->ncross, 1u);
                    if (at < EX_CROSS_CAP) { cross_list[2 * at] = (int32_t)i; cross_list[2 * at + 1] = (int32_t)j; }
                }
                const bool mine = hit && j < n_active && j > i;
                const unsigned long long mb = __ballot(mine);
                if (mine && stage != nullptr) {
                    const uint32_t r = owned + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull));
                    if (r < (uint32_t)EX_STAGE_PARTNERS) stage[(size_t)i * EX_STAGE_PARTNERS + r] = (int32_t)j;
                }
                owned += (uint32_t)__popcll(mb);
            }
            if (G.n_static > 0) {
                bool meets = false;
                for (int s0 = 0; s0 < G.n_static; s0 += 64) meets = meets || (s0 + lane < G.n_static && rec_meets_static(me, G.sbox + (s0 + lane) * SBOX_REALS));
                const int ns = __popcll(__ballot(meets));          // (at most 64 static boxes: one pass)
                if (G.static_fast ? (ns >= 2 || (ns >= 1 && G.plane_on)) : ns >= 1) any = 1;
            }
        }
        if (lane == 0) { pc[i] = ((uint64_t)owned << 32) | any; inpair[i] = (uint8_t)any; }
    }
}

// ---- 2. pairs in canonical order, the involved bodies ascending, union-find initialised ----------------------------
// inc = inclusive scan of pc.  Body i owns pairs [hi(exc), hi(exc) + owned) and, if involved, is entry lo(exc) of `inv`.
template <class T, class W, class ST = uint16_t, int NST = EXS_PARTNERS>
__device__ __forceinline__ void st_pair_write(const W &walk, int64_t n_active, const GridParams<T> &G, const uint64_t *pc,
                                              const uint64_t *inc, int32_t *pairs, int32_t *inv, int32_t *parent, const ExactCaps &cap,
                                              ExactCounts *C, int64_t first, int64_t step, const ST *staged = nullptr)
{
    const uint64_t tot = inc[n_active - 1];
    if (first == 0) {
        C->bp_overflow = G.flags[BPF_OVERFLOW];          // (bp_insert has finished)
        C->npairs = hi32(tot); C->ninv = lo32(tot);
        if (hi32(tot) > cap.pairs || lo32(tot) > cap.inv) atomicOr(&C->overflow, 1u);
    }
    if (hi32(tot) > cap.pairs || lo32(tot) > cap.inv) return;            // the host grows the capacity and runs again
    for (int64_t i = first; i < n_active; i += step) {
        const uint64_t mine = pc[i], exc = inc[i] - mine;
        if (lo32(mine)) {
            const uint32_t k = lo32(exc);
            inv[k] = (int32_t)i;
            parent[k] = (int32_t)k;
        }
        const uint32_t owned = hi32(mine);
        if (owned == 0) continue;
        int32_t *out = pairs + 2 * (size_t)hi32(exc);
        uint32_t w = 0;
        if (staged != nullptr && owned <= (uint32_t)NST) {
            for (; w < owned; w++) { out[2 * w] = (int32_t)i; out[2 * w + 1] = (int32_t)staged[(size_t)i * NST + w]; }
        } else {
            walk(i, [&](int64_t j) {
                if (j > i && j < n_active && w < owned) { out[2 * w] = (int32_t)i; out[2 * w + 1] = (int32_t)j; w++; }
            });
        }
        // this thread's own run of partners, ascending (runs are short: insertion sort in place)
        for (uint32_t a = 1; a < w; a++) {
            const int32_t v = out[2 * a + 1];
            uint32_t q = a;
            while (q > 0 && out[2 * (q - 1) + 1] > v) { out[2 * q + 1] = out[2 * (q - 1) + 1]; q--; }
            out[2 * q + 1] = v;
        }
    }
}
template <class T>
__global__ __launch_bounds__(256) void ex_pair_write(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t n_active,
                                                     GridParams<T> G, const uint64_t *__restrict__ pc, const uint64_t *__restrict__ inc,
                                                     int32_t *__restrict__ pairs, int32_t *__restrict__ inv, int32_t *__restrict__ parent,
                                                     ExactCaps cap, ExactCounts *__restrict__ C, const int32_t *__restrict__ stage)
{
    st_pair_write<T, GridWalk<T>, int32_t, EX_STAGE_PARTNERS>(GridWalk<T>{ S, gtype, G }, n_active, G, pc, inc, pairs, inv, parent, cap, C,
                                                               blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, stage);
}

// ---- 3. connected components: lock-free union-find over the involved bodies' indices k (ascending slot order) -------
__device__ __forceinline__ int uf_find(int32_t *p, int x)
{
    for (;;) {
        const int px = ((volatile int32_t *)p)[x];
        if (px == x) return x;
        const int ppx = ((volatile int32_t *)p)[px];
        if (ppx != px) ((volatile int32_t *)p)[x] = ppx;      // path halving (only ever moves a node closer to its root)
        x = px;
    }
}
__device__ __forceinline__ void uf_unite(int32_t *p, int a, int b)
{
    for (;;) {
        a = uf_find(p, a); b = uf_find(p, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }           // the larger root goes under the smaller: roots are minima
        if (atomicCAS(&p[b], b, a) == b) return;
    }
}
__device__ __forceinline__ uint32_t kidx_of(const uint64_t *pc, const uint64_t *inc, int32_t s) { return lo32(inc[s] - pc[s]); }

__device__ __forceinline__ void st_unite(const int32_t *pairs, const uint64_t *pc, const uint64_t *inc, int32_t *parent,
                                         const ExactCounts *C, uint32_t first, uint32_t step)
{
    const uint32_t np = C->overflow ? 0u : C->npairs;
    for (uint32_t p = first; p < np; p += step)
        uf_unite(parent, (int)kidx_of(pc, inc, pairs[2 * p]), (int)kidx_of(pc, inc, pairs[2 * p + 1]));
}
__global__ __launch_bounds__(256) void ex_unite(const int32_t *__restrict__ pairs, const uint64_t *__restrict__ pc,
                                                const uint64_t *__restrict__ inc, int32_t *parent, const ExactCounts *__restrict__ C)
{
    st_unite(pairs, pc, inc, parent, C, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// root[k]; rf[k] = 1 for roots, 0 elsewhere and for the padding up to the capacity (scanned next: island numbers)
__device__ __forceinline__ void st_flatten(int32_t *parent, int32_t *root, uint32_t *rf, const ExactCaps &cap, const ExactCounts *C,
                                           uint32_t first, uint32_t step)
{
    const uint32_t ninv = C->overflow ? 0u : C->ninv;
    for (uint32_t k = first; k < cap.inv; k += step) {
        uint32_t f = 0;
        if (k < ninv) { const int r = uf_find(parent, (int)k); root[k] = r; f = (r == (int)k) ? 1u : 0u; }
        rf[k] = f;
    }
}
__global__ __launch_bounds__(256) void ex_flatten(int32_t *parent, int32_t *__restrict__ root, uint32_t *__restrict__ rf,
                                                  ExactCaps cap, const ExactCounts *__restrict__ C)
{
    st_flatten(parent, root, rf, cap, C, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ---- 4. narrowphase with device-side counts (same colliders as np_plane / np_pairs) --------------------------------
template <class T> struct BodyGeomX { V3<T> x; M3<T> R; T side[3]; int gt; };
template <class T> __device__ __forceinline__ BodyGeomX<T> geom_of(const T *S, const uint8_t *gtype, int64_t i)
{
    BodyGeomX<T> g;
    g.x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
    g.R = quat_to_R(Q4<T>{ S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)], S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] });
    for (int a = 0; a < 3; a++) g.side[a] = S[slab_ix(C_SIDES + a, i)];
    g.gt = gtype[i];
    return g;
}
template <class T> __device__ __forceinline__ void put_c(T *gpos, T *gnormal, T *gdepth, size_t slot, const V3<T> &p, const V3<T> &n, T d)
{
    gpos[3 * slot] = p.x; gpos[3 * slot + 1] = p.y; gpos[3 * slot + 2] = p.z;
    gnormal[3 * slot] = n.x; gnormal[3 * slot + 1] = n.y; gnormal[3 * slot + 2] = n.z;
    gdepth[slot] = d;
}

__device__ __forceinline__ uint32_t entry_key(uint32_t e, const int32_t *pairs, const uint64_t *pc, const uint64_t *inc, const int32_t *root,
                                              const uint32_t *rinc, const ExactCaps &cap, uint32_t ninv, uint32_t np)
{
    const uint32_t e_pairs = cap.pair_entry0();
    if (e < e_pairs) { const uint32_t k = e % cap.inv; if (k < ninv) return rinc[root[k]] - 1u; }
    else if (e - e_pairs < np) return rinc[root[kidx_of(pc, inc, pairs[2 * (e - e_pairs)])]] - 1u;
    return cap.inv;                                     // padding sorts behind every island
}

// stage 5's arrays (the sort keys: entry -> island, rinc = inclusive scan of the root flags: island of root r = rinc[r] - 1);
// keys == nullptr: not wanted (the one-workgroup form makes its keys itself)
struct SortKeyArgs { const uint64_t *pc, *inc; const int32_t *root; const uint32_t *rinc; uint32_t *keys, *vals; };

// Entries as ExactCaps lays them out: ground-plane contacts of involved body k, contacts of body k with static box s,
// contacts of pair p.  cc[e] = contacts of entry e, 0 for the padding.
template <class T>
__global__ __launch_bounds__(64) void ex_narrow(const T *__restrict__ S, const uint8_t *__restrict__ gtype, const int32_t *__restrict__ inv,
                                                const int32_t *__restrict__ pairs, const GridRec<T> *__restrict__ rec, StepParams<T> P, ExactCaps cap,
                                                T *__restrict__ gpos, T *__restrict__ gnormal, T *__restrict__ gdepth,
                                                uint32_t *__restrict__ cc, ExactCounts *C, SortKeyArgs K)
{
    const uint32_t ninv = C->overflow ? 0u : C->ninv, np = C->overflow ? 0u : C->npairs;
    const uint32_t ne = cap.entries(), e_pairs = cap.pair_entry0();
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) {
        if (K.keys != nullptr) {       // stage 5 rides along (one launch fewer): entry e -> its island
            K.keys[e] = entry_key(e, pairs, K.pc, K.inc, K.root, K.rinc, cap, ninv, np); K.vals[e] = e;
            if (e == 0) C->ni = K.rinc[cap.inv - 1];
        }
        int nc = 0;
        if (P.hull_n > 0) {   // entries with a convex body in them belong to ex_narrow_convex (one wavefront each; not launched without a hull)
            bool convex = false;
            if (e < e_pairs) { const uint32_t k = e % cap.inv; convex = k < ninv && gtype[inv[k]] == GEOM_CONVEX; }
            else if (e - e_pairs < np) convex = gtype[pairs[2 * (e - e_pairs)]] == GEOM_CONVEX || gtype[pairs[2 * (e - e_pairs) + 1]] == GEOM_CONVEX;
            if (convex) continue;
        }
        if (e >= cap.inv && e < e_pairs) {
            // body k against static box s: dCollide(static geom, body geom) -- the static geoms were created first
            // (main.c:115-121), so they are o1; the contact joint is attached (0, body), i.e. reversed: normal negated
            const uint32_t s = (e - cap.inv) / cap.inv, k = (e - cap.inv) - s * cap.inv;
            if (k < ninv) {
                const int64_t i = inv[k];
                const T *sb = P.sbox + s * SBOX_REALS;
                if (rec_meets_static(rec[i], sb)) {
                    const BodyGeomX<T> Bd = geom_of<T>(S, gtype, i);
                    const V3<T> sx = { sb[SBOX_POS], sb[SBOX_POS + 1], sb[SBOX_POS + 2] };
                    M3<T> sR;
                    for (int a = 0; a < 3; a++) for (int c2 = 0; c2 < 3; c2++) sR.m[a][c2] = sb[SBOX_R + 3 * a + c2];
                    const T sside[3] = { sb[SBOX_SIDE], sb[SBOX_SIDE + 1], sb[SBOX_SIDE + 2] };
                    ContactPoint<T> c[8];
                    const int mc = P.max_contacts > 8 ? 8 : P.max_contacts;
                    bool negate = true;          // the joint's reversal
                    if (Bd.gt == GEOM_BOX) nc = box_box(sx, sR, sside, Bd.x, Bd.R, Bd.side, mc, c);
                    else if (Bd.gt == GEOM_SPHERE) { nc = sphere_box(Bd.x, Bd.side[0], sx, sR, sside, c); negate = false; }   // swapped collider: flipped twice
                    if (nc > mc) nc = mc;
                    const size_t base = cap.static_slot0() + (size_t)8 * ((size_t)s * cap.inv + k);
#pragma unroll
                    for (int q = 0; q < 8; q++) {           // (static indices: the contacts stay in registers)
                        if (q < nc) {
                            const V3<T> n = negate ? V3<T>{ -c[q].normal.x, -c[q].normal.y, -c[q].normal.z } : c[q].normal;
                            put_c(gpos, gnormal, gdepth, base + q, c[q].pos, n, c[q].depth);
                        }
                    }
                }
            }
        } else if (e < cap.inv) {
            if (e < ninv && P.plane_on) {
                const BodyGeomX<T> g = geom_of<T>(S, gtype, inv[e]);
                V3<T> cp[4]; T cd[4];
                if (g.gt == GEOM_BOX) nc = box_plane(g.x, g.R, g.side, P.pn, P.pd, P.max_contacts, cp, cd);
                else if (g.gt == GEOM_SPHERE) nc = sphere_plane(g.x, g.side[0], P.pn, P.pd, cp, cd);
                for (int c = 0; c < nc; c++) put_c(gpos, gnormal, gdepth, (size_t)8 * e + c, cp[c], P.pn, cd[c]);
            }
        } else {
            const uint32_t p = e - e_pairs;
            if (p < np) {
                const BodyGeomX<T> A = geom_of<T>(S, gtype, pairs[2 * p]);
                const BodyGeomX<T> B = geom_of<T>(S, gtype, pairs[2 * p + 1]);
                ContactPoint<T> c[8];
                bool flip = false;      // a collider exists only for the swapped class order: swap, then negate the normal
                const int mc = P.max_contacts > 8 ? 8 : P.max_contacts;
                if (A.gt == GEOM_BOX && B.gt == GEOM_BOX) nc = box_box(A.x, A.R, A.side, B.x, B.R, B.side, mc, c);
                else if (A.gt == GEOM_SPHERE && B.gt == GEOM_SPHERE) nc = sphere_sphere(A.x, A.side[0], B.x, B.side[0], c);
                else if (A.gt == GEOM_SPHERE && B.gt == GEOM_BOX) nc = sphere_box(A.x, A.side[0], B.x, B.R, B.side, c);
                else if (A.gt == GEOM_BOX && B.gt == GEOM_SPHERE) { nc = sphere_box(B.x, B.side[0], A.x, A.R, A.side, c); flip = true; }
                if (nc > mc) nc = mc;
                const size_t base = cap.pair_slot0() + (size_t)8 * p;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (k < nc) {
                        const V3<T> n = flip ? V3<T>{ -c[k].normal.x, -c[k].normal.y, -c[k].normal.z } : c[k].normal;
                        put_c(gpos, gnormal, gdepth, base + k, c[k].pos, n, c[k].depth);
                    }
                }
            }
        }
        cc[e] = (uint32_t)nc;
    }
}

// ---- 4b. entries with a convex body: one wavefront per entry ---------------------------------------------------------
// hull against the ground plane: dCollideConvexPlane as np_convex_plane restates it (points in array order, first
// max_contacts on or below the plane, both-sides rule).  Hull against a box (a static box, or a box body): this library's
// collider (dmx_batch.h, dmxBatchSetConvexHullFaces): hull vertices inside the box in array order, then box corners inside
// the hull.  Lane l tests point / face 64 j + l; ballots give array-order ranks, so the contacts are the ones a sequential
// walk keeps.
template <class T>
__global__ __launch_bounds__(256) void ex_narrow_convex(const T *__restrict__ S, const uint8_t *__restrict__ gtype,
                                                        const int32_t *__restrict__ inv, const int32_t *__restrict__ pairs,
                                                        const GridRec<T> *__restrict__ rec, StepParams<T> P, ExactCaps cap,
                                                        T *__restrict__ gpos, T *__restrict__ gnormal, T *__restrict__ gdepth,
                                                        uint32_t *__restrict__ cc, ExactCounts *__restrict__ C, int hull_pairs_elsewhere)
{
    // hull_pairs_elsewhere: pairs of two hulls are ex_narrow_hull_pairs' (a workgroup each), not this kernel's
    const uint32_t ninv = C->overflow ? 0u : C->ninv, np = C->overflow ? 0u : C->npairs;
    const uint32_t ne = cap.entries(), e_pairs = cap.pair_entry0();
    const int lane = threadIdx.x & 63;
    const int maxc = P.max_contacts < CONVEX_MAXC ? P.max_contacts : CONVEX_MAXC;
    for (uint32_t e = blockIdx.x * 4 + (threadIdx.x >> 6); e < ne; e += gridDim.x * 4) {          // wave-uniform
        int nc = 0;
        if (e < e_pairs) {
            const uint32_t k = e % cap.inv;
            if (k >= ninv) continue;
            const int64_t i = inv[k];
            if (gtype[i] != GEOM_CONVEX) continue;
            const BodyGeomX<T> H = geom_of<T>(S, gtype, i);
            if (e < cap.inv) {
                // ---- hull against the ground plane
                if (P.plane_on && P.hull_n > 0)
                    nc = wave_convex_plane<T>(H.x, H.R, H.side[0], P, maxc, lane, [&](int rank, const V3<T> &p, const V3<T> &nn, T dep) {
                        put_c(gpos, gnormal, gdepth, (size_t)8 * e + rank, p, nn, dep); }, P.hull);
            } else {
                // ---- hull against static box s: dCollide(static box, hull); the joint is attached (0, body): reversed
                const uint32_t s = (e - cap.inv) / cap.inv;
                const T *sb = P.sbox + s * SBOX_REALS;
                if (rec_meets_static(rec[i], sb) && P.hull_n > 0) {
                    const V3<T> sx = { sb[SBOX_POS], sb[SBOX_POS + 1], sb[SBOX_POS + 2] };
                    M3<T> sR;
                    for (int a = 0; a < 3; a++) for (int c2 = 0; c2 < 3; c2++) sR.m[a][c2] = sb[SBOX_R + 3 * a + c2];
                    const T sside[3] = { sb[SBOX_SIDE], sb[SBOX_SIDE + 1], sb[SBOX_SIDE + 2] };
                    const size_t slot0 = cap.static_slot0() + (size_t)8 * (e - cap.inv);
                    nc = wave_box_convex<T>(sx, sR, sside, H.x, H.R, H.side[0], P, maxc, true, lane,
                                            [&](int rank, const V3<T> &p, const V3<T> &nn, T dep) { put_c(gpos, gnormal, gdepth, slot0 + rank, p, nn, dep); }, P.hull);
                }
            }
        } else {
            const uint32_t p = e - e_pairs;
            if (p >= np) continue;
            const int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
            const int gi = gtype[i], gj = gtype[j];
            if (gi != GEOM_CONVEX && gj != GEOM_CONVEX) continue;
            if (gi == GEOM_BOX || gj == GEOM_BOX) {
                // (box i, hull j): the collider's own order, normal into i.  (hull i, box j): dCollide swaps and flips.
                const BodyGeomX<T> Bx = geom_of<T>(S, gtype, gi == GEOM_BOX ? i : j);
                const BodyGeomX<T> H = geom_of<T>(S, gtype, gi == GEOM_BOX ? j : i);
                const size_t slot0 = cap.pair_slot0() + (size_t)8 * p;
                if (P.hull_n > 0)
                    nc = wave_box_convex<T>(Bx.x, Bx.R, Bx.side, H.x, H.R, H.side[0], P, maxc, gi != GEOM_BOX, lane,
                                            [&](int rank, const V3<T> &pp, const V3<T> &nn, T dep) { put_c(gpos, gnormal, gdepth, slot0 + rank, pp, nn, dep); }, P.hull);
            } else if (gi == GEOM_CONVEX && gj == GEOM_CONVEX) {
                if (hull_pairs_elsewhere) continue;
                // hull i (geom 1, created first) against hull j: vertices of each inside the other, normals into i
                const BodyGeomX<T> A = geom_of<T>(S, gtype, i), Bh = geom_of<T>(S, gtype, j);
                const size_t slot0 = cap.pair_slot0() + (size_t)8 * p;
                const T boxA[6] = { rec[i].lo[0], rec[i].lo[1], rec[i].lo[2], rec[i].hi[0], rec[i].hi[1], rec[i].hi[2] };
                const T boxB[6] = { rec[j].lo[0], rec[j].lo[1], rec[j].lo[2], rec[j].hi[0], rec[j].hi[1], rec[j].hi[2] };
                nc = wave_convex_convex<T>(A.x, A.R, Bh.x, Bh.R, A.side[0], boxA, boxB, P, maxc, false, lane,
                                           [&](int rank, const V3<T> &pp, const V3<T> &nn, T dep) { put_c(gpos, gnormal, gdepth, slot0 + rank, pp, nn, dep); });
            } else if (gi == GEOM_SPHERE || gj == GEOM_SPHERE) {
                // (sphere i, hull j): the collider's own order, normal into the sphere.  (hull i, sphere j): dCollide swaps and flips.
                const BodyGeomX<T> Sp = geom_of<T>(S, gtype, gi == GEOM_SPHERE ? i : j);
                const BodyGeomX<T> H = geom_of<T>(S, gtype, gi == GEOM_SPHERE ? j : i);
                const size_t slot0 = cap.pair_slot0() + (size_t)8 * p;
                nc = wave_sphere_convex<T>(Sp.x, Sp.side[0], H.x, H.R, P, gi != GEOM_SPHERE, lane,
                                           [&](int rank, const V3<T> &pp, const V3<T> &nn, T dep) { put_c(gpos, gnormal, gdepth, slot0 + rank, pp, nn, dep); });
            }
        }
        if (lane == 0) cc[e] = (uint32_t)nc;
    }
}

// ---- 4c. pairs of two hulls: one WORKGROUP per pair (wg_convex_convex, dmx_collide_wave.hpp) ---------------------------------------
template <class T>
__global__ __launch_bounds__(256) void ex_narrow_hull_pairs(const T *__restrict__ S, const uint8_t *__restrict__ gtype,
                                                            const int32_t *__restrict__ pairs, const GridRec<T> *__restrict__ rec,
                                                            StepParams<T> P, ExactCaps cap, T *__restrict__ gpos, T *__restrict__ gnormal,
                                                            T *__restrict__ gdepth, uint32_t *__restrict__ cc, ExactCounts *__restrict__ C)
{
    const uint32_t np = C->overflow ? 0u : C->npairs;
    const uint32_t e_pairs = cap.pair_entry0();
    const int maxc = P.max_contacts < CONVEX_MAXC ? P.max_contacts : CONVEX_MAXC;
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {                     // workgroup-uniform
        const int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
        if (gtype[i] != GEOM_CONVEX || gtype[j] != GEOM_CONVEX) continue;
        const BodyGeomX<T> A = geom_of<T>(S, gtype, i), Bh = geom_of<T>(S, gtype, j);
        const size_t slot0 = cap.pair_slot0() + (size_t)8 * p;
        const T boxA[6] = { rec[i].lo[0], rec[i].lo[1], rec[i].lo[2], rec[i].hi[0], rec[i].hi[1], rec[i].hi[2] };
        const T boxB[6] = { rec[j].lo[0], rec[j].lo[1], rec[j].lo[2], rec[j].hi[0], rec[j].hi[1], rec[j].hi[2] };
        const int nc = wg_convex_convex<T, 4>(A.x, A.R, Bh.x, Bh.R, A.side[0], boxA, boxB, P, maxc, false,
                                              [&](int rank, const V3<T> &pp, const V3<T> &nn, T dep) { put_c(gpos, gnormal, gdepth, slot0 + rank, pp, nn, dep); });
        if (threadIdx.x == 0) cc[e_pairs + p] = (uint32_t)nc;
        __syncthreads();
    }
}

// ---- 5. sort keys: written by ex_narrow (entry_key, above) --------------------------------------------------------------
// ---- 6. sorted entries -> (contacts << 32 | is-body-entry), scanned next ------------------------------------------
__device__ __forceinline__ uint64_t gathered(uint32_t key, uint32_t e, const uint32_t *cc, const ExactCaps &cap)
{
    return key < cap.inv ? (((uint64_t)cc[e] << 32) | (e < cap.inv ? 1u : 0u)) : 0ull;
}
// (the stage-per-launch form scans this through a transform iterator: no array, no launch of its own)
struct GatherOp {
    const uint32_t *keys_s, *vals_s, *cc;
    ExactCaps cap;
    __device__ uint64_t operator()(uint32_t t) const { return gathered(keys_s[t], vals_s[t], cc, cap); }
};

// ---- 7. island boundaries: first sorted entry of each island gives its offsets ------------------------------------
__device__ __forceinline__ void st_bounds(const uint32_t *keys_s, const uint32_t *vals_s, const uint32_t *cc, const uint64_t *sinc,
                                          const ExactCaps &cap, int *body_off, int *con_off, int *row_off, ExactCounts *C, uint32_t first,
                                          uint32_t step)
{
    const uint32_t ne = cap.entries();
    for (uint32_t t = first; t < ne; t += step) {
        const uint32_t key = keys_s[t];
        if (key < cap.inv && (t == 0 || keys_s[t - 1] != key)) {
            const uint64_t exc = sinc[t] - gathered(key, vals_s[t], cc, cap);
            body_off[key] = (int)lo32(exc); con_off[key] = (int)hi32(exc); row_off[key] = 3 * (int)hi32(exc);
        }
        if (t == ne - 1) {
            const uint64_t tot = sinc[t];
            const uint32_t ni = C->ni;
            body_off[ni] = (int)lo32(tot); con_off[ni] = (int)hi32(tot); row_off[ni] = 3 * (int)hi32(tot);
            C->njoints = hi32(tot);
        }
    }
}
__global__ __launch_bounds__(256) void ex_bounds(const uint32_t *__restrict__ keys_s, const uint32_t *__restrict__ vals_s,
                                                 const uint32_t *__restrict__ cc, const uint64_t *__restrict__ sinc, ExactCaps cap,
                                                 int *__restrict__ body_off, int *__restrict__ con_off, int *__restrict__ row_off,
                                                 ExactCounts *C)
{
    st_bounds(keys_s, vals_s, cc, sinc, cap, body_off, con_off, row_off, C, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

__device__ __forceinline__ void st_bigflags(const int *con_off, const int *body_off, const ExactCaps &cap, int rpc, int big_rows,
                                            uint64_t *bg, const ExactCounts *C, uint32_t first, uint32_t step);

// ---- 8. the island-grouped body list and contact arrays -------------------------------------------------------------
__device__ __forceinline__ void st_fill(const uint32_t *keys_s, const uint32_t *vals_s, const uint64_t *sinc,
                                        const uint32_t *cc, const int32_t *inv, const int32_t *pairs, const int *con_off,
                                        const ExactCaps &cap, int rpc, int *bodies, int *cb1, int *cb2, int *csrc, int *crow,
                                        uint32_t first, uint32_t step)
{
    const uint32_t ne = cap.entries(), e_pairs = cap.pair_entry0();
    for (uint32_t t = first; t < ne; t += step) {
        const uint32_t key = keys_s[t];
        if (key >= cap.inv) continue;
        const uint32_t e = vals_s[t];
        const uint64_t exc = sinc[t] - gathered(key, e, cc, cap);
        const int d0 = (int)hi32(exc), c0 = con_off[key];
        int b1, b2, src0;
        if (e < cap.inv) { b1 = inv[e]; b2 = -1; src0 = 8 * (int)e; bodies[lo32(exc)] = b1; }
        else if (e < e_pairs) { b1 = inv[e % cap.inv]; b2 = -1; src0 = (int)(cap.static_slot0() + (size_t)8 * (e - cap.inv)); }
        else { const uint32_t p = e - e_pairs; b1 = pairs[2 * p]; b2 = pairs[2 * p + 1]; src0 = (int)(cap.pair_slot0() + (size_t)8 * p); }
        const int nc = (int)cc[e];
        for (int c = 0; c < nc; c++) {
            const int d = d0 + c;
            cb1[d] = b1; cb2[d] = b2; csrc[d] = src0 + c; crow[d] = rpc * (d - c0);
        }
    }
}
__global__ __launch_bounds__(256) void ex_fill(const uint32_t *__restrict__ keys_s, const uint32_t *__restrict__ vals_s,
                                               const uint64_t *__restrict__ sinc,
                                               const uint32_t *__restrict__ cc, const int32_t *__restrict__ inv,
                                               const int32_t *__restrict__ pairs, const int *__restrict__ con_off, ExactCaps cap, int rpc,
                                               int *__restrict__ bodies, int *__restrict__ cb1, int *__restrict__ cb2,
                                               int *__restrict__ csrc, int *__restrict__ crow, const int *__restrict__ body_off,
                                               int big_rows, uint64_t *__restrict__ bg, const ExactCounts *__restrict__ C)
{
    st_fill(keys_s, vals_s, sinc, cc, inv, pairs, con_off, cap, rpc, bodies, cb1, cb2, csrc, crow,
            blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    // stage 9 needs the island offsets only, like this one: same launch
    st_bigflags(con_off, body_off, cap, rpc, big_rows, bg, C, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ---- 9. which islands get a workgroup: bg[isl] = (rows << 32 | 1) for those, 0 otherwise and for the padding -------
__device__ __forceinline__ void st_bigflags(const int *con_off, const int *body_off, const ExactCaps &cap, int rpc, int big_rows,
                                            uint64_t *bg, const ExactCounts *C, uint32_t first, uint32_t step)
{
    const uint32_t ni = C->overflow ? 0u : C->ni;
    for (uint32_t i = first; i < cap.inv; i += step) {
        uint64_t v = 0;
        if (i < ni) {
            const int nc = con_off[i + 1] - con_off[i], m = rpc * nc;
            const bool single = (body_off[i + 1] - body_off[i]) == 1 && nc >= 1 && nc <= 8;      // solve_singles' / solve_singles_lds' islands
            if (m >= big_rows && !single) v = ((uint64_t)(uint32_t)m << 32) | 1u;
        }
        bg[i] = v;
    }
}

// which form builds an island's schedule: the workgroup's (levels_coop) for islands that need row lists at all, when few islands
// have a workgroup (nbig) -- see st_levels
constexpr int LEVELS_COOP_ROWS = WAVE_ISLAND_ROWS, LEVELS_COOP_MAX_ISLANDS = 64;
__device__ __forceinline__ bool levels_by_workgroup(int m, int nbig, int coop_rows)
{
    return coop_rows > 0 && m > coop_rows && nbig <= LEVELS_COOP_MAX_ISLANDS;
}

// ---- 10. level schedules.  One lane per island; islands own disjoint bodies, so `last` (per slot, -1 when idle) is
//          private to the lane while it works.  lev_off of island k (big index) lives at row_base + k, nlev + 1 <= rows + 1
//          entries; lev_rows / row_level at row_base. -----------------------------------------------------------------------
__device__ __forceinline__ void st_levels(const int *con_off, const int *body_off, const int *cb1, const int *cb2, const uint64_t *bg,
                                          const uint64_t *binc, const ExactCaps &cap, int rpc, int *big, int *big_list, int *lev_count,
                                          int *lev_off, int *lev_rows, int *row_level, int *last, ExactCounts *C, uint32_t first,
                                          uint32_t step, int coop_rows = 0)
{
    // coop_rows > 0: islands of more rows than that are only entered in the big list here when the list is short; their schedules
    // are built by a whole workgroup (levels_coop below) -- one lane walking thousands of contacts through device memory is a
    // millisecond.  (Thousands of such islands are the other case: a lane each, all at once, is the faster form then.)
    const uint32_t ni = C->overflow ? 0u : C->ni;
    for (uint32_t i = first; i < ni; i += step) {
        if (i == 0) {
            const uint64_t tot = binc[cap.inv - 1];
            C->nbig = lo32(tot); C->big_rows = hi32(tot);
            if (hi32(tot) > cap.rows) atomicOr(&C->overflow, 2u);
        }
        const uint64_t mine = bg[i];
        if (!lo32(mine)) { big[i] = -1; continue; }
        const uint64_t exc = binc[i] - mine;
        const int k = (int)lo32(exc), base = (int)hi32(exc), m = (int)hi32(mine);
        big[i] = base + k;
        big_list[k] = (int)i;
        atomicMax(&C->big_max_bodies, (uint32_t)(body_off[i + 1] - body_off[i]));
        atomicMax(&C->big_max_rows, (uint32_t)m);
        if ((uint32_t)(base + m) > cap.rows) continue;          // flagged above; the host grows the capacity
        if (levels_by_workgroup(m, (int)lo32(binc[cap.inv - 1]), coop_rows)) continue;
        int *lv_out = row_level + base, *off = lev_off + base + k, *rows_out = lev_rows + base;
        // consecutive contacts between the same bodies (a box's four contacts with the plane, a pair's) form one group: its
        // rows take consecutive levels, and `last` is touched once per group, not once per row
        int nlev = 0, r = 0, gb1 = -2, gb2 = -2, glast = -1;
        for (int d = con_off[i]; d < con_off[i + 1]; d++) {
            const int b1 = cb1[d], b2 = cb2[d];
            if (b1 != gb1 || b2 != gb2) {
                if (gb1 >= 0) { last[gb1] = glast; if (gb2 >= 0) last[gb2] = glast; }
                int lv = last[b1];
                if (b2 >= 0 && last[b2] > lv) lv = last[b2];
                glast = lv; gb1 = b1; gb2 = b2;
            }
            for (int q = 0; q < rpc; q++, r++) lv_out[r] = ++glast;
            if (glast + 1 > nlev) nlev = glast + 1;
        }
        for (int d = con_off[i]; d < con_off[i + 1]; d++) {       // back to the idle state
            last[cb1[d]] = -1;
            if (cb2[d] >= 0) last[cb2[d]] = -1;
        }
        lev_count[k] = nlev;
        if (m <= WAVE_ISLAND_ROWS) {
            // solve_island_wg keeps such an island's rows in one wavefront's registers and reads row_level only (and the
            // island's extent from the ends of its offsets)
            off[0] = base; off[nlev] = base + m;
            continue;
        }
        for (int q = 0; q <= nlev; q++) off[q] = 0;
        for (int q = 0; q < m; q++) off[lv_out[q] + 1]++;
        int w = 0;
        for (int q = 1; q <= nlev; q++) w = off[q] > w ? off[q] : w;
        atomicMax(&C->big_max_width, (uint32_t)w);
        off[0] = base;
        for (int q = 0; q < nlev; q++) off[q + 1] += off[q];
        // fill: rows of a level in creation order (off[lv] doubles as the fill cursor; restored afterwards)
        for (int q = 0; q < m; q++) rows_out[off[lv_out[q]]++ - base] = q;
        for (int q = nlev; q > 0; q--) off[q] = off[q - 1];
        off[0] = base;
    }
}
__global__ __launch_bounds__(64) void ex_levels(const int *__restrict__ con_off, const int *__restrict__ body_off,
                                                const int *__restrict__ cb1, const int *__restrict__ cb2,
                                                const uint64_t *__restrict__ bg, const uint64_t *__restrict__ binc, ExactCaps cap, int rpc,
                                                int *__restrict__ big, int *__restrict__ big_list, int *__restrict__ lev_count,
                                                int *__restrict__ lev_off, int *__restrict__ lev_rows, int *__restrict__ row_level,
                                                int *__restrict__ last, ExactCounts *C, int coop_rows)
{
    st_levels(con_off, body_off, cb1, cb2, bg, binc, cap, rpc, big, big_list, lev_count, lev_off, lev_rows, row_level, last, C,
              blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, coop_rows);
}

// ================================================================================================ small scenes
// A scene of a few thousand bodies gives every stage above a few microseconds of work for one workgroup; launched one by
// one, the ~25 stages (rocPRIM's scans and sort are two or three launches each, plus the memsets) cost a launch apiece and
// the tick is all launch latency (configs[0]: ~120 us of it per exact tick).  The same stage functions, in the same order,
// as two one-workgroup kernels around the narrowphase: barriers instead of launches, workgroup scans, one block radix sort
// of (island << entry bits | entry) keys -- stable by construction.  The back kernel leaves the counts and the broadphase
// flags in pinned host memory itself, so the tick's one wait on the device is all the host does.
constexpr int EXS_WG = 1024;

template <class V> __device__ __forceinline__ V wave_scan_inclusive(V x, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const V y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    return x;
}
// out[i] = in[0] + ... + in[i], i < n; thread t owns the run [t per, (t + 1) per).  wt: EXS_WG / 64 values of LDS.
template <class V> __device__ __forceinline__ void block_scan_inclusive(const V *in, V *out, uint32_t n, V *wt)
{
    const uint32_t tid = threadIdx.x, per = (n + EXS_WG - 1) / EXS_WG;
    const uint32_t a = tid * per < n ? tid * per : n, e = a + per < n ? a + per : n;
    V sum = 0;
    for (uint32_t i = a; i < e; i++) sum += in[i];
    const int lane = (int)(tid & 63u);
    const V x = wave_scan_inclusive<V>(sum, lane);
    if (lane == 63) wt[tid >> 6] = x;
    __syncthreads();
    if (tid < 64) {
        const V t = wave_scan_inclusive<V>(tid < EXS_WG / 64 ? wt[tid] : V(0), lane);
        if (tid < EXS_WG / 64) wt[tid] = t;
    }
    __syncthreads();
    V run = ((tid >> 6) > 0 ? wt[(tid >> 6) - 1] : V(0)) + x - sum;
    for (uint32_t i = a; i < e; i++) { run += in[i]; out[i] = run; }
    __syncthreads();
}

// the count record and the broadphase flags into pinned host memory: one lane per word, so the record crosses the bus as a
// couple of wide writes, not as two dozen one after the other.  Called by the whole workgroup after a barrier.
// The record's last word is the caller's sequence number and goes out after everything else has been fenced to the system:
// the host may watch for it instead of waiting for the stream (a stream synchronisation costs it 10-20 us to wake up).
__device__ __forceinline__ void publish_counts(const ExactCounts *C, const uint32_t *flags, ExactCounts *host_counts, uint32_t *host_flags,
                                               uint32_t seq)
{
    if (host_counts == nullptr) return;
    const uint32_t t = threadIdx.x;
    constexpr uint32_t NW = sizeof(ExactCounts) / 4;
    static_assert(offsetof(ExactCounts, seq) == (NW - 1) * 4, "seq is the record's last word");
    if (t < NW - 1) ((volatile uint32_t *)host_counts)[t] = ((const volatile uint32_t *)C)[t];
    else if (t >= NW && t < NW + (uint32_t)BPF_COUNT) ((volatile uint32_t *)host_flags)[t - NW] = ((const volatile uint32_t *)flags)[t - NW];
    if (t < NW + (uint32_t)BPF_COUNT) __threadfence_system();
    __syncthreads();
    if (t == 0) { ((volatile uint32_t *)host_counts)[NW - 1] = seq; __threadfence_system(); }
}

// ---- 10b. the level schedule of ONE large island, by a whole workgroup of EXS_WG threads.  Same schedule as st_levels' (a row's
// level is one more than the latest level of any earlier row on either of its bodies; a group's rows -- consecutive contacts
// between the same two bodies -- take consecutive levels), built differently: groups found by a scan, the bodies' slots mapped to
// island-local numbers through `last` (all -1 on entry and on return), the one sequential part -- the walk over the GROUPS --
// replaced by a relaxation over the groups' predecessors (below; round 3 had one lane walk them in LDS), then every row's level,
// the per-level counts, offsets and row lists in parallel.  Rows of one level touch disjoint bodies, so their order inside a level list is free (atomic cursors).
// lds: at least levels_coop_bytes(nb, groups) bytes; returns false (nothing written) when the island does not fit -- the caller
// falls back on the one-lane walk.
__host__ __device__ inline size_t levels_coop_bytes(size_t nb, size_t groups) { return 8 * nb + 24 * groups + 32; }

__device__ __forceinline__ bool levels_coop(int isl, int k, const int *con_off, const int *body_off, const int *bodies, const int *cb1,
                                            const int *cb2, int rpc, const int *big, int *lev_count, int *lev_off, int *lev_rows,
                                            int *row_level, int *last, ExactCounts *C, unsigned char *lds, size_t lds_bytes, uint32_t *wt)
{
    const int tid = threadIdx.x;
    const int c0 = con_off[isl], nc = con_off[isl + 1] - c0, b0 = body_off[isl], nb = body_off[isl + 1] - b0;
    const int m = nc * rpc, base = big[isl] - k;
    int *lv_out = row_level + base, *off = lev_off + base + k, *rows_out = lev_rows + base;
    uint32_t *gidx = reinterpret_cast<uint32_t *>(rows_out);       // scratch until the lists are filled: contact -> its group + 1
    for (int d = tid; d < nc; d += EXS_WG)
        gidx[d] = (d == 0 || cb1[c0 + d] != cb1[c0 + d - 1] || cb2[c0 + d] != cb2[c0 + d - 1]) ? 1u : 0u;
    __syncthreads();
    block_scan_inclusive<uint32_t>(gidx, gidx, (uint32_t)nc, wt);
    const int G = (int)gidx[nc - 1];
    if (levels_coop_bytes((size_t)nb, (size_t)G) > lds_bytes || nb > 65535 || G > 65535) { __syncthreads(); return false; }
    int32_t *bcur = reinterpret_cast<int32_t *>(lds);              // [nb + 1] per island-local body: its groups' count, then the run's start / cursor
    int32_t *bend = bcur + nb + 1;                                  // [nb] the run's end
    int32_t *ghead = bend + nb;                                     // [G + 1] a group's first contact (island-relative)
    int32_t *glev = ghead + G + 1;                                  // [G] the level below the group's first row
    int32_t *gend = glev + G;                                       // [G] the level of its last row
    uint16_t *gl1 = reinterpret_cast<uint16_t *>(gend + G), *gl2 = gl1 + G;     // [G] its bodies, island-local (0xffff: none)
    uint16_t *gp1 = gl2 + G, *gp2 = gp1 + G;                        // [G] the group before it on each of its bodies (0xffff: none)
    uint16_t *blist = gp2 + G;                                      // [2 G] every body's groups, ascending
    __shared__ int nlev_s;
    for (int t = tid; t < nb; t += EXS_WG) { last[bodies[b0 + t]] = t; bcur[t] = 0; }
    if (tid == 0) { ghead[G] = nc; bcur[nb] = 0; nlev_s = 0; }
    __syncthreads();
    for (int d = tid; d < nc; d += EXS_WG) {
        const int g = (int)gidx[d] - 1;
        if (d == 0 || (int)gidx[d - 1] - 1 != g) {
            ghead[g] = d;
            const int l1 = last[cb1[c0 + d]], l2 = cb2[c0 + d] >= 0 ? last[cb2[c0 + d]] : 0xffff;
            gl1[g] = (uint16_t)l1; gl2[g] = (uint16_t)l2;
            atomicAdd(&bcur[l1], 1);
            if (l2 != 0xffff) atomicAdd(&bcur[l2], 1);
        }
    }
    __syncthreads();
    // The one sequential part of the schedule is a walk over the groups in creation order: a group starts one level above the latest
    // row on either of its bodies.  That is a longest-path recurrence over "the group before this one on body 1 / on body 2" -- so
    // find those two predecessors for every group (a counting sort of the groups by body, each body's short run sorted by a lane),
    // and let every group take max(end of predecessor 1, end of predecessor 2) until nothing changes: as many rounds as the longest
    // chain of groups has links (a pile of 512 bodies: ~20), each one pass of the workgroup over LDS, instead of one lane's 580
    // dependent steps (70 us of the pen's tick).  The recurrence has one solution: the same levels.
    block_scan_inclusive<uint32_t>(reinterpret_cast<uint32_t *>(bcur), reinterpret_cast<uint32_t *>(bcur), (uint32_t)nb, wt);
    for (int t = tid; t < nb; t += EXS_WG) bend[t] = bcur[t];      // (inclusive sums: the runs' ends; a run's cursor counts down from its end)
    __syncthreads();
    for (int g = tid; g < G; g += EXS_WG) {
        const int l1 = gl1[g], l2 = gl2[g];
        blist[atomicSub(&bcur[l1], 1) - 1] = (uint16_t)g;
        if (l2 != 0xffff) blist[atomicSub(&bcur[l2], 1) - 1] = (uint16_t)g;
        gp1[g] = gp2[g] = (uint16_t)0xffffu;
    }
    __syncthreads();
    for (int t = tid; t < nb; t += EXS_WG) {                        // (bcur[t] is the run's start now)
        const int lo = bcur[t], hi = bend[t];
        for (int a = lo + 1; a < hi; a++) {                         // insertion sort: a body touches a handful of others
            const uint16_t v = blist[a];
            int c = a - 1;
            while (c >= lo && blist[c] > v) { blist[c + 1] = blist[c]; c--; }
            blist[c + 1] = v;
        }
        for (int a = lo + 1; a < hi; a++) {
            const int g = blist[a];
            if (gl1[g] == t) gp1[g] = blist[a - 1]; else gp2[g] = blist[a - 1];
        }
    }
    __syncthreads();
    for (int g = tid; g < G; g += EXS_WG) { glev[g] = -1; gend[g] = -1 + (ghead[g + 1] - ghead[g]) * rpc; }
    __syncthreads();
    for (;;) {
        bool changed = false;
        for (int g = tid; g < G; g += EXS_WG) {
            const int p1 = gp1[g], p2 = gp2[g];
            int lv = p1 != 0xffff ? gend[p1] : -1;
            if (p2 != 0xffff && gend[p2] > lv) lv = gend[p2];
            if (lv != glev[g]) { glev[g] = lv; changed = true; }
        }
        if (!__syncthreads_or(changed)) break;
        for (int g = tid; g < G; g += EXS_WG) gend[g] = glev[g] + (ghead[g + 1] - ghead[g]) * rpc;
        __syncthreads();
    }
    {
        int mx = 0;
        for (int g = tid; g < G; g += EXS_WG) mx = gend[g] + 1 > mx ? gend[g] + 1 : mx;
        if (mx > 0) atomicMax(&nlev_s, mx);
    }
    __syncthreads();
    if (tid == 0) lev_count[k] = nlev_s;
    for (int t = tid; t < nb; t += EXS_WG) last[bodies[b0 + t]] = -1;      // back to the idle state
    __syncthreads();
    const int nlev = nlev_s;
    for (int d = tid; d < nc; d += EXS_WG) {
        const int g = (int)gidx[d] - 1, l0 = glev[g] + 1 + (d - ghead[g]) * rpc;
        for (int q = 0; q < rpc; q++) lv_out[d * rpc + q] = l0 + q;
    }
    for (int q = tid; q <= nlev; q += EXS_WG) off[q] = 0;
    __syncthreads();                                                         // (gidx is dead from here on: rows_out becomes the lists)
    if (m <= WAVE_ISLAND_ROWS) {
        if (tid == 0) { off[0] = base; off[nlev] = base + m; }
        __syncthreads();
        return true;
    }
    for (int r = tid; r < m; r += EXS_WG) atomicAdd(&off[lv_out[r] + 1], 1);
    __syncthreads();
    int w = 0;
    for (int q = 1 + tid; q <= nlev; q += EXS_WG) w = off[q] > w ? off[q] : w;
    if (w > 0) atomicMax(&C->big_max_width, (uint32_t)w);
    if (tid == 0) off[0] = base;
    __syncthreads();
    block_scan_inclusive<uint32_t>(reinterpret_cast<uint32_t *>(off), reinterpret_cast<uint32_t *>(off), (uint32_t)nlev + 1u, wt);
    for (int r = tid; r < m; r += EXS_WG) rows_out[atomicAdd(&off[lv_out[r]], 1) - base] = r;       // off[lv]: the fill cursor
    __syncthreads();
    for (int hi = nlev; hi > 0; hi -= EXS_WG) {                               // cursors (= the next level's start) back to starts
        const int q = hi - tid;
        const int v = q >= 1 ? off[q - 1] : 0;
        __syncthreads();
        if (q >= 1) off[q] = v;
        __syncthreads();
    }
    if (tid == 0) off[0] = base;
    __syncthreads();
    return true;
}
// the one-lane walk of st_levels for one island (the fallback of levels_coop: an island too large for its LDS)
__device__ __forceinline__ void levels_one_lane(int isl, int k, const int *con_off, const int *cb1, const int *cb2, int rpc, const int *big,
                                                int *lev_count, int *lev_off, int *lev_rows, int *row_level, int *last, ExactCounts *C)
{
    const int base = big[isl] - k, m = (con_off[isl + 1] - con_off[isl]) * rpc;
    int *lv_out = row_level + base, *off = lev_off + base + k, *rows_out = lev_rows + base;
    int nlev = 0, r = 0, gb1 = -2, gb2 = -2, glast = -1;
    for (int d = con_off[isl]; d < con_off[isl + 1]; d++) {
        const int b1 = cb1[d], b2 = cb2[d];
        if (b1 != gb1 || b2 != gb2) {
            if (gb1 >= 0) { last[gb1] = glast; if (gb2 >= 0) last[gb2] = glast; }
            int lv = last[b1];
            if (b2 >= 0 && last[b2] > lv) lv = last[b2];
            glast = lv; gb1 = b1; gb2 = b2;
        }
        for (int q = 0; q < rpc; q++, r++) lv_out[r] = ++glast;
        if (glast + 1 > nlev) nlev = glast + 1;
    }
    for (int d = con_off[isl]; d < con_off[isl + 1]; d++) {
        last[cb1[d]] = -1;
        if (cb2[d] >= 0) last[cb2[d]] = -1;
    }
    lev_count[k] = nlev;
    if (m <= WAVE_ISLAND_ROWS) { off[0] = base; off[nlev] = base + m; return; }
    for (int q = 0; q <= nlev; q++) off[q] = 0;
    for (int q = 0; q < m; q++) off[lv_out[q] + 1]++;
    int w = 0;
    for (int q = 1; q <= nlev; q++) w = off[q] > w ? off[q] : w;
    atomicMax(&C->big_max_width, (uint32_t)w);
    off[0] = base;
    for (int q = 0; q < nlev; q++) off[q + 1] += off[q];
    for (int q = 0; q < m; q++) rows_out[off[lv_out[q]]++ - base] = q;
    for (int q = nlev; q > 0; q--) off[q] = off[q - 1];
    off[0] = base;
}
// every island of the big list that st_levels left out (more than LEVELS_COOP_ROWS rows), by the calling workgroup(s)
__device__ __forceinline__ void st_levels_coop(const int *con_off, const int *body_off, const int *bodies, const int *cb1, const int *cb2,
                                               const ExactCaps &cap, int rpc, const int *big, const int *big_list, int *lev_count,
                                               int *lev_off, int *lev_rows, int *row_level, int *last, ExactCounts *C,
                                               unsigned char *lds, size_t lds_bytes, uint32_t *wt, uint32_t first, uint32_t step)
{
    if (C->overflow) return;               // (capacities are being grown: the tick runs again)
    const uint32_t nbig = C->nbig;
    if (nbig > (uint32_t)LEVELS_COOP_MAX_ISLANDS) return;       // (st_levels has built them all, a lane each)
    for (uint32_t k = first; k < nbig; k += step) {
        const int isl = big_list[k];
        const int m = (con_off[isl + 1] - con_off[isl]) * rpc;
        if (!levels_by_workgroup(m, (int)nbig, LEVELS_COOP_ROWS)) continue;
        if (!levels_coop(isl, (int)k, con_off, body_off, bodies, cb1, cb2, rpc, big, lev_count, lev_off, lev_rows, row_level, last, C, lds,
                         lds_bytes, wt)) {
            if (threadIdx.x == 0) levels_one_lane(isl, (int)k, con_off, cb1, cb2, rpc, big, lev_count, lev_off, lev_rows, row_level, last, C);
            __syncthreads();
        }
    }
}
__global__ __launch_bounds__(EXS_WG) void ex_levels_coop(const int *__restrict__ con_off, const int *__restrict__ body_off,
                                                         const int *__restrict__ bodies, const int *__restrict__ cb1,
                                                         const int *__restrict__ cb2, ExactCaps cap, int rpc, const int *__restrict__ big,
                                                         const int *__restrict__ big_list, int *__restrict__ lev_count,
                                                         int *__restrict__ lev_off, int *__restrict__ lev_rows, int *__restrict__ row_level,
                                                         int *__restrict__ last, ExactCounts *C, uint32_t lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char coop_lds[];
    __shared__ uint32_t wt[EXS_WG / 64];
    st_levels_coop(con_off, body_off, bodies, cb1, cb2, cap, rpc, big, big_list, lev_count, lev_off, lev_rows, row_level, last, C, coop_lds,
                   lds_bytes, wt, blockIdx.x, gridDim.x);
}

// B.stamps: wall_clock64() (100 MHz) after every stage, front kernel from [0], back kernel from [32]; DMX_EXS_TIMING=1 prints the
// stage averages when the batch is destroyed
#define EXS_STAMP() do { if (threadIdx.x == 0) stamps[sk] = wall_clock64(); sk++; } while (0)

// grid fill (fill_grid's memsets + bp_insert) and stages 1-3: pairs, involved bodies, islands' roots.
// LG: the grid is built in LDS as cell runs (LdsGridWalk) instead of in the batch's bucket table -- the records sorted by cell, the
// cells' starts and the staged partners fit (exact_small_lds_bytes); the table in device memory is then not touched at all (nothing
// after this kernel reads it: the zones' rebuild fills it itself), the records still go to G.rec for the narrowphase.
template <class T, bool LG>
__global__ __launch_bounds__(EXS_WG) void ex_small_front(T *S, const uint8_t *gtype, int64_t n, int64_t n_active, GridParams<T> G,
                                                         ExactBuffers<T> B, ExactCaps cap, ExactCounts *host_counts, uint32_t *host_flags,
                                                         uint32_t seq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char exs_lds[];
    __shared__ uint64_t wt[EXS_WG / 64];
    const uint32_t tid = threadIdx.x;
    ExactCounts *C = B.counts;
    uint64_t *stamps = B.stamps; int sk = 0;
    const uint32_t cells = G.mask + 1u;
    CellRec<T> *cr = reinterpret_cast<CellRec<T> *>(exs_lds);                              // [n] records sorted by cell
    uint32_t *start = reinterpret_cast<uint32_t *>(cr + (LG ? n : 0));                     // [cells + 1]
    uint16_t *staged = reinterpret_cast<uint16_t *>(start + (LG ? cells + 1u : 0u));       // [n EXS_PARTNERS] partners above a body
    uint32_t *own_l = reinterpret_cast<uint32_t *>(staged + (LG ? (size_t)n * EXS_PARTNERS : 0));    // [n] partners above it, counted by several lanes
    uint8_t *any_l = reinterpret_cast<uint8_t *>(own_l + (LG ? n : 0));                     // [n] in any pair at all
    EXS_STAMP();
    if (LG) { for (uint32_t k = tid; k <= cells; k += EXS_WG) start[k] = 0u; }
    else { for (uint32_t k = tid; k <= G.mask; k += EXS_WG) G.count[k] = 0u; }
    if (tid < (uint32_t)BPF_COUNT) G.flags[tid] = 0u;
    if (tid < sizeof(ExactCounts) / 4) ((uint32_t *)C)[tid] = 0u;
    __syncthreads(); EXS_STAMP();
    if (LG) {
        // grid_insert's work with the bucket replaced by a count: the cell's run is laid out below
        for (int64_t i = tid; i < n; i += EXS_WG) {
            const uint8_t gt = gtype[i];
            if (gt == GEOM_NONE) continue;
            S[slab_ix(C_BPR, i)] = bound_radius<T>(gt, S, i);
            GridRec<T> r;
            body_aabb<T>(S, gtype, i, r.lo, r.hi);
            r.ix = (int)floor((double)(S[slab_ix(C_POS + 0, i)] * G.inv_cell));
            r.iz = (int)floor((double)(S[slab_ix(C_POS + 2, i)] * G.inv_cell));
            G.rec[i] = r;
            atomicAdd(&start[cell_hash(r.ix, r.iz, G.mask, G.xbits)], 1u);
        }
    } else {
        for (int64_t i = tid; i < n; i += EXS_WG) grid_insert<T>(S, gtype, i, G);      // ghosts included
    }
    if (G.hull_n > 0) {
        // convex bodies: the exact AABB over the bounding sphere's box (bp_convex_aabb's job in the stage-per-launch form), a
        // wavefront per hull, behind a barrier: another wave's thread wrote the record just now
        __syncthreads();
        for (int64_t i = tid >> 6; i < n; i += EXS_WG / 64) {
            if (gtype[i] != GEOM_CONVEX) continue;
            const V3<T> x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
            const M3<T> R = quat_to_R(Q4<T>{ S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)], S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] });
            T lo[3], hi[3];
            wave_hull_aabb<T>(x, R, G.hull, G.hull_n, (int)(tid & 63), lo, hi);
            const int l = (int)(tid & 63);
            if (l < 3) {
                const T lv = l == 0 ? lo[0] : (l == 1 ? lo[1] : lo[2]), hv = l == 0 ? hi[0] : (l == 1 ? hi[1] : hi[2]);
                G.rec[i].lo[l] = lv; G.rec[i].hi[l] = hv;
            }
        }
    }
    __syncthreads();
    if (LG) {
        // counts -> run ends (inclusive scan in place) -> every body takes the slot below its cell's end: start[h] is the run's
        // first slot when all have, and start[h + 1] -- the next run's first -- its end
        block_scan_inclusive<uint32_t>(start, start, cells, reinterpret_cast<uint32_t *>(wt));
        if (tid == 0) start[cells] = start[cells - 1];
        __syncthreads();
        for (int64_t i = tid; i < n; i += EXS_WG) {
            const uint8_t gt = gtype[i];
            if (gt == GEOM_NONE) continue;
            const GridRec<T> r = G.rec[i];       // (this thread's own write, or -- a hull's exact box -- another wave's before the barrier)
            const uint32_t at = atomicSub(&start[cell_hash(r.ix, r.iz, G.mask, G.xbits)], 1u) - 1u;
            CellRec<T> o;
#pragma unroll
            for (int a = 0; a < 3; a++) { o.lo[a] = r.lo[a]; o.hi[a] = r.hi[a]; }
            o.id = (uint32_t)i | ((uint32_t)gt << 16);
            o.key = ((uint32_t)r.ix & 0xffffu) | ((uint32_t)r.iz << 16);
            cr[at] = o;
        }
        __syncthreads();
    }
    EXS_STAMP();
    const LdsGridWalk<T> lw{ G.rec, cr, start, gtype, G.mask, G.xbits, G.class_pairs };
    const GridWalk<T> gw{ S, gtype, G };
    // a scene of far fewer bodies than the workgroup has lanes (a pile in the pen): K lanes share a body's walk -- in a crowded pen
    // every body's nine runs hold every other body, and one lane's walk over them is the stage's duration
    uint32_t K = 1;
    while (LG && K < 32u && (uint64_t)n_active * 2u * K <= (uint64_t)EXS_WG) K *= 2u;
    if (LG && K > 1u) {
        for (int64_t i = tid; i < n_active; i += EXS_WG) { own_l[i] = 0u; any_l[i] = 0; }
        __syncthreads();
        for (uint64_t idx = tid; idx < (uint64_t)n_active * K; idx += EXS_WG) {
            const int64_t i = (int64_t)(idx / K);
            if (gtype[i] == GEOM_NONE) continue;
            lw(i, [&](int64_t j) {
                any_l[i] = 1;
                if (j >= n_active) {
                    if (atomicOr(&C->cross, 1u) == 0u) { C->cross_a = (uint32_t)i; C->cross_b = (uint32_t)j; }
                    const uint32_t at = atomicAdd(&C->ncross, 1u);
                    if (at < EX_CROSS_CAP) { B.cross_list[2 * at] = (int32_t)i; B.cross_list[2 * at + 1] = (int32_t)j; }
                } else if (j > i) {
                    const uint32_t r = atomicAdd(&own_l[i], 1u);
                    if (r < (uint32_t)EXS_PARTNERS) staged[(size_t)i * EXS_PARTNERS + r] = (uint16_t)j;
                }
            }, (uint32_t)(idx % K), K);
        }
        __syncthreads();
        for (int64_t i = tid; i < n_active; i += EXS_WG) {
            uint32_t any = any_l[i];
            if (gtype[i] != GEOM_NONE && G.n_static > 0) {         // (st_pair_count's rule for bodies at static boxes)
                const GridRec<T> me = G.rec[i];
                int ns = 0;
                for (int s2 = 0; s2 < G.n_static; s2++)
                    if (rec_meets_static(me, G.sbox + s2 * SBOX_REALS)) ns++;
                if (G.static_fast ? (ns >= 2 || (ns >= 1 && G.plane_on)) : ns >= 1) any = 1;
            }
            B.pc[i] = ((uint64_t)own_l[i] << 32) | any;
            B.inpair[i] = (uint8_t)any;
        }
    }
    else if (LG) st_pair_count<T>(lw, gtype, n_active, G, B.pc, B.inpair, C, B.cross_list, tid, EXS_WG, staged);
    else    st_pair_count<T, GridWalk<T>, int32_t, EX_STAGE_PARTNERS>(gw, gtype, n_active, G, B.pc, B.inpair, C, B.cross_list, tid, EXS_WG, B.stage);
    __syncthreads(); EXS_STAMP();
    block_scan_inclusive<uint64_t>(B.pc, B.inc, (uint32_t)n_active, wt); EXS_STAMP();
    if (LG) st_pair_write<T>(lw, n_active, G, B.pc, B.inc, B.pairs, B.inv, B.parent, cap, C, tid, EXS_WG, staged);
    else    st_pair_write<T, GridWalk<T>, int32_t, EX_STAGE_PARTNERS>(gw, n_active, G, B.pc, B.inc, B.pairs, B.inv, B.parent, cap, C, tid, EXS_WG, B.stage);
    __syncthreads(); EXS_STAMP();
    st_unite(B.pairs, B.pc, B.inc, B.parent, C, tid, EXS_WG);
    __syncthreads(); EXS_STAMP();
    st_flatten(B.parent, B.root, B.rf, cap, C, tid, EXS_WG);
    __syncthreads(); EXS_STAMP();
    block_scan_inclusive<uint32_t>(B.rf, B.rinc, cap.inv, reinterpret_cast<uint32_t *>(wt)); EXS_STAMP();
    publish_counts(C, G.flags, host_counts, host_flags, seq);    // (the scan above ended on a barrier)
}

// stages 5-10 (the narrowphase ran in between): entries sorted by island, joints in creation order, level schedules.
// ITEMS entries per thread (2, 4 or 8: up to 8192 entries).
template <class T, int ITEMS>
__global__ __launch_bounds__(EXS_WG) void ex_small_back(ExactBuffers<T> B, ExactCaps cap, int rpc, int big_rows, const uint32_t *flags,
                                                        StepDiag *diag, ExactCounts *host_counts, uint32_t *host_flags, uint32_t seq,
                                                        int lds_slots, uint32_t lds_bytes)
{
    // lds_slots > 0: the level schedules' per-slot `last` array lives in LDS for this launch (that many slots, all idle) -- the
    // stage is one lane per island walking a chain of read-modify-writes of it, a round trip to L2 each otherwise.  The same
    // lds_bytes of dynamic LDS then serve the large islands' schedules (levels_coop), which use B.last in device memory.
    extern __shared__ __attribute__((aligned(16))) int32_t last_l[];
    for (int k = threadIdx.x; k < lds_slots; k += EXS_WG) last_l[k] = -1;        // (barriers follow before its first use)
    using sort_t = rocprim::block_radix_sort<uint32_t, EXS_WG, ITEMS>;
    __shared__ typename sort_t::storage_type sort_storage;
    __shared__ uint64_t wt[EXS_WG / 64];
    const uint32_t tid = threadIdx.x;
    ExactCounts *C = B.counts;
    uint64_t *stamps = B.stamps + 32; int sk = 0;
    EXS_STAMP();
    const uint32_t ninv = C->overflow ? 0u : C->ninv, np = C->overflow ? 0u : C->npairs;
    const uint32_t ne = cap.entries();
    const uint32_t ni = B.rinc[cap.inv - 1];            // islands; the padding sorts behind them all with key ni
    unsigned ebits = 1, kbits = 1;
    while ((1u << ebits) < ne) ebits++;                 // an entry's index: the low bits, so the keys are in entry order to begin with
    while ((1u << kbits) <= ni) kbits++;
    uint32_t keys[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t e = tid * ITEMS + j;
        uint32_t key = 0xffffffffu;
        if (e < ne) {
            key = entry_key(e, B.pairs, B.pc, B.inc, B.root, B.rinc, cap, ninv, np);
            key = ((key < ni ? key : ni) << ebits) | e;
        }
        keys[j] = key;
    }
    if (tid == 0) C->ni = ni;
    __syncthreads(); EXS_STAMP();
    // LSD radix sort is stable: sorting the island bits alone leaves each island's entries in entry (= joint creation) order
    sort_t().sort(keys, sort_storage, ebits, ebits + kbits);
    __syncthreads(); EXS_STAMP();
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t t = tid * ITEMS + j;
        if (t < ne) {
            uint32_t key = (keys[j] >> ebits) & ((1u << kbits) - 1u);
            const uint32_t e = keys[j] & ((1u << ebits) - 1u);
            if (key >= ni) key = cap.inv;
            B.keys_s[t] = key; B.vals_s[t] = e;
            B.sc[t] = gathered(key, e, B.cc, cap);
        }
    }
    __syncthreads(); EXS_STAMP();
    block_scan_inclusive<uint64_t>(B.sc, B.sinc, ne, wt); EXS_STAMP();
    st_bounds(B.keys_s, B.vals_s, B.cc, B.sinc, cap, B.body_off, B.con_off, B.row_off, C, tid, EXS_WG);
    __syncthreads(); EXS_STAMP();
    st_fill(B.keys_s, B.vals_s, B.sinc, B.cc, B.inv, B.pairs, B.con_off, cap, rpc, B.bodies, B.cb1, B.cb2, B.csrc, B.crow, tid, EXS_WG);
    st_bigflags(B.con_off, B.body_off, cap, rpc, big_rows, B.bg, C, tid, EXS_WG);
    __syncthreads(); EXS_STAMP();
    block_scan_inclusive<uint64_t>(B.bg, B.binc, cap.inv, wt); EXS_STAMP();
    st_levels(B.con_off, B.body_off, B.cb1, B.cb2, B.bg, B.binc, cap, rpc, B.big, B.big_list, B.lev_count, B.lev_off, B.lev_rows,
              B.row_level, lds_slots > 0 ? last_l : B.last, C, tid, EXS_WG, LEVELS_COOP_ROWS);
    __syncthreads();
    st_levels_coop(B.con_off, B.body_off, B.bodies, B.cb1, B.cb2, cap, rpc, B.big, B.big_list, B.lev_count, B.lev_off, B.lev_rows,
                   B.row_level, B.last, C, reinterpret_cast<unsigned char *>(last_l), lds_bytes, reinterpret_cast<uint32_t *>(wt), 0u, 1u);
    __syncthreads(); EXS_STAMP();
    if (tid == 0) {
        diag->contacts = 0ull; diag->residual = 0.0;       // the island kernels add to it next
        // careful_tick's speculative launches (solve_island_wg<64> over the capacity, the fused step behind it) read this
        C->spec_ok = (C->overflow == 0u && C->bp_overflow == 0u && flags[BPF_OVERFLOW] == 0u && C->cross == 0u && C->nbig == C->ni &&
                      C->big_max_width <= 64u && C->big_max_bodies <= EX_SPEC_ISLAND_BODIES) ? 1u : 0u;
    }
    __syncthreads();
    publish_counts(C, flags, host_counts, host_flags, seq);
}

__global__ void ex_fill_i32(int32_t *p, int32_t v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

inline unsigned grid_for(size_t n) { size_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

}  // namespace

size_t exact_temp_bytes(const ExactCaps &cap, int64_t n_active)
{
    size_t a = 0, b = 0, c = 0, d = 0;
    const size_t ne = (size_t)cap.entries();
    (void)rocprim::inclusive_scan(nullptr, a, (uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)n_active, rocprim::plus<uint64_t>());
    (void)rocprim::inclusive_scan(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)cap.inv, rocprim::plus<uint32_t>());
    (void)rocprim::inclusive_scan(nullptr, c,
                                  rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u),
                                                                   GatherOp{ nullptr, nullptr, nullptr, cap }),
                                  (uint64_t *)nullptr, ne, rocprim::plus<uint64_t>());
    (void)rocprim::radix_sort_pairs(nullptr, d, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, ne, 0, 32);
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    m = m > d ? m : d;
    return m + 256;
}

hipError_t exact_init_last(int32_t *last, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(ex_fill_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, last, -1, (size_t)n);
    return hipGetLastError();
}

#define EX_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

template <class T>
hipError_t launch_exact_pairs(const T *S, const uint8_t *gtype, int64_t n_active, const GridParams<T> &G,
                              const ExactBuffers<T> &B, const ExactCaps &cap, hipStream_t st)
{
    size_t tb = B.temp_bytes;          // (B.counts was zeroed by the caller, with the grid)
    // a few thousand bodies: a wavefront per body (a lane per candidate); more: a lane per body
    static const bool wave_on = [] { const char *e = getenv("DMX_PAIR_COUNT_WAVE"); return !(e && atoi(e) == 0); }();
    if (wave_on && n_active <= 8192 && G.n_static <= 64)
        hipLaunchKernelGGL((ex_pair_count_wave<T>), dim3((unsigned)((n_active + 3) / 4)), dim3(256), 0, st, S, gtype, n_active, G, B.pc, B.inpair, B.counts,
                           B.cross_list, B.stage);
    else
        hipLaunchKernelGGL((ex_pair_count<T>), dim3((unsigned)((n_active + 255) / 256)), dim3(256), 0, st, S, gtype, n_active, G, B.pc, B.inpair, B.counts, B.cross_list, B.stage);
    EX_TRY(rocprim::inclusive_scan(B.temp, tb, B.pc, B.inc, (size_t)n_active, rocprim::plus<uint64_t>(), st));
    hipLaunchKernelGGL((ex_pair_write<T>), dim3((unsigned)((n_active + 255) / 256)), dim3(256), 0, st, S, gtype, n_active, G, B.pc, B.inc,
                       B.pairs, B.inv, B.parent, cap, B.counts, B.stage);
    return hipGetLastError();
}

// the staged pipeline's last launch: the count record and the flags into pinned host memory (publish_counts), so that the host
// watches for the sequence number instead of copying the record back and synchronising the stream (28 -> 11 us of idle device
// per exact tick)
__global__ __launch_bounds__(64) void ex_publish(const ExactCounts *C, const uint32_t *flags, ExactCounts *host_counts, uint32_t *host_flags, uint32_t seq)
{
    publish_counts(C, flags, host_counts, host_flags, seq);
}
hipError_t launch_exact_publish(const ExactCounts *counts, const uint32_t *flags, ExactCounts *host_counts, uint32_t *host_flags, uint32_t seq,
                                hipStream_t st)
{
    hipLaunchKernelGGL(ex_publish, dim3(1), dim3(64), 0, st, counts, flags, host_counts, host_flags, seq);
    return hipGetLastError();
}

template <class T>
hipError_t launch_exact_group(const T *S, const uint8_t *gtype, int64_t n_active, const GridParams<T> &G, const StepParams<T> &P,
                              const ExactBuffers<T> &B, const ExactCaps &cap, int rpc, int big_rows, hipStream_t st)
{
    const size_t ne = (size_t)cap.entries();
    size_t tb = B.temp_bytes;
    hipLaunchKernelGGL(ex_unite, dim3(grid_for(cap.pairs)), dim3(256), 0, st, B.pairs, B.pc, B.inc, B.parent, B.counts);
    hipLaunchKernelGGL(ex_flatten, dim3(grid_for(cap.inv)), dim3(256), 0, st, B.parent, B.root, B.rf, cap, B.counts);
    tb = B.temp_bytes;
    EX_TRY(rocprim::inclusive_scan(B.temp, tb, B.rf, B.rinc, (size_t)cap.inv, rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL((ex_narrow<T>), dim3((unsigned)((ne + 63) / 64)), dim3(64), 0, st, S, gtype, B.inv, B.pairs, G.rec, P, cap,
                       B.gpos, B.gnormal, B.gdepth, B.cc, B.counts, SortKeyArgs{ B.pc, B.inc, B.root, B.rinc, B.keys, B.vals });
    if (P.hull_n > 0) {
        hipLaunchKernelGGL((ex_narrow_convex<T>), dim3((unsigned)std::min<size_t>((ne + 3) / 4, 65535)), dim3(256), 0, st, S, gtype, B.inv,
                           B.pairs, G.rec, P, cap, B.gpos, B.gnormal, B.gdepth, B.cc, B.counts, 1);
        hipLaunchKernelGGL((ex_narrow_hull_pairs<T>), dim3((unsigned)std::min<size_t>(std::max<size_t>(cap.pairs, 1), 4096)), dim3(256), 0, st, S, gtype,
                           B.pairs, G.rec, P, cap, B.gpos, B.gnormal, B.gdepth, B.cc, B.counts);
    }
    int bits = 1;
    while ((1u << bits) <= cap.inv && bits < 32) bits++;        // keys are island numbers < cap.inv and the padding key cap.inv
    tb = B.temp_bytes;
    EX_TRY(rocprim::radix_sort_pairs(B.temp, tb, B.keys, B.keys_s, B.vals, B.vals_s, ne, 0, (unsigned)bits, st));
    tb = B.temp_bytes;
    EX_TRY(rocprim::inclusive_scan(B.temp, tb,
                                   rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u),
                                                                    GatherOp{ B.keys_s, B.vals_s, B.cc, cap }),
                                   B.sinc, ne, rocprim::plus<uint64_t>(), st));
    hipLaunchKernelGGL(ex_bounds, dim3(grid_for(ne)), dim3(256), 0, st, B.keys_s, B.vals_s, B.cc, B.sinc, cap, B.body_off, B.con_off, B.row_off,
                       B.counts);
    hipLaunchKernelGGL(ex_fill, dim3(grid_for(std::max<size_t>(ne, cap.inv))), dim3(256), 0, st, B.keys_s, B.vals_s, B.sinc, B.cc, B.inv,
                       B.pairs, B.con_off, cap, rpc, B.bodies, B.cb1, B.cb2, B.csrc, B.crow, B.body_off, big_rows, B.bg, B.counts);
    tb = B.temp_bytes;
    EX_TRY(rocprim::inclusive_scan(B.temp, tb, B.bg, B.binc, (size_t)cap.inv, rocprim::plus<uint64_t>(), st));
    hipLaunchKernelGGL(ex_levels, dim3((unsigned)(((size_t)cap.inv + 63) / 64)), dim3(64), 0, st, B.con_off, B.body_off, B.cb1, B.cb2, B.bg, B.binc,
                       cap, rpc, B.big, B.big_list, B.lev_count, B.lev_off, B.lev_rows, B.row_level, B.last, B.counts, LEVELS_COOP_ROWS);
    {
        // the islands that left out (a pile in the reference's pen: thousands of contacts in one island), a workgroup each
        const size_t lds = 128 * 1024;
        EX_TRY(hipFuncSetAttribute((const void *)&ex_levels_coop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ex_levels_coop, dim3(32), dim3(EXS_WG), lds, st, B.con_off, B.body_off, B.bodies, B.cb1, B.cb2, cap, rpc, B.big, B.big_list,
                           B.lev_count, B.lev_off, B.lev_rows, B.row_level, B.last, B.counts, (uint32_t)lds);
    }
    return hipGetLastError();
}

// LDS the one-workgroup front kernel needs to keep the whole grid there (LdsGridWalk): records sorted by cell, cell starts,
// staged partners.  0: does not fit (or DMX_SMALL_LDS_GRID=0) -- the kernel walks the bucket table in device memory.
size_t exact_small_lds_bytes(int64_t n, uint32_t grid_mask, size_t real_bytes)
{
    static const bool on = [] { const char *e = getenv("DMX_SMALL_LDS_GRID"); return !(e && atoi(e) == 0); }();
    if (!on || n > 65535) return 0;
    const size_t rec = real_bytes == 4 ? sizeof(CellRec<float>) : sizeof(CellRec<double>);
    size_t b = (size_t)n * rec + ((size_t)grid_mask + 2) * 4 + (size_t)n * EXS_PARTNERS * 2 + (size_t)n * 5;
    b = (b + 15) & ~(size_t)15;
    return b <= 150 * 1024 ? b : 0;        // of the CU's 160 KiB; the kernel's static LDS is a few hundred bytes
}

bool exact_small_fits(int64_t n, uint32_t grid_mask, const ExactCaps &cap)
{
    return n <= 8192 && grid_mask < 32768u && exact_back_fits(cap);
}
bool exact_back_fits(const ExactCaps &cap)
{
    return cap.entries() <= (uint32_t)(EXS_WG * 8) && cap.inv <= 8192u;
}

// islands' roots of the pairs launch_exact_pairs left (stages 3: union-find, flatten, root numbering) -- the part of
// launch_exact_group that ex_small_front also ends with, for a caller that goes on with launch_exact_small_group
template <class T>
hipError_t launch_exact_roots(const ExactBuffers<T> &B, const ExactCaps &cap, hipStream_t st)
{
    size_t tb = B.temp_bytes;
    hipLaunchKernelGGL(ex_unite, dim3(grid_for(cap.pairs)), dim3(256), 0, st, B.pairs, B.pc, B.inc, B.parent, B.counts);
    hipLaunchKernelGGL(ex_flatten, dim3(grid_for(cap.inv)), dim3(256), 0, st, B.parent, B.root, B.rf, cap, B.counts);
    EX_TRY(rocprim::inclusive_scan(B.temp, tb, B.rf, B.rinc, (size_t)cap.inv, rocprim::plus<uint32_t>(), st));
    return hipGetLastError();
}
template hipError_t launch_exact_roots<float>(const ExactBuffers<float> &, const ExactCaps &, hipStream_t);
template hipError_t launch_exact_roots<double>(const ExactBuffers<double> &, const ExactCaps &, hipStream_t);

template <class T>
hipError_t launch_exact_small_front(T *S, const uint8_t *gtype, int64_t n, int64_t n_active, const GridParams<T> &G, const ExactBuffers<T> &B,
                                    const ExactCaps &cap, ExactCounts *host_counts, uint32_t *host_flags, uint32_t seq, hipStream_t st)
{
    const size_t lds = exact_small_lds_bytes(n, G.mask, sizeof(T));
    if (lds != 0) {
        if (lds > 64 * 1024) {       // (the default limit on dynamic LDS; a table write, and scenes this size are the rarer ones)
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&ex_small_front<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) return ea;
        }
        hipLaunchKernelGGL((ex_small_front<T, true>), dim3(1), dim3(EXS_WG), lds, st, S, gtype, n, n_active, G, B, cap, host_counts, host_flags, seq);
    } else {
        hipLaunchKernelGGL((ex_small_front<T, false>), dim3(1), dim3(EXS_WG), 0, st, S, gtype, n, n_active, G, B, cap, host_counts, host_flags, seq);
    }
    return hipGetLastError();
}

template <class T>
hipError_t launch_exact_small_group(const T *S, const uint8_t *gtype, const GridParams<T> &G, const StepParams<T> &P, const ExactBuffers<T> &B,
                                    const ExactCaps &cap, int rpc, int big_rows, StepDiag *diag, ExactCounts *host_counts,
                                    uint32_t *host_flags, uint32_t seq, int64_t n_slots, hipStream_t st)
{
    const size_t ne = (size_t)cap.entries();
    // dynamic LDS beside the block sort's storage: the per-slot `last` array of the one-lane schedules (up to 8192 slots), then the
    // large islands' group arrays (levels_coop)
    const int ls = (n_slots > 0 && n_slots <= 8192) ? (int)n_slots : 0;
    const size_t lds = 64 * 1024;
    {
        const void *fn = ne <= 2 * EXS_WG ? (const void *)&ex_small_back<T, 2> : ne <= 4 * EXS_WG ? (const void *)&ex_small_back<T, 4> : (const void *)&ex_small_back<T, 8>;
        const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL((ex_narrow<T>), dim3((unsigned)((ne + 63) / 64)), dim3(64), 0, st, S, gtype, B.inv, B.pairs, G.rec, P, cap,
                       B.gpos, B.gnormal, B.gdepth, B.cc, B.counts, SortKeyArgs{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr });
    if (P.hull_n > 0) {
        hipLaunchKernelGGL((ex_narrow_convex<T>), dim3((unsigned)std::min<size_t>((ne + 3) / 4, 65535)), dim3(256), 0, st, S, gtype, B.inv,
                           B.pairs, G.rec, P, cap, B.gpos, B.gnormal, B.gdepth, B.cc, B.counts, 1);
        hipLaunchKernelGGL((ex_narrow_hull_pairs<T>), dim3((unsigned)std::min<size_t>(std::max<size_t>(cap.pairs, 1), 4096)), dim3(256), 0, st, S, gtype,
                           B.pairs, G.rec, P, cap, B.gpos, B.gnormal, B.gdepth, B.cc, B.counts);
    }
    const uint32_t *flags = G.flags;
    if (ne <= 2 * EXS_WG)
        hipLaunchKernelGGL((ex_small_back<T, 2>), dim3(1), dim3(EXS_WG), lds, st, B, cap, rpc, big_rows, flags, diag, host_counts, host_flags, seq, ls, (uint32_t)lds);
    else if (ne <= 4 * EXS_WG)
        hipLaunchKernelGGL((ex_small_back<T, 4>), dim3(1), dim3(EXS_WG), lds, st, B, cap, rpc, big_rows, flags, diag, host_counts, host_flags, seq, ls, (uint32_t)lds);
    else
        hipLaunchKernelGGL((ex_small_back<T, 8>), dim3(1), dim3(EXS_WG), lds, st, B, cap, rpc, big_rows, flags, diag, host_counts, host_flags, seq, ls, (uint32_t)lds);
    return hipGetLastError();
}

#define DMX_EXS_INST(T)                                                                                                                  \
    template hipError_t launch_exact_small_front<T>(T *, const uint8_t *, int64_t, int64_t, const GridParams<T> &, const ExactBuffers<T> &, \
                                                    const ExactCaps &, ExactCounts *, uint32_t *, uint32_t, hipStream_t);                \
    template hipError_t launch_exact_small_group<T>(const T *, const uint8_t *, const GridParams<T> &, const StepParams<T> &,            \
                                                    const ExactBuffers<T> &, const ExactCaps &, int, int, StepDiag *, ExactCounts *,     \
                                                    uint32_t *, uint32_t, int64_t, hipStream_t);
DMX_EXS_INST(float)
DMX_EXS_INST(double)

template hipError_t launch_exact_pairs<float>(const float *, const uint8_t *, int64_t, const GridParams<float> &, const ExactBuffers<float> &,
                                              const ExactCaps &, hipStream_t);
template hipError_t launch_exact_pairs<double>(const double *, const uint8_t *, int64_t, const GridParams<double> &, const ExactBuffers<double> &,
                                               const ExactCaps &, hipStream_t);
template hipError_t launch_exact_group<float>(const float *, const uint8_t *, int64_t, const GridParams<float> &, const StepParams<float> &,
                                              const ExactBuffers<float> &, const ExactCaps &, int, int, hipStream_t);
template hipError_t launch_exact_group<double>(const double *, const uint8_t *, int64_t, const GridParams<double> &, const StepParams<double> &,
                                               const ExactBuffers<double> &, const ExactCaps &, int, int, hipStream_t);

// HIP loads a translation unit's code object at the first launch of one of its kernels -- a couple of milliseconds each, which an
// interactive caller would meet as a hitch at the first tick that needs the exact pipeline.  dmxBatchCreate asks for one
// kernel's attributes per unit instead (dmx_preload_code, dmx_batch.cpp): the load happens there.
hipError_t dmx_touch_exact(int real_bytes)
{
    // (the unit's code object, and -- what costs more -- each kernel's own first-use set-up: every kernel an exact tick or a fused
    //  tick may launch, in the batch's precision)
    hipFuncAttributes a;
    hipError_t e = hipSuccess;
    auto touch = [&](const void *k) { const hipError_t r = hipFuncGetAttributes(&a, k); if (r != hipSuccess) e = r; };
    touch((const void *)&ex_fill_i32);
    touch((const void *)&ex_unite);
    touch((const void *)&ex_flatten);
    touch((const void *)&ex_levels);
    touch((const void *)&ex_levels_coop);
    touch((const void *)&ex_publish);
    touch((const void *)&ex_bounds);
    touch((const void *)&ex_fill);
    if (real_bytes == 4) {
        touch((const void *)&ex_small_front<float, true>);
        touch((const void *)&ex_small_front<float, false>);
        touch((const void *)&ex_small_back<float, 2>);
        touch((const void *)&ex_small_back<float, 4>);
        touch((const void *)&ex_small_back<float, 8>);
        touch((const void *)&ex_narrow<float>);
        touch((const void *)&ex_pair_count<float>);
        touch((const void *)&ex_pair_count_wave<float>);
        touch((const void *)&ex_pair_write<float>);
    } else {
        touch((const void *)&ex_small_front<double, true>);
        touch((const void *)&ex_small_front<double, false>);
        touch((const void *)&ex_small_back<double, 2>);
        touch((const void *)&ex_small_back<double, 4>);
        touch((const void *)&ex_small_back<double, 8>);
        touch((const void *)&ex_narrow<double>);
        touch((const void *)&ex_pair_count<double>);
        touch((const void *)&ex_pair_count_wave<double>);
        touch((const void *)&ex_pair_write<double>);
    }
    return e;
}

}  // namespace dmx
