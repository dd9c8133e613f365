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
    const int nc = wave_convex_plane<T>(x, R, S[slab_ix(C_SIDES + 0, i)], P, maxc, lane, [&](int rank, const V3<T> &p, const V3<T> &, T dep) {
        out[4 * rank + 0] = p.x; out[4 * rank + 1] = p.y; out[4 * rank + 2] = p.z; out[4 * rank + 3] = dep; }, P.hull);
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

// dSpaceCollide's AABB test between a hull and a static box as the oracle (and the exact tick) makes it: on the hull's EXACT world box,
// the bounds of its transformed points.  The fused path tests the bounding sphere's box first -- conservative, and free -- and for
// colliders that are exact geometry that changes nothing IN REAL ARITHMETIC: a vertex inside the box means the true boxes overlap.
// In floating point a vertex within rounding of a face can be inside by the collider's arithmetic while the two AABBs, rounded
// their own way, miss each other by an ulp -- one contact in 10^5 a metre from the origin, one scene in a hundred at 7 km in f32
// (scripts/fuzz_hulls_r04.py found it).  So a pair that made contacts is confirmed: at once when a contact vertex lies inside the
// box's AABB (wave_box_convex_walk), by the exact box otherwise (a wavefront's pass over the hull, once in a long while).
template <class T>
__device__ __forceinline__ bool hull_aabb_meets_static(const V3<T> &x, const M3<T> &R, const T *pts, int hull_n, int lane, const T *sb)
{
    T lo[3], hi[3];
    wave_hull_aabb<T>(x, R, pts, hull_n, lane, lo, hi);
    return aabb_meets_static(lo, hi, sb);
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
    const T wide = radius * T(1.0001) + T(8) * RealEps<T>::v() * (tabs(x.x) + tabs(x.y) + tabs(x.z));      // (as in np_convex_static_tile)
    const T lo[3] = { x.x - wide, x.y - wide, x.z - wide }, hi[3] = { x.x + wide, x.y + wide, x.z + wide };
    const int maxc = P.max_contacts < CONVEX_MAXC ? P.max_contacts : CONVEX_MAXC;
    int nc = 0;
    if (P.hull_n > 0) {
        if (P.plane_on)
            nc = wave_convex_plane<T>(x, R, radius, P, maxc, lane, [&](int rank, const V3<T> &p, const V3<T> &nn, T dep) { put_sc(P.sbuf, i, rank, p, nn, dep); }, P.hull);
        for (int s = 0; s < P.n_static; s++) {
            const T *sb = P.sbox + s * SBOX_REALS;
            if (!aabb_meets_static(lo, hi, sb)) continue;
            const V3<T> sx = { sb[SBOX_POS], sb[SBOX_POS + 1], sb[SBOX_POS + 2] };
            M3<T> sR;
            for (int a = 0; a < 3; a++) for (int c2 = 0; c2 < 3; c2++) sR.m[a][c2] = sb[SBOX_R + 3 * a + c2];
            const T sside[3] = { sb[SBOX_SIDE], sb[SBOX_SIDE + 1], sb[SBOX_SIDE + 2] };
            const int base = nc;
            bool meet_exact = false;
            int k = wave_box_convex<T>(sx, sR, sside, x, R, radius, P, maxc, true, lane, [&](int rank, const V3<T> &p, const V3<T> &nn, T dep) {
                if (base + rank < SC_MAXC) put_sc(P.sbuf, i, base + rank, p, nn, dep); }, P.hull, sb + SBOX_LO, &meet_exact);
            if (k > 0 && !meet_exact && !hull_aabb_meets_static<T>(x, R, P.hull, P.hull_n, lane, sb)) k = 0;
            nc += k;
        }
    }
    if (lane == 0) P.scount[i] = nc > SC_MAXC ? SC_MAXC + 1 : nc;
}

// convex hulls, a 64-body tile per workgroup (round 4).  The wavefront-per-body form above is a launch of n wavefronts, each alive
// ~12 us and waiting for two thirds of it (profiles/r04_np_convex_static_experiments.txt): its pose, then the static box, then one
// pass of 64 hull points after another -- every pass a round trip to L2, and the next one not begun before this one's ballot says
// that room is left.  Here sixteen wavefronts share one copy of the hull's points in LDS (15 KB in f32 for the teapot), filled once
// for the 64 bodies of the tile; a wavefront takes four bodies in turn, their poses fetched together by the lanes before the
// first, so a pass costs an LDS read.  The walk itself -- wave_box_convex, wave_convex_plane -- is the same code on the same
// values in the same order: identical contacts.
constexpr int NPC_TILE = 64, NPC_WAVES = 16, NPC_PER_WAVE = NPC_TILE / NPC_WAVES;
constexpr int NPC_SBOX = 64;         // static boxes staged in LDS beside the hull (the rest are read where they lie)
template <class T>
__global__ __launch_bounds__(NPC_WAVES * 64) void np_convex_static_tile(const T *__restrict__ S, const uint8_t *__restrict__ gtype, int64_t n, StepParams<T> P)
{
    extern __shared__ __align__(16) unsigned char npc_raw[];
    T *pts = reinterpret_cast<T *>(npc_raw);                          // [3 hull_n]
    T *sbl = pts + 3 * (size_t)P.hull_n;                              // [min(n_static, NPC_SBOX)][SBOX_REALS]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Everything the tile needs from memory is asked for at once -- what the wavefront form reads one after another, each a round
    // trip: the refusal flags, the bodies' classes, their poses (lanes 8 j .. 8 j + 7 fetch position, quaternion and bounding
    // radius of the wavefront's j-th body), then -- behind one barrier, and only in a tile that holds a hull -- the hull's points
    // and the static boxes into LDS.
    const int64_t i0 = (int64_t)blockIdx.x * NPC_TILE + w * NPC_PER_WAVE;
    T pose = T(0);
    bool live = false;
    {
        const int j = lane >> 3, c = lane & 7;
        const int64_t ib = i0 + j;
        if (j < NPC_PER_WAVE && ib < n) {
            live = gtype[ib] == GEOM_CONVEX && !(P.skip != nullptr && P.skip[ib]);
            const int comp = c < 3 ? C_POS + c : (c < 7 ? C_QUAT + (c - 3) : C_SIDES);
            pose = S[slab_ix(comp, ib)];
        }
    }
    // a chunk in which a body has left its zone is rolled back whole, and a speculative launch the device's record refuses does
    // nothing: the step kernel behind this one returns at once in both cases (step_plane / step_contacts), so its contacts need not be made
    const bool refused = (P.bp_check && P.bp_flags[BPF_VIOLATION] != 0u) || (P.gate != nullptr && *P.gate == 0u);
    const unsigned long long live_mask = __ballot(live);
    const int ns_lds = P.n_static < NPC_SBOX ? P.n_static : NPC_SBOX;
    if (!__syncthreads_or(live_mask != 0ull && !refused) || P.hull_n <= 0) return;      // (block-uniform)
    for (int e = threadIdx.x; e < 3 * P.hull_n; e += NPC_WAVES * 64) pts[e] = P.hull[e];
    for (int e = threadIdx.x; e < ns_lds * SBOX_REALS; e += NPC_WAVES * 64) sbl[e] = P.sbox[e];
    __syncthreads();
    const int maxc = P.max_contacts < CONVEX_MAXC ? P.max_contacts : CONVEX_MAXC;
    // What a pair (body, static box) needs before its walk -- the body's rotation matrix, its box against the static box's, the
    // walk's filter, which of the box's corners are near the hull -- is a few hundred instructions of plain arithmetic, the same for
    // all 64 lanes.  So the wavefront does it for its four bodies at once, a body a lane (lane l: body l & 3; the corners: lanes
    // 8 j + c, body j, corner c), and each body's walk fetches its values from its lane.
    const int jl = lane & 3, jc = (lane >> 3) & 3;
    const V3<T> xl = { __shfl(pose, 8 * jl + 0, 64), __shfl(pose, 8 * jl + 1, 64), __shfl(pose, 8 * jl + 2, 64) };
    const M3<T> Rl = quat_to_R(Q4<T>{ __shfl(pose, 8 * jl + 3, 64), __shfl(pose, 8 * jl + 4, 64), __shfl(pose, 8 * jl + 5, 64), __shfl(pose, 8 * jl + 6, 64) });
    const T radius_l = __shfl(pose, 8 * jl + 7, 64);      // the hull's bounding radius; its AABB is that sphere's box (body_aabb)
    // (the bounding sphere's box, a rounding's worth wider: it only has to CONTAIN the exact box -- pairs that make contacts are
    //  confirmed against that one, hull_aabb_meets_static)
    const T wide = radius_l * T(1.0001) + T(8) * RealEps<T>::v() * (tabs(xl.x) + tabs(xl.y) + tabs(xl.z));
    const T lo[3] = { xl.x - wide, xl.y - wide, xl.z - wide }, hi[3] = { xl.x + wide, xl.y + wide, xl.z + wide };
    const V3<T> xc = { __shfl(pose, 8 * jc + 0, 64), __shfl(pose, 8 * jc + 1, 64), __shfl(pose, 8 * jc + 2, 64) };
    const T radius_c = __shfl(pose, 8 * jc + 7, 64);
    const bool live_l = (live_mask >> (8 * jl)) & 1ull;
    int nc_l = 0;                                                    // lane j: the contacts of body j so far
    auto fetch_R = [&](int j) {
        M3<T> R;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) R.m[a][c2] = __shfl(Rl.m[a][c2], j, 64);
        return R;
    };
    if (P.plane_on) {
        for (int j = 0; j < NPC_PER_WAVE; j++) {
            if (!((live_mask >> (8 * j)) & 1ull)) continue;          // wave-uniform
            const int64_t i = i0 + j;
            const V3<T> x = { __shfl(pose, 8 * j + 0, 64), __shfl(pose, 8 * j + 1, 64), __shfl(pose, 8 * j + 2, 64) };
            const int k = wave_convex_plane<T>(x, fetch_R(j), __shfl(pose, 8 * j + 7, 64), P, maxc, lane,
                                               [&](int rank, const V3<T> &p, const V3<T> &nn, T dep) { put_sc(P.sbuf, i, rank, p, nn, dep); }, pts);
            if (lane == j) nc_l = k;
        }
    }
    auto boxes = [&](const T *sbase, int count) {                    // (twice below: the boxes in LDS, the boxes in memory)
        for (int s = 0; s < count; s++) {
            const T *sb = sbase + s * SBOX_REALS;
            const unsigned meet = (unsigned)(__ballot(live_l && aabb_meets_static(lo, hi, sb)) & 0xfull);
            if (meet == 0u) continue;
            const V3<T> sx = { sb[SBOX_POS], sb[SBOX_POS + 1], sb[SBOX_POS + 2] };
            M3<T> sR;
            for (int a = 0; a < 3; a++) for (int c2 = 0; c2 < 3; c2++) sR.m[a][c2] = sb[SBOX_R + 3 * a + c2];
            const T half[3] = { T(0.5) * sb[SBOX_SIDE], T(0.5) * sb[SBOX_SIDE + 1], T(0.5) * sb[SBOX_SIDE + 2] };
            const T baabb[6] = { sb[SBOX_LO], sb[SBOX_LO + 1], sb[SBOX_LO + 2], sb[SBOX_HI], sb[SBOX_HI + 1], sb[SBOX_HI + 2] };      // (once a box, not once a body)
            BoxFilter<T> Fl = box_filter<T>(sx, sR, half, xl, Rl, radius_l);
            if (P.hull_nofilter & 1) Fl.bound = Limits<T>::inf();
            const unsigned long long corners = __ballot(box_corner_near<T>(sx, sR, half, xc, radius_c, lane & 7));
            for (int j = 0; j < NPC_PER_WAVE; j++) {
                if (!((meet >> j) & 1u)) continue;                   // wave-uniform
                const int64_t i = i0 + j;
                const V3<T> x = { __shfl(pose, 8 * j + 0, 64), __shfl(pose, 8 * j + 1, 64), __shfl(pose, 8 * j + 2, 64) };
                const BoxFilter<T> F = { { __shfl(Fl.u.x, j, 64), __shfl(Fl.u.y, j, 64), __shfl(Fl.u.z, j, 64) }, __shfl(Fl.off, j, 64), __shfl(Fl.bound, j, 64), __shfl(Fl.slack, j, 64) };
                const int base = __shfl(nc_l, j, 64);
                const M3<T> R = fetch_R(j);
                bool meet_exact = false;
                int k = wave_box_convex_walk<T>(sx, sR, half, x, R, F, (unsigned)((corners >> (8 * j)) & 0xffull), P, maxc, true, lane,
                    [&](int rank, const V3<T> &p, const V3<T> &nn, T dep) { if (base + rank < SC_MAXC) put_sc(P.sbuf, i, base + rank, p, nn, dep); }, pts,
                    baabb, &meet_exact);
                if (k > 0 && !meet_exact && !(P.hull_nofilter & 2) && !hull_aabb_meets_static<T>(x, R, pts, P.hull_n, lane, sb)) k = 0;
                if (lane == j) nc_l += k;
            }
        }
    };
    boxes(sbl, ns_lds);
    boxes(P.sbox + (size_t)ns_lds * SBOX_REALS, P.n_static - ns_lds);
    if (lane < NPC_PER_WAVE && live_l) P.scount[i0 + lane] = nc_l > SC_MAXC ? SC_MAXC + 1 : nc_l;
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
    if (P.hull_n > 0) {
        // the tile form while the hull's points fit the LDS a launch may ask for without more ado (64 KB: some 5 000 points in f32)
        static const bool per_body = getenv("DMX_HULL_WAVE_PER_BODY") != nullptr;
        const size_t lds = ((size_t)3 * P.hull_n + (size_t)(P.n_static < NPC_SBOX ? P.n_static : NPC_SBOX) * SBOX_REALS) * sizeof(T);
        if (!per_body && lds <= 64 * 1024)
            hipLaunchKernelGGL((np_convex_static_tile<T>), dim3((unsigned)((n + NPC_TILE - 1) / NPC_TILE)), dim3(NPC_WAVES * 64), lds, st, S, gtype, n, P);
        else
            hipLaunchKernelGGL((np_convex_static<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, S, gtype, n, P);
    }
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
