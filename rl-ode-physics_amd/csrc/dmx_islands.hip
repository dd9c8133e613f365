// dmx_islands.hip -- general dynamics-island step: dWorldQuickStep over islands of any size, fed by an
// explicit contact list (the contact joints of dJointCreateContact + dJointAttach,
// /root/reference/src/main.c:690-691, or the device narrowphase's output).
//
// The SOR sweep is a sequential Gauss-Seidel over the island's rows in creation order.  Two kernels run the
// same per-body / per-contact / per-row phase functions and therefore give the same bits:
//   solve_islands    -- one lane per island: plenty of small islands in flight at once;
//   solve_island_wg  -- one workgroup per LARGE island (a pile): every phase is spread over the workgroup's lanes;
//                       the sweep follows a level schedule (row r's level = 1 + the latest level of an earlier row
//                       sharing a body with r): rows of one level touch disjoint bodies, so updating them
//                       concurrently gives exactly the sequential result; one barrier per level.
// Rows and per-body scratch live in HBM/L2; solve_island_wg keeps the constraint-force accumulators in LDS during the
// sweeps and prefetches each lane's next row across the level barrier.  Single bodies resting on the ground plane never come here: they take
// the fused register-resident path (step_plane).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "dmx_internal.hpp"
#include "dmx_exact.hpp"
#include "dmx_math.hpp"
#include "dmx_step_fused.hpp"

#include "dmx_island_rows.hpp"

namespace dmx {


// ================================================================================ one lane per single-body island, rows in registers
// A body touching static geometry only (the ground plane, static boxes) is an island of its own: all its rows share the one
// body, so the sweep is a pure chain -- a wavefront per island (solve_island_wg) would run one lane at a time.  Here a LANE
// owns the island: up to SINGLE_MAXC contacts x 3 rows live in its registers (no row traffic at all), 64 islands per wave.
// Same phase arithmetic as stage_body / contact_rows / body_tmp / row_setup / row_sor / finish_body with the second body
// absent, operation for operation: same bits.  (What a box resting on the reference's floor, main.c:115, costs per tick.)
constexpr int SINGLE_MAXC = 4;          // rows in registers (solve_singles)
constexpr int SINGLE_MAXC_LDS = 8;      // rows in LDS (solve_singles_lds): a convex hull's eight contacts with the floor

template <class T> __device__ __forceinline__ bool island_is_single(const IslandSet<T> &I, int isl)
{
    const int nb = I.body_off[isl + 1] - I.body_off[isl], nc = I.con_off[isl + 1] - I.con_off[isl];
    return nb == 1 && nc >= 1 && nc <= SINGLE_MAXC_LDS;
}

template <class T> struct RowS { T J[6], iMJ[6], rhs, ad, lam; };

template <class T>
__global__ __launch_bounds__(64) void solve_singles(T *__restrict__ S, const uint8_t *__restrict__ bflags, int64_t stride,
                                                    IslandSet<T> I, StepParams<T> P, StepDiag *__restrict__ diag)
{
    const int isl = blockIdx.x * blockDim.x + threadIdx.x;
    if (isl >= I.n_islands || !island_is_single(I, isl)) return;
    const T h = P.h, hinv = T(1) / h;
    const int s = I.bodies[I.body_off[isl]];
    const int c0 = I.con_off[isl], nc = I.con_off[isl + 1] - c0;
    if (nc > SINGLE_MAXC) return;                             // five to eight contacts: solve_singles_lds
    T b[BW_COUNT];
    stage_body(S, bflags, stride, I, P, b, s, 0);
    const bool own_surface = I.cmu != nullptr, ind = I.csrc != nullptr;
    const V3<T> x1 = ldS(S, stride, C_POS, s), v1 = ldS(S, stride, C_LVEL, s), w1 = ldS(S, stride, C_AVEL, s);
    constexpr int MAXR = 3 * SINGLE_MAXC;
    RowS<T> row[MAXR];
    bool valid[MAXR];
    T lo_f[SINGLE_MAXC], hi_f[SINGLE_MAXC];           // friction bounds of contact c (normal rows: [0, inf))
    // ---- contact_rows, second body absent
#pragma unroll
    for (int c = 0; c < SINGLE_MAXC; c++) {
#pragma unroll
        for (int d = 0; d < 3; d++) valid[3 * c + d] = false;
        lo_f[c] = hi_f[c] = T(0);
        if (c < nc) {
            const int ci = c0 + c;
            const size_t gi = ind ? (size_t)I.csrc[ci] : (size_t)ci;
            const V3<T> normal = ld3((ind ? I.gnormal : I.cnormal) + 3 * gi);
            const V3<T> cpos = ld3((ind ? I.gpos : I.cpos) + 3 * gi);
            const V3<T> c1 = { cpos.x - x1.x, cpos.y - x1.y, cpos.z - x1.z };
            const int mode = own_surface ? I.cmode[ci] : P.surf_mode;
            T mu = own_surface ? I.cmu[ci] : P.mu;
            if (mu < 0) mu = 0;
            const int rpc = mu > 0 ? 3 : 1;
            V3<T> dir[3];
            dir[0] = normal;
            dir[1] = dir[2] = { T(0), T(0), T(0) };
            if (rpc == 3) plane_space(normal, dir[1], dir[2]);
            lo_f[c] = -mu; hi_f[c] = mu;
#pragma unroll
            for (int dnum = 0; dnum < 3; dnum++) {
                if (dnum < rpc) {
                    RowS<T> &r = row[3 * c + dnum];
                    valid[3 * c + dnum] = true;
                    r.J[0] = dir[dnum].x; r.J[1] = dir[dnum].y; r.J[2] = dir[dnum].z;
                    const V3<T> a = cross(c1, dir[dnum]);
                    r.J[3] = a.x; r.J[4] = a.y; r.J[5] = a.z;
                    T cval = T(0), cfm = P.cfm;
                    if (dnum == 0) {
                        T erp = P.erp;
                        if (mode & SURF_SOFT_ERP) erp = own_surface ? I.csoft_erp[ci] : T(0);
                        if (mode & SURF_SOFT_CFM) cfm = own_surface ? I.csoft_cfm[ci] : T(0);
                        T depth = ind ? I.gdepth[gi] : I.cdepth[ci];
                        if (depth < 0) depth = 0;
                        cval = (hinv * erp) * depth;
                        if (mode & SURF_BOUNCE) {
                            const T outgoing = dot3p(r.J, v1) + dot3p(r.J + 3, w1);
                            const T bv = own_surface ? I.cbounce_vel[ci] : P.bounce_vel;
                            if (bv >= 0 && (-outgoing) > bv) {
                                const T newc = -(own_surface ? I.cbounce[ci] : P.bounce) * outgoing;
                                if (newc > cval) cval = newc;
                            }
                        }
                    }
                    r.rhs = cval; r.ad = cfm; r.lam = T(0);
                }
            }
        }
    }
    body_tmp(S, stride, b, s, hinv);
    // ---- row_setup
#pragma unroll
    for (int i = 0; i < MAXR; i++) {
        if (valid[i]) {
            RowS<T> &r = row[i];
            T sum = T(0);
#pragma unroll
            for (int j = 0; j < 6; j++) sum = fma_(r.J[j], b[BW_TMP + j], sum);
            r.rhs = fma_(r.rhs, hinv, -sum);
            r.ad *= hinv;
#pragma unroll
            for (int j = 0; j < 3; j++) r.iMJ[j] = b[BW_INVM] * r.J[j];
            const V3<T> ja1 = { r.J[3], r.J[4], r.J[5] };
            r.iMJ[3] = dot3p(b + BW_INVI + 0, ja1); r.iMJ[4] = dot3p(b + BW_INVI + 3, ja1); r.iMJ[5] = dot3p(b + BW_INVI + 6, ja1);
            T s2 = T(0);
#pragma unroll
            for (int j = 0; j < 6; j++) s2 = fma_(r.iMJ[j], r.J[j], s2);
            const T cfm = r.ad;
            const T ad = P.sor_w / (s2 + cfm);
#pragma unroll
            for (int j = 0; j < 6; j++) r.J[j] *= ad;
            r.rhs *= ad;
            r.ad = ad * cfm;
        }
    }
    // ---- the sweeps (row_sor), rows in creation order
    double resid = 0.0;
    T *fc = b + BW_FC;
    for (int it = 0; it < P.iters; it++) {
        const bool last = (it == P.iters - 1);
#pragma unroll
        for (int i = 0; i < MAXR; i++) {
            if (valid[i]) {
                RowS<T> &r = row[i];
                const T old = r.lam;
                T delta = fma_(-old, r.ad, r.rhs);
                delta -= fma_(fc[5], r.J[5], fma_(fc[4], r.J[4], fma_(fc[3], r.J[3], fma_(fc[2], r.J[2], fma_(fc[1], r.J[1], fc[0] * r.J[0])))));
                const T lo = (i % 3 == 0) ? T(0) : lo_f[i / 3], hi = (i % 3 == 0) ? Limits<T>::inf() : hi_f[i / 3];
                const T nl = old + delta;
                if (nl < lo) { delta = lo - old; r.lam = lo; }
                else if (nl > hi) { delta = hi - old; r.lam = hi; }
                else r.lam = nl;
#pragma unroll
                for (int j = 0; j < 6; j++) fc[j] = fma_(delta, r.iMJ[j], fc[j]);
                if (last) resid += (double)tabs(delta);
            }
        }
    }
    finish_body(S, bflags, stride, b, s, true, h);
    atomicAdd(&diag->contacts, (unsigned long long)nc);
    atomicAdd(&diag->residual, resid);
}

// The same island shape with five to eight contacts (a convex hull on the floor: up to 24 rows): too many rows for a lane's
// registers, so their J and M^-1 J^T live in LDS, one column per lane (field f of row r of lane l at
// [(r * RS_FIELDS + f) * lanes + l]: a wavefront's access to one field is one conflict-free LDS row); rhs, Ad cfm and lambda
// stay in registers.  A row update is one batch of twelve independent LDS reads, then arithmetic: the same sequence once more.
enum : int { RS_J = 0, RS_IMJ = 6, RS_FIELDS = 12 };      // per row in LDS: J (scaled by Ad) and M^-1 J^T; rhs, Ad cfm, lambda stay in registers

template <class T>
__global__ __launch_bounds__(64) void solve_singles_lds(T *__restrict__ S, const uint8_t *__restrict__ bflags, int64_t stride,
                                                        IslandSet<T> I, StepParams<T> P, StepDiag *__restrict__ diag)
{
    extern __shared__ __align__(16) unsigned char rs_raw[];
    T *rs = reinterpret_cast<T *>(rs_raw);
    constexpr int MAXR = 3 * SINGLE_MAXC_LDS;
    const int lanes = blockDim.x, lane = threadIdx.x;
    const int isl = blockIdx.x * lanes + lane;
    if (isl >= I.n_islands || !island_is_single(I, isl)) return;
    const int c0 = I.con_off[isl], nc = I.con_off[isl + 1] - c0;
    if (nc <= SINGLE_MAXC) return;                            // solve_singles has it
    const T h = P.h, hinv = T(1) / h;
    const int s = I.bodies[I.body_off[isl]];
    T b[BW_COUNT];
    stage_body(S, bflags, stride, I, P, b, s, 0);
    const bool own_surface = I.cmu != nullptr, ind = I.csrc != nullptr;
    const V3<T> x1 = ldS(S, stride, C_POS, s), v1 = ldS(S, stride, C_LVEL, s), w1 = ldS(S, stride, C_AVEL, s);
    auto at = [&](int r, int f) -> T & { return rs[(size_t)(r * RS_FIELDS + f) * lanes + lane]; };
    unsigned valid = 0;                                       // bit r: slot r (= 3 * contact + direction) holds a row
    T lo_f[SINGLE_MAXC_LDS], hi_f[SINGLE_MAXC_LDS];
    T rhsr[MAXR], adr[MAXR], lamr[MAXR];
#pragma unroll
    for (int q = 0; q < MAXR; q++) rhsr[q] = adr[q] = lamr[q] = T(0);
#pragma unroll
    for (int c = 0; c < SINGLE_MAXC_LDS; c++) {
        lo_f[c] = hi_f[c] = T(0);
        if (c >= nc) continue;
        const int ci = c0 + c;
        const size_t gi = ind ? (size_t)I.csrc[ci] : (size_t)ci;
        const V3<T> normal = ld3((ind ? I.gnormal : I.cnormal) + 3 * gi);
        const V3<T> cpos = ld3((ind ? I.gpos : I.cpos) + 3 * gi);
        const V3<T> c1 = { cpos.x - x1.x, cpos.y - x1.y, cpos.z - x1.z };
        const int mode = own_surface ? I.cmode[ci] : P.surf_mode;
        T mu = own_surface ? I.cmu[ci] : P.mu;
        if (mu < 0) mu = 0;
        const int rpc = mu > 0 ? 3 : 1;
        V3<T> dir[3];
        dir[0] = normal;
        dir[1] = dir[2] = { T(0), T(0), T(0) };
        if (rpc == 3) plane_space(normal, dir[1], dir[2]);
        lo_f[c] = -mu; hi_f[c] = mu;
#pragma unroll
        for (int dnum = 0; dnum < 3; dnum++) {
            if (dnum >= rpc) continue;
            const int r = 3 * c + dnum;
            valid |= 1u << r;
            T J[6] = { dir[dnum].x, dir[dnum].y, dir[dnum].z, T(0), T(0), T(0) };
            const V3<T> a = cross(c1, dir[dnum]);
            J[3] = a.x; J[4] = a.y; J[5] = a.z;
            T cval = T(0), cfm = P.cfm;
            if (dnum == 0) {
                T erp = P.erp;
                if (mode & SURF_SOFT_ERP) erp = own_surface ? I.csoft_erp[ci] : T(0);
                if (mode & SURF_SOFT_CFM) cfm = own_surface ? I.csoft_cfm[ci] : T(0);
                T depth = ind ? I.gdepth[gi] : I.cdepth[ci];
                if (depth < 0) depth = 0;
                cval = (hinv * erp) * depth;
                if (mode & SURF_BOUNCE) {
                    const T outgoing = dot3p(J, v1) + dot3p(J + 3, w1);
                    const T bv = own_surface ? I.cbounce_vel[ci] : P.bounce_vel;
                    if (bv >= 0 && (-outgoing) > bv) {
                        const T newc = -(own_surface ? I.cbounce[ci] : P.bounce) * outgoing;
                        if (newc > cval) cval = newc;
                    }
                }
            }
            for (int j = 0; j < 6; j++) at(r, RS_J + j) = J[j];
#pragma unroll
            for (int q = 0; q < MAXR; q++) if (q == r) { rhsr[q] = cval; adr[q] = cfm; }      // (static indices keep the arrays in registers)
        }
    }
    body_tmp(S, stride, b, s, hinv);
#pragma unroll
    for (int i = 0; i < MAXR; i++) {                          // row_setup
        if (!(valid >> i & 1u)) continue;
        T J[6], iMJ[6];
#pragma unroll
        for (int j = 0; j < 6; j++) J[j] = at(i, RS_J + j);
        T sum = T(0);
#pragma unroll
        for (int j = 0; j < 6; j++) sum = fma_(J[j], b[BW_TMP + j], sum);
        T rhs = fma_(rhsr[i], hinv, -sum);
        const T cfm = adr[i] * hinv;
        for (int j = 0; j < 3; j++) iMJ[j] = b[BW_INVM] * J[j];
        const V3<T> ja1 = { J[3], J[4], J[5] };
        iMJ[3] = dot3p(b + BW_INVI + 0, ja1); iMJ[4] = dot3p(b + BW_INVI + 3, ja1); iMJ[5] = dot3p(b + BW_INVI + 6, ja1);
        T s2 = T(0);
        for (int j = 0; j < 6; j++) s2 = fma_(iMJ[j], J[j], s2);
        const T ad = P.sor_w / (s2 + cfm);
#pragma unroll
        for (int j = 0; j < 6; j++) { at(i, RS_J + j) = J[j] * ad; at(i, RS_IMJ + j) = iMJ[j]; }
        rhs *= ad;
        rhsr[i] = rhs;
        adr[i] = ad * cfm;
        lamr[i] = T(0);
    }
    double resid = 0.0;
    T *fc = b + BW_FC;
    // One sweep (row_sor, rows in creation order).  A row's twelve LDS words do not depend on the rows before it: they are
    // fetched, in one batch of independent reads, while the row before is being updated -- the chain of updates never waits
    // for LDS.  (Slots without a row are fetched too and not used.)  LAST: the final sweep also sums |delta lambda|.
    auto sweep = [&](auto LAST) {
        T rw[RS_FIELDS], nx[RS_FIELDS];
#pragma unroll
        for (int f = 0; f < RS_FIELDS; f++) rw[f] = at(0, f);
#pragma unroll
        for (int i = 0; i < MAXR; i++) {
            if (i + 1 < MAXR) {
#pragma unroll
                for (int f = 0; f < RS_FIELDS; f++) nx[f] = at(i + 1, f);
            }
            if (valid >> i & 1u) {
                const T old = lamr[i];
                T delta = fma_(-old, adr[i], rhsr[i]);
                delta -= fma_(fc[5], rw[RS_J + 5], fma_(fc[4], rw[RS_J + 4], fma_(fc[3], rw[RS_J + 3],
                         fma_(fc[2], rw[RS_J + 2], fma_(fc[1], rw[RS_J + 1], fc[0] * rw[RS_J + 0])))));
                const T lo = (i % 3 == 0) ? T(0) : lo_f[i / 3], hi = (i % 3 == 0) ? Limits<T>::inf() : hi_f[i / 3];
                const T nl = old + delta;
                T lam = nl;
                if (nl < lo) { delta = lo - old; lam = lo; }
                else if (nl > hi) { delta = hi - old; lam = hi; }
                lamr[i] = lam;
#pragma unroll
                for (int j = 0; j < 6; j++) fc[j] = fma_(delta, rw[RS_IMJ + j], fc[j]);
                if (decltype(LAST)::value) resid += (double)tabs(delta);
            }
            if (i + 1 < MAXR) {
#pragma unroll
                for (int f = 0; f < RS_FIELDS; f++) rw[f] = nx[f];
            }
            // (one row ahead and no further: left to itself the scheduler hoists every fetch of the unrolled sweep to its top,
            //  288 registers of them)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int it = 0; it + 1 < P.iters; it++) sweep(std::false_type{});
    if (P.iters > 0) sweep(std::true_type{});
    finish_body(S, bflags, stride, b, s, true, h);
    atomicAdd(&diag->contacts, (unsigned long long)nc);
    atomicAdd(&diag->residual, resid);
}

// ================================================================================ one lane per island
template <class T>
__global__ __launch_bounds__(64) void solve_islands(T *__restrict__ S, const uint8_t *__restrict__ bflags,
                                                    int64_t stride, IslandSet<T> I, StepParams<T> P,
                                                    StepDiag *__restrict__ diag)
{
    const int isl = blockIdx.x * blockDim.x + threadIdx.x;
    if (isl >= I.n_islands) return;
    if (I.big != nullptr && I.big[isl] >= 0) return;          // a workgroup owns this one (solve_island_wg)
    if (I.singles && island_is_single(I, isl)) return;        // one body, a few static contacts: solve_singles
    const T h = P.h, hinv = T(1) / h;
    const int b0 = I.body_off[isl], nb = I.body_off[isl + 1] - b0;
    const int c0 = I.con_off[isl], nc = I.con_off[isl + 1] - c0;
    const int r0 = I.row_off[isl];
    T *bs = I.bscr + (size_t)b0 * BW_COUNT;
    T *rows = I.rows + (size_t)r0 * RW_COUNT;
    int *jb = I.rowjb + 2 * (size_t)r0;

    for (int k = 0; k < nb; k++) stage_body(S, bflags, stride, I, P, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], k);
    int m = 0;
    for (int c = 0; c < nc; c++) {
        contact_rows(S, stride, I, P, rows, jb, c0 + c, m, hinv);
        m += contact_rpc(I, P, c0 + c);
    }
    double resid = 0.0;
    if (m > 0) {
        for (int k = 0; k < nb; k++) body_tmp(S, stride, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], hinv);
        for (int i = 0; i < m; i++) row_setup(rows, jb, bs, i, hinv, P.sor_w);
        for (int it = 0; it < P.iters; it++) {
            const bool last = (it == P.iters - 1);
            const int *ord = I.order != nullptr ? I.order + (size_t)(it >> 3) * I.order_stride + r0 : nullptr;
            for (int i = 0; i < m; i++) {
                const T d = row_sor(rows, jb, bs, ord != nullptr ? ord[i] : i);
                if (last) resid += (double)d;
            }
        }
    }
    for (int k = 0; k < nb; k++) finish_body(S, bflags, stride, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], m > 0, h);
    if (nc > 0) {
        atomicAdd(&diag->contacts, (unsigned long long)nc);
        atomicAdd(&diag->residual, resid);
    }
}

// ---- rows -> (lane, slot) of the wavefront that keeps them in registers --------------------------------------------------
// A lane has RPL slots of one row each.  Laid out plainly (row v in slot v / 64 of lane v % 64) the rows of one level sit in
// any slot, so every level step tests all RPL slots of every lane, and rows of one level that sit in different slots are
// updated one slot after the other.  Laid out BY LEVEL CLASS -- a row of level L in slot L mod RPL, lanes filled in row
// order -- a level step concerns one slot only: one test, one pass.  tab[c * 64 +
// t] = the row lane t holds in slot c (-1: none).  Returns false (and the plain layout) when a class has more than 64 rows.
// Called by the wavefront's 64 lanes; the caller synchronises before reading tab.
template <int RPL, class LevelOf>
__device__ __forceinline__ bool assign_row_slots(int nrow, int lane, short *tab, LevelOf level_of)
{
    int cnt[RPL];
#pragma unroll
    for (int c = 0; c < RPL; c++) { cnt[c] = 0; tab[c * 64 + lane] = (short)-1; }
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = 0; base < nrow; base += 64) {
        const int v = base + lane;
        const int cls = v < nrow ? level_of(v) % RPL : -1;
#pragma unroll
        for (int c = 0; c < RPL; c++) {
            const unsigned long long mk = __ballot(cls == c);
            if (cls == c) {
                const int rank = cnt[c] + __popcll(mk & lt);
                if (rank < 64) tab[c * 64 + rank] = (short)v;
            }
            cnt[c] += __popcll(mk);
        }
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < RPL; c++) ok = ok && cnt[c] <= 64;
    if (!ok) {
#pragma unroll
        for (int c = 0; c < RPL; c++) { const int v = lane + 64 * c; tab[c * 64 + lane] = (short)(v < nrow ? v : -1); }
    }
    return ok;
}

// One sweep of the wavefront's rows.  classed: slot lv mod RPL holds every row of level lv (assign_row_slots).
template <class T, int RPL, bool LAST>
__device__ __forceinline__ void wave_sweep(RowRegs<T> (&mine)[RPL], const int (&my_level)[RPL], int nlev, bool classed, T *fc_lds, bool eager,
                                           double &resid)
{
    if (classed) {
        for (int lv0 = 0; lv0 < nlev; lv0 += RPL) {
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                if (lv0 + j < nlev) {
                    if (my_level[j] == lv0 + j) {
                        const T d = row_sor_lds<T, false>(nullptr, mine[j], fc_lds, eager);
                        if (LAST) resid += (double)d;
                    }
                    __syncthreads();
                }
            }
        }
    } else {
        for (int lv = 0; lv < nlev; lv++) {
#pragma unroll
            for (int j = 0; j < RPL; j++)
                if (my_level[j] == lv) {
                    const T d = row_sor_lds<T, false>(nullptr, mine[j], fc_lds, eager);
                    if (LAST) resid += (double)d;
                }
            __syncthreads();
        }
    }
}

// The sweeps of a small island by one wavefront (the first, if the launch has four): every row is OWNED by a lane and stays
// in that lane's registers, RPL rows per lane; a level step is "lanes whose row is in this level update it"; between two
// barriers only the bodies' accumulators in LDS are touched.  Returns the lane's share of the last sweep's residual.
template <class T, int RPL>
__device__ __forceinline__ double wave_island_sweeps(T *rows, const int *jb, const int *row_level, int m, int nlev, int iters, int tid,
                                                     T *fc_lds, bool eager)
{
    __shared__ short tab[RPL * 64];
    __shared__ int tab_classed;
    if (tid < 64) {
        const bool ok = RPL > 1 ? assign_row_slots<RPL>(m, tid, tab, [&](int v) { return row_level[v]; }) : false;
        if (RPL == 1) tab[tid] = (short)(tid < m ? tid : -1);
        if (tid == 0) tab_classed = ok ? 1 : 0;
    }
    __syncthreads();
    const bool classed = tab_classed != 0;
    RowRegs<T> mine[RPL];
    int my_level[RPL];
#pragma unroll
    for (int j = 0; j < RPL; j++) {
        const int r = tid < 64 ? (int)tab[j * 64 + tid] : -1;
        my_level[j] = -1;
        if (r >= 0) { row_load(rows, jb, r, mine[j]); my_level[j] = row_level[r]; }
    }
    __syncthreads();
    double resid = 0.0;
    for (int it = 0; it + 1 < iters; it++) wave_sweep<T, RPL, false>(mine, my_level, nlev, classed, fc_lds, eager, resid);
    if (iters > 0) wave_sweep<T, RPL, true>(mine, my_level, nlev, classed, fc_lds, eager, resid);      // the last sweep also sums its |delta lambda|
#pragma unroll
    for (int j = 0; j < RPL; j++)
        if (my_level[j] >= 0) rows[(size_t)mine[j].row * RW_COUNT + RW_LAM] = mine[j].lam;
    return resid;
}

// ---- the same sweeps with a CONTACT (its normal and two friction rows) as the unit a lane owns -----------------------------
// The three rows of a contact are consecutive in creation order and share both bodies, so they sit on three consecutive levels
// and nothing else touches those bodies in between: a lane that owns all three fetches the two bodies' accumulators once,
// updates them in registers through the three rows, and writes them back once -- one LDS round trip per contact instead of
// per row, with exactly the row-by-row arithmetic.  Contact level = (level of its first row) / 3: two contacts sharing a body
// are at least three row levels apart, so they never get the same contact level and keep their order.  For islands whose
// contacts all carry the batch's surface with friction (three rows each).
template <class T> struct ContactRegs { T J[3][12], iMJ[3][12], rhs[3], ad[3], lo[3], hi[3], lam[3]; int l1, l2, row0; };

template <class T>
__device__ __forceinline__ void contact_load(const T *rows, const int *jb, int r0, ContactRegs<T> &c)
{
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const T *row = rows + (size_t)(r0 + d) * RW_COUNT;
#pragma unroll
        for (int j = 0; j < 12; j++) { c.J[d][j] = row[RW_J + j]; c.iMJ[d][j] = row[RW_IMJ + j]; }
        c.rhs[d] = row[RW_RHS]; c.ad[d] = row[RW_AD]; c.lo[d] = row[RW_LO]; c.hi[d] = row[RW_HI]; c.lam[d] = row[RW_LAM];
    }
    c.l1 = jb[2 * r0]; c.l2 = jb[2 * r0 + 1];
    c.row0 = r0;
}

// the same three rows made in registers, never stored: contact_rows + row_setup on thread-local arrays (the functions the
// other paths run on the row arrays in HBM, so the same bits), then straight into the lane's ContactRegs
template <class T>
__device__ __forceinline__ void contact_build(const T *S, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P, const T *bs, int ci,
                                              T hinv, ContactRegs<T> &c)
{
    T lrows[3 * RW_COUNT];
    int ljb[6];
    contact_rows<T, 3>(S, stride, I, P, lrows, ljb, ci, 0, hinv);
#pragma unroll
    for (int d = 0; d < 3; d++) row_setup(lrows, ljb, bs, d, hinv, P.sor_w);
    contact_load(lrows, ljb, 0, c);
}

// FAST: the friction rows are unbounded (mu = inf, the reference's surface, main.c:687): their clamp -- two comparisons that are
// false and the selects behind them -- is left out; the normal row keeps its own.  Same values either way.
template <class T, bool LAST, bool FAST = false>
__device__ __forceinline__ void contact_sor_lds(ContactRegs<T> &c, T *fc, bool eager, double &resid)
{
    T *fc1 = fc + 6 * c.l1;
    T *fc2 = fc + 6 * (c.l2 >= 0 ? c.l2 : c.l1);
    const bool two = c.l2 >= 0;
    T a[6], b[6];
#pragma unroll
    for (int j = 0; j < 6; j++) a[j] = fc1[j];
    if (two || eager) {
#pragma unroll
        for (int j = 0; j < 6; j++) b[j] = fc2[j];
    } else {
#pragma unroll
        for (int j = 0; j < 6; j++) b[j] = T(0);
    }
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const T *J = c.J[d];
        const T old = c.lam[d];
        T delta = fma_(-old, c.ad[d], c.rhs[d]);
        delta -= fma_(a[5], J[5], fma_(a[4], J[4], fma_(a[3], J[3], fma_(a[2], J[2], fma_(a[1], J[1], a[0] * J[0])))));
        if (two)
            delta -= fma_(b[5], J[11], fma_(b[4], J[10], fma_(b[3], J[9], fma_(b[2], J[8], fma_(b[1], J[7], b[0] * J[6])))));
        const T nl = old + delta;
        if (FAST && d > 0) c.lam[d] = nl;
        else if (nl < c.lo[d]) { delta = c.lo[d] - old; c.lam[d] = c.lo[d]; }
        else if (nl > c.hi[d]) { delta = c.hi[d] - old; c.lam[d] = c.hi[d]; }
        else c.lam[d] = nl;
#pragma unroll
        for (int j = 0; j < 6; j++) a[j] = fma_(delta, c.iMJ[d][j], a[j]);
        if (two) {
#pragma unroll
            for (int j = 0; j < 6; j++) b[j] = fma_(delta, c.iMJ[d][6 + j], b[j]);
        }
        if (LAST) resid += (double)tabs(delta);
    }
#pragma unroll
    for (int j = 0; j < 6; j++) fc1[j] = a[j];
    if (two) {
#pragma unroll
        for (int j = 0; j < 6; j++) fc2[j] = b[j];
    }
}

// one sweep over the wavefront's contacts; slot cl mod CPL holds every contact of contact level cl when classed
template <class T, int CPL, bool LAST>
__device__ __forceinline__ void wave_contact_sweep(ContactRegs<T> (&mine)[CPL], const int (&my_cl)[CPL], int n_clev, bool classed, T *fc_lds,
                                                   bool eager, double &resid)
{
    if (classed) {
        for (int cl0 = 0; cl0 < n_clev; cl0 += CPL) {
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                if (cl0 + j < n_clev) {
                    if (my_cl[j] == cl0 + j) contact_sor_lds<T, LAST>(mine[j], fc_lds, eager, resid);
                    __syncthreads();
                }
            }
        }
    } else {
        for (int cl = 0; cl < n_clev; cl++) {
#pragma unroll
            for (int j = 0; j < CPL; j++)
                if (my_cl[j] == cl) contact_sor_lds<T, LAST>(mine[j], fc_lds, eager, resid);
            __syncthreads();
        }
    }
}

// one island's sweeps with contacts as units (crow: the island's contacts' first rows, island-relative).  load(v, c): contact
// v's rows into c -- from the row arrays, or made on the spot (contact_build: then nothing is written back either).
template <class T, int CPL, class Load>
__device__ __forceinline__ double wave_island_contact_sweeps(T *rows, const int *row_level, const int *crow, int nc, int nlev,
                                                             int iters, int tid, T *fc_lds, bool eager, bool write_back, Load load)
{
    __shared__ short tab[CPL * 64];
    __shared__ int tab_classed;
    if (tid < 64) {
        const bool ok = CPL > 1 ? assign_row_slots<CPL>(nc, tid, tab, [&](int v) { return row_level[crow[v]] / 3; }) : false;
        if (CPL == 1) tab[tid] = (short)(tid < nc ? tid : -1);
        if (tid == 0) tab_classed = ok ? 1 : 0;
    }
    __syncthreads();
    const bool classed = tab_classed != 0;
    ContactRegs<T> mine[CPL];
    int my_cl[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        const int v = tid < 64 ? (int)tab[j * 64 + tid] : -1;
        my_cl[j] = -1;
        if (v >= 0) { load(v, mine[j]); mine[j].row0 = crow[v]; my_cl[j] = row_level[crow[v]] / 3; }
    }
    __syncthreads();
    const int n_clev = (nlev + 2) / 3;
    double resid = 0.0;
    for (int it = 0; it + 1 < iters; it++) wave_contact_sweep<T, CPL, false>(mine, my_cl, n_clev, classed, fc_lds, eager, resid);
    if (iters > 0) wave_contact_sweep<T, CPL, true>(mine, my_cl, n_clev, classed, fc_lds, eager, resid);
    if (write_back) {
#pragma unroll
        for (int j = 0; j < CPL; j++)
            if (my_cl[j] >= 0) {
#pragma unroll
                for (int d = 0; d < 3; d++) rows[(size_t)(mine[j].row0 + d) * RW_COUNT + RW_LAM] = mine[j].lam[d];
            }
    }
    return resid;
}

// rows a thread of the register-resident form holds at most: a workgroup of 256 owns up to 3 072 rows in f32, 1 536 in f64 (the
// reference's pen holds at most 512 bodies, body.h:6 -- their pile is 2 000-2 600 rows)
template <class T> constexpr int REGS_ROWS = sizeof(T) == 4 ? 3072 : 1536;      // rows a workgroup of the register form holds at most
template <class T, int WG> constexpr int REGS_ROWS_PER_THREAD = REGS_ROWS<T> / WG;

// The sweeps of a large island by a whole workgroup with every row in REGISTERS: the row at position t of the island's level
// lists (rows grouped by level, lev_rows) belongs to thread t mod WG, RPL rows per thread, for all twenty sweeps; a level step is
// "threads holding a row of this level update it" and an LDS hand-over of the accumulators.  Nothing is fetched from device
// memory between the first sweep and the last.  (Streaming a lane's row of each level from L2 instead -- solve_island_wg's
// general form -- makes every level step one L2 round trip long: 0.9 us in the reference's pen, 45 levels x 20 sweeps.)  A level's
// rows are consecutive positions, so they sit in at most two of a thread's slots (one, if the level is no wider than what is
// left of the slot): a wavefront runs one or two row updates per level, not RPL.  Returns the thread's share of the last sweep's residual.
template <class T, int RPL, int WG>
__device__ __forceinline__ double wg_island_sweeps(T *rows, const int *jb, const int *row_level, const int *lev_rows, int m, int nlev, int iters,
                                                   int tid, T *fc_lds)
{
    RowRegs<T> mine[RPL];
    int my_level[RPL];
#pragma unroll
    for (int j = 0; j < RPL; j++) {
        const int t = tid + WG * j;
        my_level[j] = -1;
        if (t < m) { const int r = lev_rows[t]; row_load(rows, jb, r, mine[j]); my_level[j] = row_level[r]; }
    }
    double resid = 0.0;
    for (int it = 0; it < iters; it++) {
        const bool last = it + 1 == iters;
        for (int lv = 0; lv < nlev; lv++) {
#pragma unroll
            for (int j = 0; j < RPL; j++)
                if (my_level[j] == lv) {
                    const T d = row_sor_lds<T, false>(nullptr, mine[j], fc_lds, true);
                    if (last) resid += (double)d;
                }
            lds_barrier();
        }
    }
#pragma unroll
    for (int j = 0; j < RPL; j++)
        if (my_level[j] >= 0) rows[(size_t)mine[j].row * RW_COUNT + RW_LAM] = mine[j].lam;
    return resid;
}

// The same with CONTACTS as units (three rows per contact throughout): the u-th contact in level order belongs to thread u mod WG,
// CPL contacts a thread.  A contact's normal and two friction rows sit on consecutive levels on the same two bodies, so a lane that
// holds all three fetches the bodies' accumulators once, carries them through the three row updates in registers and writes them
// back once: one LDS round trip and one barrier per CONTACT level where the row form pays three (the pen's pile: 150 row levels x
// 20 sweeps, each ~260 cycles of which ~130 are the round trip and the barrier).  The arithmetic per row is row_sor_lds's, in the
// same order: same bits.  cfirst: nc ints of LDS scratch.
template <class T, int CPL, int WG>
__device__ __forceinline__ double wg_island_contact_sweeps(T *rows, const int *jb, const int *row_level, const int *lev_rows, const int *lev_off,
                                                           int m, int nlev, int iters, int tid, T *fc_lds, int *cfirst)
{
    const int nc = m / 3, n_clev = nlev / 3;
    // contacts in level order: the rows of level 3 cl are the first rows of contact level cl's contacts, and the levels below hold
    // exactly three rows of every earlier contact
    for (int p = tid; p < m; p += WG) {
        const int r = lev_rows[p], lv = row_level[r];
        if (lv % 3 == 0) { const int a = lev_off[lv] - lev_off[0]; cfirst[a / 3 + (p - a)] = r; }
    }
    __syncthreads();
    ContactRegs<T> mine[CPL];
    int my_cl[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        const int u = tid + WG * j;
        my_cl[j] = -1;
        if (u < nc) { const int r0 = cfirst[u]; contact_load(rows, jb, r0, mine[j]); my_cl[j] = row_level[r0] / 3; }
    }
    double resid = 0.0;
    // (do this island's friction rows ever clamp?  Asked of the rows themselves: the bounds sit in the registers just loaded)
    int unbounded = 1;
#pragma unroll
    for (int j = 0; j < CPL; j++)
        if (my_cl[j] >= 0)
            unbounded &= (mine[j].lo[1] == -Limits<T>::inf() && mine[j].hi[1] == Limits<T>::inf() && mine[j].lo[2] == -Limits<T>::inf() &&
                          mine[j].hi[2] == Limits<T>::inf()) ? 1 : 0;
    const bool fast = __syncthreads_and(unbounded) != 0;
    auto sweeps = [&](auto FASTT) {
        constexpr bool F = decltype(FASTT)::value;
        for (int it = 0; it + 1 < iters; it++)
            for (int cl = 0; cl < n_clev; cl++) {
#pragma unroll
                for (int j = 0; j < CPL; j++)
                    if (my_cl[j] == cl) contact_sor_lds<T, false, F>(mine[j], fc_lds, true, resid);
                lds_barrier();
            }
        if (iters > 0)
            for (int cl = 0; cl < n_clev; cl++) {
#pragma unroll
                for (int j = 0; j < CPL; j++)
                    if (my_cl[j] == cl) contact_sor_lds<T, true, F>(mine[j], fc_lds, true, resid);
                lds_barrier();
            }
    };
    if (fast) sweeps(std::true_type{}); else sweeps(std::false_type{});
#pragma unroll
    for (int j = 0; j < CPL; j++)
        if (my_cl[j] >= 0) {
#pragma unroll
            for (int d = 0; d < 3; d++) rows[(size_t)(mine[j].row0 + d) * RW_COUNT + RW_LAM] = mine[j].lam[d];
        }
    return resid;
}

// ================================================================================ one workgroup per large island
constexpr int FC_LDS_BYTES = 48 * 1024;     // islands of up to 2048 (f32) / 1024 (f64) bodies keep their accumulators in LDS
// (WAVE_ISLAND_ROWS, dmx_internal.hpp: islands of up to that many rows are solved by one wavefront with the rows in registers;
//  such an island has at most 2 x 256 bodies, which always fit the LDS above)

// REGS: islands of up to WG x 12 (f32) / WG x 6 (f64) rows keep them in registers for the sweeps (wg_island_sweeps); a separate
// instantiation, so that the streaming forms keep their register budget
template <class T, int WG, bool REGS>
__device__ __forceinline__ void solve_island_wg_body(T *__restrict__ S, const uint8_t *__restrict__ bflags,
                                                     int64_t stride, const IslandSet<T> &I, const StepParams<T> &P,
                                                     StepDiag *__restrict__ diag, int lds_bodies, const ExactCounts *__restrict__ dc,
                                                     int sched_ints)
{
    // dc: a launch enqueued before the host has seen the tick's counts (careful_tick, small scenes) -- the grid covers the
    // capacity, the record on the device says how many islands there are and whether this launch may act at all
    if (dc != nullptr && (dc->spec_ok == 0u || blockIdx.x >= dc->nbig)) return;
    const int isl = I.big_list[blockIdx.x];
    const int tid = threadIdx.x;
    const T h = P.h, hinv = T(1) / h;
    const int b0 = I.body_off[isl], nb = I.body_off[isl + 1] - b0;
    const int c0 = I.con_off[isl], nc = I.con_off[isl + 1] - c0;
    const int r0 = I.row_off[isl];
    T *bs = I.bscr + (size_t)b0 * BW_COUNT;
    T *rows = I.rows + (size_t)r0 * RW_COUNT;
    int *jb = I.rowjb + 2 * (size_t)r0;
    const int lv0 = I.big[isl];                               // this island's slice of the level schedule
    const int nlev = I.lev_count[blockIdx.x];
    const int *lev_off = I.lev_off + lv0;                     // [nlev+1], offsets into lev_rows (island-relative rows)
    const int m = lev_off[nlev] - lev_off[0];

    // Small island, three rows per contact throughout (the batch's surface with friction, or per-contact surfaces that all
    // have it): one wavefront, a lane owns a contact and makes its rows in its own registers -- they never go to HBM.
    const bool wave = nb <= lds_bodies && nlev > 0 && m <= WAVE_ISLAND_ROWS;         // (workgroup-uniform, like all of this)
    bool by_contact = false;
    if (wave) {
        by_contact = I.cmu == nullptr && P.mu > 0;
        if (I.cmu != nullptr) {
            int all3 = 1;
            for (int c = tid; c < nc; c += WG) all3 &= I.cmu[c0 + c] > 0 ? 1 : 0;
            by_contact = __syncthreads_and(all3) != 0;
        }
        by_contact = by_contact && (nc <= 64 || sizeof(T) == 4);      // (a contact's rows are 90 reals of registers: two per lane in f32)
    }

    for (int k = tid; k < nb; k += WG) stage_body(S, bflags, stride, I, P, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], k);
    __syncthreads();
    if (!by_contact)
        for (int c = tid; c < nc; c += WG) contact_rows(S, stride, I, P, rows, jb, c0 + c, I.crow[c0 + c], hinv);
    for (int k = tid; k < nb; k += WG) body_tmp(S, stride, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], hinv);
    __syncthreads();
    if (!by_contact) {
        for (int i = tid; i < m; i += WG) row_setup(rows, jb, bs, i, hinv, P.sor_w);
        __syncthreads();
    }

    double resid = 0.0;
    // LDS staging: the only data one level hands to the next is the bodies' constraint-force accumulators (6 reals per
    // body); they live in LDS for the sweeps.  A row's own data does not depend on other rows, so each lane fetches its
    // row of the NEXT level before it works on this one: between two barriers only LDS traffic and arithmetic remain.
    // dynamic LDS, sized by the launch for the largest island in it (small islands must not reserve 48 KB each: that
    // would cap a CU at three of them)
    extern __shared__ __align__(16) unsigned char fc_raw[];
    T *fc_lds = reinterpret_cast<T *>(fc_raw);
    const bool use_lds = nb <= lds_bodies && nlev > 0;    // workgroup-uniform
    if (use_lds && m <= WAVE_ISLAND_ROWS) {
        // Small island, one wavefront (the first, if the launch has four): every row is OWNED by a lane for the whole solve
        // (row r by lane r mod 64) and stays in that lane's registers; a level step is "lanes whose row is in this level
        // update it".  Nothing is fetched between two barriers but the bodies' accumulators in LDS.  Only row_level is read
        // of the schedule: the device-side builder (dmx_exact.hip) leaves the per-level row lists of such islands unbuilt.
        for (int k = tid; k < nb; k += WG)
            for (int j = 0; j < 6; j++) fc_lds[6 * k + j] = bs[(size_t)k * BW_COUNT + BW_FC + j];
        const int *row_level = I.row_level + lev_off[0];
        // (rows per lane as a template parameter: an island of up to 64 rows pays for one row's tests per level, not four)
        const bool eager = (dc != nullptr ? dc->nbig : gridDim.x) < 2048u;           // few islands: every one waits on its own chain of rows
        auto build = [&](int v, ContactRegs<T> &c) { contact_build(S, stride, I, P, bs, c0 + v, hinv, c); };
        if (by_contact && nc <= 64) resid = wave_island_contact_sweeps<T, 1>(rows, row_level, I.crow + c0, nc, nlev, P.iters, tid, fc_lds, eager, false, build);
        else if (by_contact) resid = wave_island_contact_sweeps<T, sizeof(T) == 4 ? 2 : 1>(rows, row_level, I.crow + c0, nc, nlev, P.iters, tid, fc_lds, eager, false, build);
        else if (m <= 64) resid = wave_island_sweeps<T, 1>(rows, jb, row_level, m, nlev, P.iters, tid, fc_lds, eager);
        else if (m <= 128) resid = wave_island_sweeps<T, 2>(rows, jb, row_level, m, nlev, P.iters, tid, fc_lds, eager);
        else resid = wave_island_sweeps<T, WAVE_ISLAND_ROWS / 64>(rows, jb, row_level, m, nlev, P.iters, tid, fc_lds, eager);
        for (int k = tid; k < nb; k += WG)
            for (int j = 0; j < 6; j++) bs[(size_t)k * BW_COUNT + BW_FC + j] = fc_lds[6 * k + j];
    } else if (REGS && use_lds && m <= REGS_ROWS<T>) {
        const int contact_scratch_ints = sched_ints;          // (REGS launches: room behind the accumulators for one int per contact)
        for (int k = tid; k < nb; k += WG)
            for (int j = 0; j < 6; j++) fc_lds[6 * k + j] = bs[(size_t)k * BW_COUNT + BW_FC + j];
        __syncthreads();
        const int *row_level = I.row_level + lev_off[0], *lev_rows = I.lev_rows + lev_off[0];
        // (rows per thread as a template parameter: 32 registers a row in f32, 64 in f64 -- of a lane's 512 at one wave per SIMD)
        constexpr int RMAX = REGS_ROWS_PER_THREAD<T, WG>;          // 6 (f32) / 3 (f64) at 512 threads, 12 / 6 at 256
        // three rows per contact throughout (the batch's surface with friction, or per-contact surfaces that all have it) and at
        // most two contacts a thread (a contact is 90 registers in f32): contacts as units
        bool all3 = sizeof(T) == 4 && m == 3 * nc && nlev % 3 == 0 && nc <= 2 * WG && contact_scratch_ints >= nc && (I.cmu != nullptr || P.mu > 0);
        if (all3 && I.cmu != nullptr) {
            int a3 = 1;
            for (int c = tid; c < nc; c += WG) a3 &= I.cmu[c0 + c] > 0 ? 1 : 0;
            all3 = __syncthreads_and(a3) != 0;
        }
        int *cfirst = reinterpret_cast<int *>(fc_lds + (size_t)6 * lds_bodies);
        if (all3 && nc <= WG) resid = wg_island_contact_sweeps<T, 1, WG>(rows, jb, row_level, lev_rows, lev_off, m, nlev, P.iters, tid, fc_lds, cfirst);
        else if (all3) resid = wg_island_contact_sweeps<T, sizeof(T) == 4 ? 2 : 1, WG>(rows, jb, row_level, lev_rows, lev_off, m, nlev, P.iters, tid, fc_lds, cfirst);
        else if (m <= 2 * WG) resid = wg_island_sweeps<T, 2, WG>(rows, jb, row_level, lev_rows, m, nlev, P.iters, tid, fc_lds);
        else if (m <= 3 * WG || RMAX <= 3) resid = wg_island_sweeps<T, RMAX < 3 ? RMAX : 3, WG>(rows, jb, row_level, lev_rows, m, nlev, P.iters, tid, fc_lds);
        else if (m <= 4 * WG || RMAX <= 4) resid = wg_island_sweeps<T, RMAX < 4 ? RMAX : 4, WG>(rows, jb, row_level, lev_rows, m, nlev, P.iters, tid, fc_lds);
        else if (m <= 6 * WG || RMAX <= 6) resid = wg_island_sweeps<T, RMAX < 6 ? RMAX : 6, WG>(rows, jb, row_level, lev_rows, m, nlev, P.iters, tid, fc_lds);
        else if (m <= 8 * WG || RMAX <= 8) resid = wg_island_sweeps<T, RMAX < 8 ? RMAX : 8, WG>(rows, jb, row_level, lev_rows, m, nlev, P.iters, tid, fc_lds);
        else resid = wg_island_sweeps<T, RMAX, WG>(rows, jb, row_level, lev_rows, m, nlev, P.iters, tid, fc_lds);
        __syncthreads();
        for (int k = tid; k < nb; k += WG)
            for (int j = 0; j < 6; j++) bs[(size_t)k * BW_COUNT + BW_FC + j] = fc_lds[6 * k + j];
    } else if (use_lds) {
        for (int k = tid; k < nb; k += WG)
            for (int j = 0; j < 6; j++) fc_lds[6 * k + j] = bs[(size_t)k * BW_COUNT + BW_FC + j];
        // The schedule itself in LDS when the launch made room for it (sched_ints): a level step's row is found by two dependent
        // look-ups (the level's offset, then the row list) before the row can be fetched -- three L2 round trips in a chain, the
        // whole length of a level step, when they go to device memory; here they are LDS reads and the fetch is issued two
        // steps ahead of its use.
        int *loff = reinterpret_cast<int *>(fc_lds + (size_t)6 * lds_bodies), *lrows = loff + nlev + 1;
        const bool sched = sched_ints >= nlev + 1 + m;
        if (sched) {
            for (int q = tid; q <= nlev; q += WG) loff[q] = lev_off[q] - lev_off[0];
            for (int t = tid; t < m; t += WG) lrows[t] = I.lev_rows[lev_off[0] + t];
        }
        __syncthreads();
        const int total = nlev * P.iters;
        auto first_row = [&](int lv) {
            if (sched) { const int t = loff[lv] + tid; return t < loff[lv + 1] ? lrows[t] : -1; }
            const int t = lev_off[lv] + tid;
            return t < lev_off[lv + 1] ? I.lev_rows[t] : -1;
        };
        auto next_level = [&](int lv) { return lv + 1 == nlev ? 0 : lv + 1; };
        // rows of steps g, g + 1, g + 2 (a lane's row of each level: position tid of the level's list)
        RowRegs<T> cur, n1, n2;
        int lv0 = 0, lv1 = next_level(0), lv2 = next_level(lv1);
        int rc = total > 0 ? first_row(lv0) : -1, r1 = total > 1 ? first_row(lv1) : -1, r2 = -1;
        if (rc >= 0) row_load(rows, jb, rc, cur);
        if (r1 >= 0) row_load(rows, jb, r1, n1);
        for (int g = 0; g < total; g++) {
            const bool last = g >= total - nlev;
            r2 = g + 2 < total ? first_row(lv2) : -1;
            if (r2 >= 0) row_load(rows, jb, r2, n2);
            if (rc >= 0) {
                const T d = row_sor_lds(rows, cur, fc_lds);
                if (last) resid += (double)d;
                // schedules of one or two levels: a fetch issued before this update holds the old multiplier
                if (r1 == rc) n1.lam = cur.lam;
                if (r2 == rc) n2.lam = cur.lam;
            }
            {                                                                       // levels wider than the workgroup
                const int a = sched ? loff[lv0] + lev_off[0] : lev_off[lv0], e = sched ? loff[lv0 + 1] + lev_off[0] : lev_off[lv0 + 1];
                for (int t = a + tid + WG; t < e; t += WG) {
                    RowRegs<T> x;
                    row_load(rows, jb, I.lev_rows[t], x);
                    const T d = row_sor_lds(rows, x, fc_lds);
                    if (last) resid += (double)d;
                }
            }
            // the next level reads the accumulators this one wrote: an LDS hand-over.  NOT __syncthreads(): that also waits for
            // every outstanding global access (vmcnt(0)) -- the rows fetched for the levels ahead, which are in flight precisely
            // so that nobody waits for them; with it every level step was one L2 round trip long (800 ns in the pen's pile).
            lds_barrier();
            cur = n1; rc = r1; n1 = n2; r1 = r2;
            lv0 = lv1; lv1 = lv2; lv2 = next_level(lv2);
        }
        for (int k = tid; k < nb; k += WG)                     // back for finish_body (same lane, same bodies)
            for (int j = 0; j < 6; j++) bs[(size_t)k * BW_COUNT + BW_FC + j] = fc_lds[6 * k + j];
    } else {
        for (int it = 0; it < P.iters; it++) {
            const bool last = (it == P.iters - 1);
            for (int lv = 0; lv < nlev; lv++) {
                const int a = lev_off[lv], e = lev_off[lv + 1];
                for (int t = a + tid; t < e; t += WG) {
                    const T d = row_sor(rows, jb, bs, I.lev_rows[t]);
                    if (last) resid += (double)d;
                }
                __syncthreads();                              // the next level reads the fc this one wrote
            }
        }
    }
    for (int k = tid; k < nb; k += WG) finish_body(S, bflags, stride, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], m > 0, h);

    // residual: wave reduction, then one atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) resid += __shfl_xor(resid, o, 64);
    if ((tid & 63) == 0) atomicAdd(&diag->residual, resid);
    if (tid == 0) atomicAdd(&diag->contacts, (unsigned long long)nc);
}
template <class T, int WG, bool REGS = false>
__global__ __launch_bounds__(WG) void solve_island_wg(T *__restrict__ S, const uint8_t *__restrict__ bflags,
                                                      int64_t stride, IslandSet<T> I, StepParams<T> P,
                                                      StepDiag *__restrict__ diag, int lds_bodies, const ExactCounts *__restrict__ dc,
                                                      int sched_ints)
{
    solve_island_wg_body<T, WG, REGS>(S, bflags, stride, I, P, diag, lds_bodies, dc, sched_ints);
}
// The tail of a small-scene exact tick in ONE launch: workgroups [0, max_big) are solve_island_wg<64>'s (speculative form: they ask
// the device's record whether their island exists), the workgroups behind them step 64 bodies each of everyone else with the fused
// ground-plane kernel's own code (step_plane_body, dmx_step_fused.hpp; Pf carries the involved bodies' skip mask and the same
// gate).  The two touch disjoint bodies; one after the other they cost 49 + 15 us on a 1 024-body scene, side by side 49.
template <class T>
__global__ __launch_bounds__(64) void solve_islands_and_step(T *__restrict__ S, const uint8_t *__restrict__ bflags,
                                                             const uint8_t *__restrict__ gtype, int64_t stride, int64_t n, IslandSet<T> I,
                                                             StepParams<T> P, StepParams<T> Pf, StepDiag *__restrict__ diag_isl,
                                                             StepDiag *__restrict__ diag_fused, int lds_bodies,
                                                             const ExactCounts *__restrict__ dc, unsigned max_big)
{
    if (blockIdx.x < max_big) solve_island_wg_body<T, 64, false>(S, bflags, stride, I, P, diag_isl, lds_bodies, dc, 0);
    else step_plane_body<T, false, 4>(S, S, gtype, stride, n, Pf, diag_fused, (int64_t)(blockIdx.x - max_big) * 64 + threadIdx.x);
}

// ================================================================================ dWorldStep: the island's LCP solved exactly
// One workgroup per island with rows.  Same rows as the SOR kernels (stage_body / contact_rows / body_tmp / row_setup),
// then  A = J M^-1 J^T + diag(cfm / h)  and block principal pivoting on  A lambda = b + w, lo <= lambda <= hi  (free /
// at-lo / at-hi sets, Cholesky of the free block, flip the violators; Murty's single flip once the violation count has
// stalled three times); every loop has a fixed operation order, whatever the thread mapping.  A lives in HBM/L2 (m^2 reals
// per island, scratch sized by the host); the free block's factor is staged in LDS when it fits.
enum : int { LCP_FREE = 0, LCP_LO = 1, LCP_HI = 2 };

template <class T> __device__ __forceinline__ T dot6acc(const T *a, const T *b, T acc)
{
#pragma unroll
    for (int k = 0; k < 6; k++) acc = fma_(a[k], b[k], acc);
    return acc;
}

template <class T, int WG>
__global__ __launch_bounds__(WG) void lcp_island_wg(T *__restrict__ S, const uint8_t *__restrict__ bflags, int64_t stride,
                                                    IslandSet<T> I, StepParams<T> P, StepDiag *__restrict__ diag,
                                                    T *__restrict__ scratch, const long long *__restrict__ scratch_off,
                                                    int *__restrict__ iscratch, int lds_rows)
{
    const int isl = I.big_list[blockIdx.x];
    const int tid = threadIdx.x;
    const T h = P.h, hinv = T(1) / h;
    const int b0 = I.body_off[isl], nb = I.body_off[isl + 1] - b0;
    const int c0 = I.con_off[isl], nc = I.con_off[isl + 1] - c0;
    const int r0 = I.row_off[isl];
    T *bs = I.bscr + (size_t)b0 * BW_COUNT;
    T *rows = I.rows + (size_t)r0 * RW_COUNT;
    int *jb = I.rowjb + 2 * (size_t)r0;
    // rows of this island: contacts are laid out by crow (first row of each contact), rpc rows each
    const int m = nc > 0 ? I.crow[c0 + nc - 1] + contact_rpc(I, P, c0 + nc - 1) : 0;

    for (int k = tid; k < nb; k += WG) stage_body(S, bflags, stride, I, P, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], k);
    __syncthreads();
    for (int c = tid; c < nc; c += WG) contact_rows(S, stride, I, P, rows, jb, c0 + c, I.crow[c0 + c], hinv);
    for (int k = tid; k < nb; k += WG) body_tmp(S, stride, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], hinv);
    __syncthreads();
    for (int i = tid; i < m; i += WG) row_setup<T, false>(rows, jb, bs, i, hinv, P.sor_w);
    __syncthreads();

    T *A = scratch + scratch_off[blockIdx.x];
    T *Mg = A + (size_t)m * m, *rg = Mg + (size_t)m * m, *lam = rg + m, *wv = lam + m;
    int *state = iscratch + 3 * (size_t)r0, *idx = state + m, *viol = idx + m;
    extern __shared__ __align__(16) unsigned char lcp_raw[];
    __shared__ int s_nf, s_done;
    __shared__ T s_tol;

    // A = J iMJ^T over shared bodies + diag(cfm / h)
    for (long long e = tid; e < (long long)m * m; e += WG) {
        const int i = (int)(e / m), j = (int)(e - (long long)i * m);
        const T *ji = rows + (size_t)i * RW_COUNT + RW_J, *pj = rows + (size_t)j * RW_COUNT + RW_IMJ;
        const int i1 = jb[2 * i], i2 = jb[2 * i + 1], j1 = jb[2 * j], j2 = jb[2 * j + 1];
        T a = T(0);
        if (i1 == j1) a = dot6acc(ji, pj, a);
        if (j2 >= 0 && i1 == j2) a = dot6acc(ji, pj + 6, a);
        if (i2 >= 0 && i2 == j1) a = dot6acc(ji + 6, pj, a);
        if (i2 >= 0 && j2 >= 0 && i2 == j2) a = dot6acc(ji + 6, pj + 6, a);
        if (i == j) a += rows[(size_t)i * RW_COUNT + RW_AD];
        A[e] = a;
    }
    if (tid == 0) {
        T bmax = T(0);
        for (int i = 0; i < m; i++) { const T v = tabs(rows[(size_t)i * RW_COUNT + RW_RHS]); if (v > bmax) bmax = v; }
        s_tol = (sizeof(T) == 4 ? T(1e-5) : T(1e-11)) * (T(1) + bmax);
    }
    for (int i = tid; i < m; i += WG) { state[i] = LCP_FREE; lam[i] = T(0); }
    __syncthreads();
    const T tol = s_tol;
    int best = m + 1, patience = 3;                     // (only thread 0's copies matter)
    const int max_rounds = 20 * m + 100;
    for (int round = 0;; round++) {
        if (tid == 0) {
            int nf = 0;
            for (int i = 0; i < m; i++) {
                if (state[i] == LCP_FREE) idx[nf++] = i;
                else lam[i] = state[i] == LCP_LO ? rows[(size_t)i * RW_COUNT + RW_LO] : rows[(size_t)i * RW_COUNT + RW_HI];
            }
            s_nf = nf;
        }
        __syncthreads();
        const int nf = s_nf;
        const bool in_lds = nf <= lds_rows;             // the free block's factor and right-hand side fit in LDS
        T *M = in_lds ? reinterpret_cast<T *>(lcp_raw) : Mg;
        T *r = in_lds ? M + (size_t)nf * nf : rg;
        for (int a = tid; a < nf; a += WG) {
            const int i = idx[a];
            T s = rows[(size_t)i * RW_COUNT + RW_RHS];
            for (int j = 0; j < m; j++) if (state[j] != LCP_FREE && lam[j] != T(0)) s -= A[(size_t)i * m + j] * lam[j];
            r[a] = s;
        }
        for (long long e = tid; e < (long long)nf * nf; e += WG) {
            const int a = (int)(e / nf), c = (int)(e - (long long)a * nf);
            if (c <= a) M[e] = A[(size_t)idx[a] * m + idx[c]];
        }
        __syncthreads();
        // Cholesky, right-looking, lower triangle in place
        for (int k = 0; k < nf; k++) {
            if (tid == 0) { T d = M[(size_t)k * nf + k]; d = tsqrt<T>(d > T(0) ? d : tol); M[(size_t)k * nf + k] = d; }
            __syncthreads();
            const T d = M[(size_t)k * nf + k];
            for (int i = k + 1 + tid; i < nf; i += WG) M[(size_t)i * nf + k] /= d;
            __syncthreads();
            const int cnt = nf - k - 1;
            const long long total = (long long)cnt * (cnt + 1) / 2;
            for (long long e = tid; e < total; e += WG) {
                int ii = (int)((tsqrt<double>(8.0 * (double)e + 1.0) - 1.0) * 0.5);
                while ((long long)ii * (ii + 1) / 2 > e) ii--;
                while ((long long)(ii + 1) * (ii + 2) / 2 <= e) ii++;
                const int jj = (int)(e - (long long)ii * (ii + 1) / 2);
                const int i = k + 1 + ii, j = k + 1 + jj;
                M[(size_t)i * nf + j] -= M[(size_t)i * nf + k] * M[(size_t)j * nf + k];
            }
            __syncthreads();
        }
        // L y = r (column oriented), L^T x = y
        for (int k = 0; k < nf; k++) {
            if (tid == 0) r[k] /= M[(size_t)k * nf + k];
            __syncthreads();
            const T rk = r[k];
            for (int i = k + 1 + tid; i < nf; i += WG) r[i] -= M[(size_t)i * nf + k] * rk;
            __syncthreads();
        }
        for (int k = nf - 1; k >= 0; k--) {
            if (tid == 0) r[k] /= M[(size_t)k * nf + k];
            __syncthreads();
            const T rk = r[k];
            for (int i = tid; i < k; i += WG) r[i] -= M[(size_t)k * nf + i] * rk;
            __syncthreads();
        }
        for (int a = tid; a < nf; a += WG) lam[idx[a]] = r[a];
        __syncthreads();
        for (int i = tid; i < m; i += WG) {
            T s = -rows[(size_t)i * RW_COUNT + RW_RHS];
            for (int j = 0; j < m; j++) s += A[(size_t)i * m + j] * lam[j];
            wv[i] = s;
        }
        __syncthreads();
        if (tid == 0) {
            int nv = 0, top = -1;
            for (int i = 0; i < m; i++) {
                const T lo = rows[(size_t)i * RW_COUNT + RW_LO], hi = rows[(size_t)i * RW_COUNT + RW_HI];
                int v = 0;
                if (state[i] == LCP_FREE) v = (lam[i] < lo - tol) ? 1 : (lam[i] > hi + tol) ? 2 : 0;
                else if (state[i] == LCP_LO) v = wv[i] < -tol ? 3 : 0;
                else v = wv[i] > tol ? 3 : 0;
                viol[i] = v;
                if (v) { nv++; top = i; }
            }
            int done = (nv == 0 || round >= max_rounds) ? 1 : 0;
            if (!done) {
                bool all = true;
                if (nv < best) { best = nv; patience = 3; }
                else if (patience > 0) patience--;
                else all = false;
                for (int i = 0; i < m; i++) {
                    if (!viol[i] || (!all && i != top)) continue;
                    state[i] = viol[i] == 1 ? LCP_LO : viol[i] == 2 ? LCP_HI : LCP_FREE;
                }
            }
            s_done = done;
        }
        __syncthreads();
        if (s_done) break;
    }
    // clamp what the tolerance let through; cforce = M^-1 J^T lambda, rows in order per body
    for (int i = tid; i < m; i += WG) {
        T l = lam[i];
        if (state[i] == LCP_FREE) {
            const T lo = rows[(size_t)i * RW_COUNT + RW_LO], hi = rows[(size_t)i * RW_COUNT + RW_HI];
            if (l < lo) l = lo;
            if (l > hi) l = hi;
        }
        lam[i] = l;
        rows[(size_t)i * RW_COUNT + RW_LAM] = l;
    }
    __syncthreads();
    double resid = 0.0;
    for (int k = tid; k < nb; k += WG) {
        T f[6] = { T(0), T(0), T(0), T(0), T(0), T(0) };
        for (int i = 0; i < m; i++) {
            const T *ip = rows + (size_t)i * RW_COUNT + RW_IMJ;
            if (jb[2 * i] == k) { for (int q = 0; q < 6; q++) f[q] = fma_(lam[i], ip[q], f[q]); }
            if (jb[2 * i + 1] == k) { for (int q = 0; q < 6; q++) f[q] = fma_(lam[i], ip[6 + q], f[q]); }
        }
        for (int q = 0; q < 6; q++) bs[(size_t)k * BW_COUNT + BW_FC + q] = f[q];
    }
    for (int i = tid; i < m; i += WG) {
        const T w_i = wv[i];
        resid += (double)(state[i] == LCP_FREE ? tabs(w_i) : (state[i] == LCP_LO ? (w_i < T(0) ? -w_i : T(0)) : (w_i > T(0) ? w_i : T(0))));
    }
    for (int k = tid; k < nb; k += WG) finish_body(S, bflags, stride, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], m > 0, h);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) resid += __shfl_xor(resid, o, 64);
    if ((tid & 63) == 0) atomicAdd(&diag->residual, resid);
    if (tid == 0) atomicAdd(&diag->contacts, (unsigned long long)nc);
}

template <class T>
hipError_t launch_islands_exact(T *S, const uint8_t *bflags, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P,
                                StepDiag *diag, T *scratch, const long long *scratch_off, int *iscratch, int max_rows, hipStream_t st)
{
    // (max_rows < 0: the caller solves the islands with rows itself -- lcp_island_lds, dmx_lcp.hip -- and wants the row-less ones only)
    if (I.n_islands <= 0) return hipSuccess;
    if (I.n_big < I.n_islands) {       // islands without rows: free bodies
        const unsigned grid = (unsigned)((I.n_islands + 63) / 64);
        hipLaunchKernelGGL((solve_islands<T>), dim3(grid), dim3(64), 0, st, S, bflags, stride, I, P, diag);
    }
    if (I.n_big > 0 && max_rows >= 0) {
        // LDS for the free block's factor + right-hand side, up to 60 KB
        int lds_rows = 0;
        while ((size_t)(lds_rows + 1) * (lds_rows + 2) * sizeof(T) <= (size_t)60 * 1024 && lds_rows < max_rows) lds_rows++;
        const size_t lds = (size_t)lds_rows * (lds_rows + 1) * sizeof(T);
        hipLaunchKernelGGL((lcp_island_wg<T, 256>), dim3((unsigned)I.n_big), dim3(256), lds, st, S, bflags, stride, I, P, diag,
                           scratch, scratch_off, iscratch, lds_rows);
    }
    return hipGetLastError();
}

template <class T>
hipError_t launch_islands(T *S, const uint8_t *bflags, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P,
                          StepDiag *diag, hipStream_t st)
{
    if (I.n_islands <= 0) return hipSuccess;
    if (I.n_big < I.n_islands) {
        const unsigned grid = (unsigned)((I.n_islands + 63) / 64);
        hipLaunchKernelGGL((solve_islands<T>), dim3(grid), dim3(64), 0, st, S, bflags, stride, I, P, diag);
        if (I.singles) {
            hipLaunchKernelGGL((solve_singles<T>), dim3(grid), dim3(64), 0, st, S, bflags, stride, I, P, diag);
            // islands of 5..8 contacts keep their rows' J and M^-1 J^T in LDS: 64 lanes x 24 rows x 12 fields (f32: 72 KB, two waves
            // per CU; f64: 144 KB of the CU's 160 KB)
            const int lanes = 64;
            const size_t lds = (size_t)lanes * 3 * SINGLE_MAXC_LDS * RS_FIELDS * sizeof(T);
            // (set per launch: the attribute belongs to the function ON THE CURRENT DEVICE, and the call is a table write -- a
            //  process-wide "done once" flag would leave a second device, or a second thread racing the first, without it)
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_singles_lds<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) return ea;
            hipLaunchKernelGGL((solve_singles_lds<T>), dim3((unsigned)((I.n_islands + lanes - 1) / lanes)), dim3(lanes), lds, st, S, bflags,
                               stride, I, P, diag);
        }
    }
    if (I.n_big > 0) {
        // islands of up to lds_bodies bodies keep their accumulators in LDS (larger ones: HBM/L2)
        const int lds_cap = FC_LDS_BYTES / (int)(6 * sizeof(T));
        const int lds_bodies = std::min<int>(std::max(I.big_max_bodies, 1), lds_cap);
        size_t lds = (size_t)lds_bodies * 6 * sizeof(T);
        // a launch whose largest island has more rows than one wavefront holds but few enough for a workgroup's registers (and not
        // thousands of islands: the form runs one workgroup per compute unit) keeps every island's rows in registers for the sweeps
        // (512 threads, two waves per SIMD: six rows a thread all in the 256 architectural registers -- at 256 threads the same rows
        //  are twelve a thread, half of them in accumulator registers that have to be copied out before every use, and twice the
        //  slots to skip per level; DMX_REGS_WG=256 for that form)
        static const int regs_wg = [] { const char *e = getenv("DMX_REGS_WG"); return e && atoi(e) == 256 ? 256 : 512; }();
        if (I.big_max_rows > WAVE_ISLAND_ROWS && I.big_max_rows <= REGS_ROWS<T> && I.n_big <= 1024) {
            // (f64 rows are 64 registers: three a thread at 512 threads spill; 256 threads, six a thread, do not.  Up to 1 024 rows
            //  256 threads hold them in four slots a thread without the accumulator half, and a level step has four waves to
            //  bring to the barrier, not eight: the pen 96 bodies 264 vs 270 us, 400 bodies 645 vs 583, 512 bodies 961 vs 793)
            // (+ one int per contact behind the accumulators: the contact form's list of contacts in level order)
            static const bool by_contact = [] { const char *e = getenv("DMX_REGS_BY_CONTACT"); return !(e && atoi(e) == 0); }();
            const int scratch = by_contact && sizeof(T) == 4 ? REGS_ROWS<T> / 3 : 0;
            lds += (size_t)scratch * sizeof(int);
            if (regs_wg == 512 && sizeof(T) == 4 && I.big_max_rows > 1024)
                hipLaunchKernelGGL((solve_island_wg<T, 512, true>), dim3((unsigned)I.n_big), dim3(512), lds, st, S, bflags, stride, I, P, diag, lds_bodies,
                                   (const ExactCounts *)nullptr, scratch);
            else
                hipLaunchKernelGGL((solve_island_wg<T, 256, true>), dim3((unsigned)I.n_big), dim3(256), lds, st, S, bflags, stride, I, P, diag, lds_bodies,
                                   (const ExactCounts *)nullptr, scratch);
        } else {
            // room for an island's level schedule behind the accumulators (offsets + row lists; big_rows_total bounds any one island's):
            // taken when it is modest -- a few large islands (a pile in the pen) -- not when thousands of small ones share the launch
            const size_t sched = I.big_rows_total > 0 ? (size_t)2 * I.big_rows_total + 2 : 0;
            int sched_ints = 0;
            if (sched > 0 && lds + sched * sizeof(int) <= (size_t)96 * 1024 && I.n_big <= 64) { sched_ints = (int)sched; lds += sched * sizeof(int); }
            if (lds > (size_t)64 * 1024) {
                const void *fn = I.big_max_width <= 64 ? (const void *)&solve_island_wg<T, 64, false> : (const void *)&solve_island_wg<T, 256, false>;
                const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (ea != hipSuccess) return ea;
            }
            if (I.big_max_width <= 64)      // no level has more than 64 rows: one wavefront per island, barriers cost nothing
                hipLaunchKernelGGL((solve_island_wg<T, 64>), dim3((unsigned)I.n_big), dim3(64), lds, st, S, bflags, stride, I, P, diag, lds_bodies, (const ExactCounts *)nullptr, sched_ints);
            else
                hipLaunchKernelGGL((solve_island_wg<T, 256>), dim3((unsigned)I.n_big), dim3(256), lds, st, S, bflags, stride, I, P, diag, lds_bodies, (const ExactCounts *)nullptr, sched_ints);
        }
    }
    return hipGetLastError();
}

// The island solve of a small-scene exact tick, enqueued BEHIND the bookkeeping kernels and before the host has seen their
// counts: one-wavefront workgroups over `max_big` islands (the capacity), every one asking the device's count record whether it
// exists and whether the launch may act (ExactCounts::spec_ok: all islands are solve_island_wg<64>'s kind, nothing overflowed).
// I's counts are not read.
template <class T>
hipError_t launch_islands_speculative(T *S, const uint8_t *bflags, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P,
                                      StepDiag *diag, const ExactCounts *counts_dev, unsigned max_big, hipStream_t st)
{
    if (max_big == 0) return hipSuccess;
    const int lds_bodies = (int)EX_SPEC_ISLAND_BODIES;        // (<= FC_LDS_BYTES' worth: every island of a tick that passes takes the LDS path, as in launch_islands)
    const size_t lds = (size_t)lds_bodies * 6 * sizeof(T);
    hipLaunchKernelGGL((solve_island_wg<T, 64>), dim3(max_big), dim3(64), lds, st, S, bflags, stride, I, P, diag, lds_bodies, counts_dev, 0);
    return hipGetLastError();
}
// ... and the same with the fused ground-plane step for everyone else in the launch (solve_islands_and_step): scenes of boxes and
// spheres on the plane, no static boxes, no hulls, no pending external forces
template <class T>
hipError_t launch_islands_and_step_speculative(T *S, const uint8_t *bflags, const uint8_t *gtype, int64_t stride, int64_t n, const IslandSet<T> &I,
                                               const StepParams<T> &P, const StepParams<T> &Pf, StepDiag *diag_isl, StepDiag *diag_fused,
                                               const ExactCounts *counts_dev, unsigned max_big, hipStream_t st)
{
    const int lds_bodies = (int)EX_SPEC_ISLAND_BODIES;
    const size_t lds = (size_t)lds_bodies * 6 * sizeof(T);
    const unsigned blocks = max_big + (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL((solve_islands_and_step<T>), dim3(blocks), dim3(64), lds, st, S, bflags, gtype, stride, n, I, P, Pf, diag_isl, diag_fused,
                       lds_bodies, counts_dev, max_big);
    return hipGetLastError();
}
template hipError_t launch_islands_and_step_speculative<float>(float *, const uint8_t *, const uint8_t *, int64_t, int64_t, const IslandSet<float> &,
                                                               const StepParams<float> &, const StepParams<float> &, StepDiag *, StepDiag *,
                                                               const ExactCounts *, unsigned, hipStream_t);
template hipError_t launch_islands_and_step_speculative<double>(double *, const uint8_t *, const uint8_t *, int64_t, int64_t, const IslandSet<double> &,
                                                                const StepParams<double> &, const StepParams<double> &, StepDiag *, StepDiag *,
                                                                const ExactCounts *, unsigned, hipStream_t);

template hipError_t launch_islands_speculative<float>(float *, const uint8_t *, int64_t, const IslandSet<float> &, const StepParams<float> &,
                                                      StepDiag *, const ExactCounts *, unsigned, hipStream_t);
template hipError_t launch_islands_speculative<double>(double *, const uint8_t *, int64_t, const IslandSet<double> &, const StepParams<double> &,
                                                       StepDiag *, const ExactCounts *, unsigned, hipStream_t);

template hipError_t launch_islands_exact<float>(float *, const uint8_t *, int64_t, const IslandSet<float> &, const StepParams<float> &,
                                                StepDiag *, float *, const long long *, int *, int, hipStream_t);
template hipError_t launch_islands_exact<double>(double *, const uint8_t *, int64_t, const IslandSet<double> &, const StepParams<double> &,
                                                 StepDiag *, double *, const long long *, int *, int, hipStream_t);
template hipError_t launch_islands<float>(float *, const uint8_t *, int64_t, const IslandSet<float> &,
                                          const StepParams<float> &, StepDiag *, hipStream_t);
template hipError_t launch_islands<double>(double *, const uint8_t *, int64_t, const IslandSet<double> &,
                                           const StepParams<double> &, StepDiag *, hipStream_t);

// HIP loads a translation unit's code object at the first launch of one of its kernels -- a couple of milliseconds each, which an
// interactive caller would meet as a hitch at the first tick that needs the exact pipeline.  dmxBatchCreate asks for one
// kernel's attributes per unit instead (dmx_preload_code, dmx_batch.cpp): the load happens there.
hipError_t dmx_touch_islands(int real_bytes)
{
    // (the unit's code object, and -- what costs more -- each kernel's own first-use set-up: every kernel an exact tick or a fused
    //  tick may launch, in the batch's precision)
    hipFuncAttributes a;
    hipError_t e = hipSuccess;
    auto touch = [&](const void *k) { const hipError_t r = hipFuncGetAttributes(&a, k); if (r != hipSuccess) e = r; };
    if (real_bytes == 4) {
        touch((const void *)&solve_islands<float>);
        touch((const void *)&solve_singles<float>);
        touch((const void *)&solve_singles_lds<float>);
        touch((const void *)&solve_island_wg<float, 64, false>);
        touch((const void *)&solve_island_wg<float, 256, false>);
        touch((const void *)&solve_island_wg<float, 256, true>);
        touch((const void *)&solve_island_wg<float, 512, true>);
        touch((const void *)&solve_islands_and_step<float>);
    } else {
        touch((const void *)&solve_islands<double>);
        touch((const void *)&solve_singles<double>);
        touch((const void *)&solve_singles_lds<double>);
        touch((const void *)&solve_island_wg<double, 64, false>);
        touch((const void *)&solve_island_wg<double, 256, false>);
        touch((const void *)&solve_island_wg<double, 256, true>);
        touch((const void *)&solve_island_wg<double, 512, true>);
        touch((const void *)&solve_islands_and_step<double>);
    }
    return e;
}

}  // namespace dmx
