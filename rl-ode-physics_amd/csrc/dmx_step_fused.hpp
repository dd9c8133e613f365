// dmx_step_fused.hpp -- the fused tick of a single-body island on the ground plane as a DEVICE function (step_plane_body), with
// the helpers the fused kernels share, so that the step can ride in another kernel's launch: dmx_kernels.hip wraps it as
// step_plane; dmx_islands.hip puts it behind the island solve of a small-scene exact tick (solve_islands_and_step: the two
// touch disjoint bodies, one launch runs both kinds of workgroup side by side).
#pragma once
#include <hip/hip_runtime.h>
#include "dmx_internal.hpp"
#include "dmx_math.hpp"
#include "dmx_collide.hpp"

namespace dmx {

// Broadphase safe-zone test of one body (pre-step position): 2 = outside its zone (a body pair may exist),
// 1 = has used more than a quarter of the radius (zones should be refreshed soon), 0 = well inside.
template <class T> __device__ __forceinline__ int zone_state(T dx, T dz, T safe)
{
    const T d2 = dx * dx + dz * dz, s2 = safe * safe;
    if (!(d2 < s2)) return 2;
    return (d2 < s2 * T(0.0625)) ? 0 : 1;
}
// ... and the same question for the static boxes: does the body's bounding sphere (centre x, radius r) reach into the
// AABB of any static box?  2 if so (a contact with static geometry may exist: the exact path decides), else 0.
template <class T> __device__ __forceinline__ int static_state(const StepParams<T> &P, T x, T y, T z, T r)
{
    int st = 0;
    for (int s = 0; s < P.n_static; s++) {
        const T *b = P.sbox + s * SBOX_REALS;
        if (!(x - r > b[SBOX_HI + 0] || x + r < b[SBOX_LO + 0] || y - r > b[SBOX_HI + 1] || y + r < b[SBOX_LO + 1] ||
              z - r > b[SBOX_HI + 2] || z + r < b[SBOX_LO + 2])) st = 2;
    }
    return st;
}
// one flag write per wave at most, and none once the flag is already up
__device__ __forceinline__ void report_zone(int state, uint32_t *flags)
{
    const unsigned long long v = __ballot(state == 2), w = __ballot(state == 1);
    if ((v | w) == 0ull) return;
    const unsigned long long act = __ballot(true);
    if ((int)(threadIdx.x & 63) != __builtin_ctzll(act)) return;       // first active lane reports
    if (v != 0ull && flags[BPF_VIOLATION] == 0u) atomicOr(&flags[BPF_VIOLATION], 1u);
    if (w != 0ull && flags[BPF_WARN] == 0u) atomicOr(&flags[BPF_WARN], 1u);
}

// multi-GPU boundary rows: the new state also goes, AoS, to the exchange's send buffer
template <class T>
__device__ __forceinline__ void pack_boundary(const StepParams<T> &P, int64_t i, const V3<T> &x, const Q4<T> &q,
                                              const V3<T> &v, const V3<T> &w)
{
    if (P.pack_out == nullptr) return;
    int64_t slot;
    if (i < P.pack_lo) slot = i;
    else if (i >= P.pack_hi && i < P.pack_hi + P.pack_lo) slot = P.pack_lo + (i - P.pack_hi);     // (slots behind the upper row: spare slots of the rank, not boundary bodies)
    else return;
    T *o = P.pack_out + slot * C_MASS;
    o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z;
    o[7] = v.x; o[8] = v.y; o[9] = v.z; o[10] = w.x; o[11] = w.y; o[12] = w.z;
}

// One body: external force/torque -> new velocities (no constraints) -> new pose.
//   facc = fext + m g ; tacc = text + gyro
//   v += (h/m) facc ; w += Iw^-1 (h tacc)
//   x += h v ; q += h/2 (0,w) q ; q /= |q|
template <class T>
__device__ __forceinline__ void free_body_step(V3<T> &x, Q4<T> &q, V3<T> &v, V3<T> &w,
                                               T mass, const V3<T> &Ib, V3<T> facc, V3<T> tacc,
                                               const V3<T> &g, T h, int gyro)
{
    const M3<T> R = quat_to_R(q);
    const T invMass = T(1) / mass;
    const V3<T> invIb = { T(1) / Ib.x, T(1) / Ib.y, T(1) / Ib.z };
    facc.x = fma_(mass, g.x, facc.x); facc.y = fma_(mass, g.y, facc.y); facc.z = fma_(mass, g.z, facc.z);
    const M3<T> invIw = rotate_diag(R, invIb);
    if (gyro != 0 && !isotropic(Ib)) {
        const M3<T> Iw = rotate_diag(R, Ib);
        add_gyro_torque(tacc, Iw, w, h, gyro);
    }
    const T hm = h * invMass;
    v.x = fma_(hm, facc.x, v.x); v.y = fma_(hm, facc.y, v.y); v.z = fma_(hm, facc.z, v.z);
    tacc.x *= h; tacc.y *= h; tacc.z *= h;
    const V3<T> dw = mulv(invIw, tacc);
    w.x += dw.x; w.y += dw.y; w.z += dw.z;
    x.x = fma_(h, v.x, x.x); x.y = fma_(h, v.y, x.y); x.z = fma_(h, v.z, x.z);
    integrate_quat(q, w, h);
}

template <class T> __device__ __forceinline__ T wave_sum(T x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// ---------------------------------------------------------------------------------------------
// step_plane's body for body i (a whole wavefront calls it: the diagnostics are a wave reduction, one slot per 64 bodies).
// NC = contact slots per body: 4 (box-plane yields at most 4 contacts) or CONVEX_MAXC when the batch has convex bodies,
// whose plane contacts np_convex_plane left in P.cbuf.
// ---------------------------------------------------------------------------------------------
template <class T, bool EXT, int NC>
__device__ __forceinline__ void step_plane_body(T *S, T *So, const uint8_t *__restrict__ gtype, int64_t stride, int64_t n,
                                                const StepParams<T> &P, StepDiag *__restrict__ diag, const int64_t i)
{
    int my_contacts = 0;
    double my_resid = 0.0;
    // a chunk in which a body has left its zone is rolled back whole: once the flag is up, its remaining ticks need no work
    if (P.bp_check && P.bp_flags[BPF_VIOLATION] != 0u) return;
    if (P.gate != nullptr && *P.gate == 0u) return;
    if (i < n && !(P.skip != nullptr && P.skip[i])) {
        V3<T> x = { S[slab_ix(C_POS + 0, i)], S[slab_ix(C_POS + 1, i)], S[slab_ix(C_POS + 2, i)] };
        if (P.bp_check) {
            int zs = zone_state(x.x - S[slab_ix(C_BPX, i)], x.z - S[slab_ix(C_BPZ, i)], S[slab_ix(C_BPSAFE, i)]);
            if (P.n_static > 0) {
                const int z2 = static_state(P, x.x, x.y, x.z, S[slab_ix(C_BPR, i)]);
                zs = z2 > zs ? z2 : zs;
            }
            report_zone(zs, P.bp_flags);
        }
        Q4<T> q = { S[slab_ix(C_QUAT + 0, i)], S[slab_ix(C_QUAT + 1, i)],
                    S[slab_ix(C_QUAT + 2, i)], S[slab_ix(C_QUAT + 3, i)] };
        V3<T> v = { S[slab_ix(C_LVEL + 0, i)], S[slab_ix(C_LVEL + 1, i)], S[slab_ix(C_LVEL + 2, i)] };
        V3<T> w = { S[slab_ix(C_AVEL + 0, i)], S[slab_ix(C_AVEL + 1, i)], S[slab_ix(C_AVEL + 2, i)] };
        const T mass = S[slab_ix(C_MASS, i)];
        const V3<T> Ib = { S[slab_ix(C_INERTIA + 0, i)], S[slab_ix(C_INERTIA + 1, i)],
                           S[slab_ix(C_INERTIA + 2, i)] };
        const T side[3] = { S[slab_ix(C_SIDES + 0, i)], S[slab_ix(C_SIDES + 1, i)],
                            S[slab_ix(C_SIDES + 2, i)] };
        const int gt = gtype[i];
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

        // ---- narrowphase (dCollide) --------------------------------------------------------
        constexpr int MAXC = NC, MAXR = 3 * NC;
        V3<T> cp[MAXC];
        T cd[MAXC];
        int nc = 0;
        if (P.plane_on) {
            if (gt == GEOM_BOX) nc = box_plane(x, R, side, P.pn, P.pd, P.max_contacts, cp, cd);
            else if (gt == GEOM_SPHERE) nc = sphere_plane(x, side[0], P.pn, P.pd, cp, cd);
            else if (NC >= CONVEX_MAXC && gt == GEOM_CONVEX) {
                nc = P.ccount[i];
                const T *cb = P.cbuf + (size_t)i * CONVEX_MAXC * 4;
#pragma unroll
                for (int k = 0; k < MAXC; k++)
                    if (k < nc) { cp[k] = { cb[4 * k], cb[4 * k + 1], cb[4 * k + 2] }; cd[k] = cb[4 * k + 3]; }
            }
        }
        my_contacts = nc;

        if (nc > 0) {
            // ---- rows: contact k contributes [n | c x n], [t1 | c x t1], [t2 | c x t2] ----
            const int rpc = P.mu > 0 ? 3 : 1;
            const T hinv = T(1) / h;
            V3<T> dir[3];
            dir[0] = P.pn;
            plane_space(P.pn, dir[1], dir[2]);
            // v/h + M^-1 f
            const V3<T> tl = { fma_(facc.x, invMass, v.x * hinv), fma_(facc.y, invMass, v.y * hinv),
                               fma_(facc.z, invMass, v.z * hinv) };
            V3<T> ta = mulv(invIw, tacc);
            ta.x = fma_(w.x, hinv, ta.x); ta.y = fma_(w.y, hinv, ta.y); ta.z = fma_(w.z, hinv, ta.z);
            const V3<T> iml[3] = { { invMass * dir[0].x, invMass * dir[0].y, invMass * dir[0].z },
                                   { invMass * dir[1].x, invMass * dir[1].y, invMass * dir[1].z },
                                   { invMass * dir[2].x, invMass * dir[2].y, invMass * dir[2].z } };
            const T cfm = P.cfm * hinv;

            T Ad[MAXR], rhs[MAXR], adcfm[MAXR], lam[MAXR];
            // row limits are implied by the row kind: normal rows [0, inf), friction rows [-mu, mu]
            const T lo_f = -P.mu, hi_f = P.mu, hi_n = Limits<T>::inf();
            V3<T> Ja[MAXR], iMa[MAXR], Jl[MAXR];      // Jl = the row's linear Jacobian (its direction) times Ad, as J *= Ad leaves it
#pragma unroll
            for (int r = 0; r < MAXR; r++) {      // rows of absent contacts stay zero
                Ad[r] = rhs[r] = adcfm[r] = lam[r] = T(0);
                Ja[r] = { T(0), T(0), T(0) }; iMa[r] = { T(0), T(0), T(0) }; Jl[r] = { T(0), T(0), T(0) };
            }
#pragma unroll
            for (int k = 0; k < MAXC; k++) {
                if (k < nc) {
                    const V3<T> c1 = { cp[k].x - x.x, cp[k].y - x.y, cp[k].z - x.z };
#pragma unroll
                    for (int dnum = 0; dnum < 3; dnum++) {
                        const int r = 3 * k + dnum;
                        if (dnum < rpc) {
                            const V3<T> ja = cross(c1, dir[dnum]);
                            T c = T(0);
                            if (dnum == 0) {
                                T depth = cd[k];
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
                            const V3<T> ima = mulv(invIw, ja);
                            T s2 = iml[dnum].x * dir[dnum].x;
                            s2 = fma_(iml[dnum].y, dir[dnum].y, s2); s2 = fma_(iml[dnum].z, dir[dnum].z, s2);
                            s2 = fma_(ima.x, ja.x, s2); s2 = fma_(ima.y, ja.y, s2); s2 = fma_(ima.z, ja.z, s2);
                            const T ad = P.sor_w / (s2 + cfm);
                            Ad[r] = ad;
                            Ja[r] = { ja.x * ad, ja.y * ad, ja.z * ad };
                            Jl[r] = { dir[dnum].x * ad, dir[dnum].y * ad, dir[dnum].z * ad };
                            iMa[r] = ima;
                            rhs[r] = b * ad;
                            adcfm[r] = ad * cfm;
                            lam[r] = T(0);
                        }
                    }
                }
            }

            // ---- SOR-PGS: lambda = 0 start, rows in creation order --------------------------
            // Branch-free row update: the contact loop bound is the wave's maximum contact count (a scalar
            // branch), lanes with fewer contacts carry zeroed rows -- whose delta is exactly zero; clamping is by select.
            V3<T> fl = { T(0), T(0), T(0) }, fa = { T(0), T(0), T(0) };
            int ncu = 0;     // largest contact count among the wave's active lanes (wave-uniform by construction)
#pragma unroll
            for (int k = 0; k < MAXC; k++)
                if (__ballot(nc > k) != 0ull) ncu = k + 1;
            T rsum = T(0);
            // One sweep over the wave's rows.  FAST: the friction rows are unbounded (mu = inf, the reference's surface): no
            // friction clamp.  LAST: only the final sweep tallies |delta lambda|.  Same arithmetic in every variant.
            // FULL: every contact slot of every active lane is taken and friction rows exist (a box resting on the plane: four
            // contacts x three rows) -- the sweep is one straight line of 3 MAXC row updates, no scalar branch per contact / row
            auto sweep = [&](auto FAST, auto LAST, auto FULL) {
#pragma unroll
                for (int k = 0; k < MAXC; k++) {
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
                                fl.x = fma_(delta, iml[dnum].x, fl.x); fl.y = fma_(delta, iml[dnum].y, fl.y);
                                fl.z = fma_(delta, iml[dnum].z, fl.z);
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
            if (fast && ncu == MAXC && rpc == 3) {
                for (int it = 0; it + 1 < P.iters; it++) sweep(true_type{}, false_type{}, true_type{});
                if (P.iters > 0) sweep(true_type{}, true_type{}, true_type{});
            } else if (fast) {
                for (int it = 0; it + 1 < P.iters; it++) sweep(true_type{}, false_type{}, false_type{});
                if (P.iters > 0) sweep(true_type{}, true_type{}, false_type{});
            } else {
                for (int it = 0; it + 1 < P.iters; it++) sweep(false_type{}, false_type{}, false_type{});
                if (P.iters > 0) sweep(false_type{}, true_type{}, false_type{});
            }
            my_resid = (double)rsum;      // |delta lambda| summed over the last sweep
            // v += h * (M^-1 J^T lambda)
            v.x = fma_(h, fl.x, v.x); v.y = fma_(h, fl.y, v.y); v.z = fma_(h, fl.z, v.z);
            w.x = fma_(h, fa.x, w.x); w.y = fma_(h, fa.y, w.y); w.z = fma_(h, fa.z, w.z);
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
    // ---- diagnostics: wavefront reduction (__shfl_xor), one plain store per wave into the wave's own slot.
    // No atomics: 4096 same-address atomics per launch serialise at ~11 ns each, longer than the whole kernel.
    const int wc = wave_sum<int>(my_contacts);
    const double wr = wave_sum<double>(my_resid);
    if ((i & 63) == 0 && i < n) {        // waves wholly past the range own no slot
        StepDiag d;
        d.contacts = (unsigned long long)wc;
        d.residual = wr;
        diag[i >> 6] = d;
    }
}

}  // namespace dmx
