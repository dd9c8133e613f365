/*
 * oracle/orc_step.c -- dWorldQuickStep restated (TEST INFRASTRUCTURE, see orc.h).
 *
 * Reference call site: dWorldStep(world, physicsTime) at main.c:213, stepped
 * with QuickStep semantics as BASELINE.json's configs name.
 * Everything below is [ODE-recall] of ODE 0.13-0.16:
 *   util.cpp   dxProcessIslands  -> islands()
 *   quickstep.cpp dxQuickStepIsland stages 0-6 -> step_island()
 *   quickstep.cpp SOR_LCP        -> sor_lcp()
 *   contact.cpp dxJointContact::getInfo1/getInfo2 -> contact_rows()
 *   util.cpp   dxStepBody        -> step_body()
 */
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

typedef struct {
    /* per island, sized on demand */
    int *ibody, *ijoint, *stack;
    real *invI, *J, *iMJ, *rhs, *cfm, *lo, *hi, *lambda, *Ad, *fc, *tmp1, *c;
    int *jb, *findex, *order, *local;
    size_t cap_b, cap_m;
    /* adjacency */
    int *adj_off, *adj, *adj_fill;
    size_t cap_adj_b, cap_adj;
} scratch_t;

static scratch_t S;

static void need_bodies(size_t nb)
{
    if (nb <= S.cap_b) return;
    S.cap_b = nb * 2;
    S.invI = (real *)realloc(S.invI, S.cap_b * 12 * sizeof(real));
    S.fc = (real *)realloc(S.fc, S.cap_b * 6 * sizeof(real));
    S.tmp1 = (real *)realloc(S.tmp1, S.cap_b * 6 * sizeof(real));
}

static void need_rows(size_t m)
{
    if (m <= S.cap_m) return;
    S.cap_m = m * 2;
    S.J = (real *)realloc(S.J, S.cap_m * 12 * sizeof(real));
    S.iMJ = (real *)realloc(S.iMJ, S.cap_m * 12 * sizeof(real));
    S.rhs = (real *)realloc(S.rhs, S.cap_m * sizeof(real));
    S.c = (real *)realloc(S.c, S.cap_m * sizeof(real));
    S.cfm = (real *)realloc(S.cfm, S.cap_m * sizeof(real));
    S.lo = (real *)realloc(S.lo, S.cap_m * sizeof(real));
    S.hi = (real *)realloc(S.hi, S.cap_m * sizeof(real));
    S.lambda = (real *)realloc(S.lambda, S.cap_m * sizeof(real));
    S.Ad = (real *)realloc(S.Ad, S.cap_m * sizeof(real));
    S.jb = (int *)realloc(S.jb, S.cap_m * 2 * sizeof(int));
    S.findex = (int *)realloc(S.findex, S.cap_m * sizeof(int));
    S.order = (int *)realloc(S.order, S.cap_m * sizeof(int));
}

/* dxStepBody: x += h v; q += h * 1/2 (0,w) q; normalise; R = R(q) */
static void step_body(orc_body *b, real h)
{
    for (int j = 0; j < 3; j++) b->pos[j] = FMA(h, b->lvel[j], b->pos[j]);
    real dq[4];
    orc_w_to_dq(b->avel, b->q, dq);
    for (int j = 0; j < 4; j++) b->q[j] = FMA(h, dq[j], b->q[j]);
    orc_normalize4(b->q);
    orc_q_to_R(b->q, b->R);
}

/* gyroscopic torque into tacc */
static void gyro_torque(const orc_world *w, orc_body *b, real h)
{
    /* An isotropic inertia tensor (the reference's every body: AddBody leaves ODE's default mass, m = 1, I = identity,
     * main.c:695-733) has no gyroscopic torque: w x (I w) = I (w x w) = 0 exactly.  What the formulas below would add
     * is the rounding of R I R^T and of the 3 x 3 solve -- 1e-8 of noise -- at a few hundred operations per body per tick; both
     * this restatement and the product leave it out (round 4; an early-out, so -0 / +0 questions do not arise: tacc is untouched). */
    if (b->I[0] == b->I[5] && b->I[5] == b->I[10] && b->I[1] == 0 && b->I[2] == 0 && b->I[4] == 0 && b->I[6] == 0 && b->I[8] == 0 &&
        b->I[9] == 0)
        return;
    real tmp[12], I[12];
    orc_mul2_333(tmp, b->I, b->R);      /* I_b R^T   */
    orc_mul0_333(I, b->R, tmp);         /* R I_b R^T */
    if (w->gyro_mode == ORC_GYRO_EXPLICIT) {
        real L[3], c[3];
        orc_mul0_331(L, I, b->avel);
        orc_cross3(c, b->avel, L);
        b->tacc[0] -= c[0]; b->tacc[1] -= c[1]; b->tacc[2] -= c[2];
        return;
    }
    /* implicit (Lacoursiere): Itild = I - h [L]x ; tau = (I Itild^-1 - 1) L / h */
    real L[3];
    orc_mul0_331(L, I, b->avel);
    real It[12];
    memset(It, 0, sizeof(It));
    /* dSetCrossMatrixMinus(It, L, 4) */
    It[1] = L[2];  It[2] = -L[1];
    It[4] = -L[2]; It[6] = L[0];
    It[8] = L[1];  It[9] = -L[0];
    for (int i = 0; i < 12; i++) It[i] = FMA(It[i], h, I[i]);
    real hinv = R(1.0) / h;
    L[0] *= hinv; L[1] *= hinv; L[2] *= hinv;
    real itInv[12];
    if (orc_invert3(itInv, It)) {
        orc_mul0_333(It, I, itInv);
        It[0] -= 1; It[5] -= 1; It[10] -= 1;
        real tau[3];
        orc_mul0_331(tau, It, L);
        b->tacc[0] += tau[0]; b->tacc[1] += tau[1]; b->tacc[2] += tau[2];
    }
}

/* dxJointContact::getInfo1: rows for this contact */
static int contact_m(const orc_joint *j)
{
    real mu = j->mu < 0 ? 0 : j->mu;
    return (mu > 0) ? 3 : 1;
}

/* dxJointContact::getInfo2: fills m rows starting at row r */
static void contact_rows(const orc_world *w, const orc_joint *j, int lb1, int lb2,
                         int r, real fps)
{
    const orc_body *B1 = &w->bodies[j->b1];
    const orc_body *B2 = j->b2 >= 0 ? &w->bodies[j->b2] : NULL;
    int m = contact_m(j);
    real normal[3], c1[3], c2[3] = { 0, 0, 0 };
    for (int k = 0; k < 3; k++) {
        normal[k] = j->reverse ? -j->geom.normal[k] : j->geom.normal[k];
        c1[k] = j->geom.pos[k] - B1->pos[k];
        if (B2) c2[k] = j->geom.pos[k] - B2->pos[k];
    }
    real *J = S.J + 12 * (size_t)r;
    memset(J, 0, (size_t)m * 12 * sizeof(real));
    for (int k = 0; k < m; k++) {
        S.jb[2 * (r + k)] = lb1; S.jb[2 * (r + k) + 1] = lb2;
        S.c[r + k] = 0; S.cfm[r + k] = w->cfm;
        S.findex[r + k] = -1;
    }
    /* normal row */
    J[0] = normal[0]; J[1] = normal[1]; J[2] = normal[2];
    orc_cross3(J + 3, c1, normal);
    if (B2) {
        J[6] = -normal[0]; J[7] = -normal[1]; J[8] = -normal[2];
        orc_cross3(J + 9, c2, normal);
        J[9] = -J[9]; J[10] = -J[10]; J[11] = -J[11];
    }
    real k_erp = fps * w->erp;
    real depth = j->geom.depth;          /* min_depth = 0 */
    if (depth < 0) depth = 0;
    real pushout = k_erp * depth;
    S.c[r] = pushout;                    /* max_vel = infinity */
    if (j->mode & ORC_CONTACT_BOUNCE) {
        real outgoing = orc_dot3(J, B1->lvel) + orc_dot3(J + 3, B1->avel);   /* two fused dots, then one add */
        if (B2) outgoing += orc_dot3(J + 6, B2->lvel) + orc_dot3(J + 9, B2->avel);
        if (j->bounce_vel >= 0 && (-outgoing) > j->bounce_vel) {
            real newc = -j->bounce * outgoing;
            if (newc > S.c[r]) S.c[r] = newc;
        }
    }
    S.lo[r] = 0; S.hi[r] = ORC_INF;
    if (m >= 2) {
        real t1[3], t2[3];
        orc_plane_space(normal, t1, t2);
        real *J1 = J + 12, *J2 = J + 24;
        J1[0] = t1[0]; J1[1] = t1[1]; J1[2] = t1[2];
        orc_cross3(J1 + 3, c1, t1);
        if (B2) {
            J1[6] = -t1[0]; J1[7] = -t1[1]; J1[8] = -t1[2];
            orc_cross3(J1 + 9, c2, t1);
            J1[9] = -J1[9]; J1[10] = -J1[10]; J1[11] = -J1[11];
        }
        S.lo[r + 1] = -j->mu; S.hi[r + 1] = j->mu;
        J2[0] = t2[0]; J2[1] = t2[1]; J2[2] = t2[2];
        orc_cross3(J2 + 3, c1, t2);
        if (B2) {
            J2[6] = -t2[0]; J2[7] = -t2[1]; J2[8] = -t2[2];
            orc_cross3(J2 + 9, c2, t2);
            J2[9] = -J2[9]; J2[10] = -J2[10]; J2[11] = -J2[11];
        }
        S.lo[r + 2] = -j->mu; S.hi[r + 2] = j->mu;
    }
}

/* SOR_LCP */
static double sor_lcp(const orc_world *w, int m, int nb, const int *ibody)
{
    real *J = S.J, *iMJ = S.iMJ, *b = S.rhs, *lambda = S.lambda, *Ad = S.Ad, *fc = S.fc;
    const int *jb = S.jb;
    double resid = 0;
    /* iMJ = inv(M) J^T */
    for (int i = 0; i < m; i++) {
        real *ip = iMJ + 12 * (size_t)i; const real *jp = J + 12 * (size_t)i;
        int b1 = jb[2 * i], b2 = jb[2 * i + 1];
        real k = w->bodies[ibody[b1]].flags & ORC_BODY_KINEMATIC ? 0 : w->bodies[ibody[b1]].invMass;
        for (int j = 0; j < 3; j++) ip[j] = k * jp[j];
        orc_mul0_331(ip + 3, S.invI + 12 * (size_t)b1, jp + 3);
        if (b2 >= 0) {
            k = w->bodies[ibody[b2]].flags & ORC_BODY_KINEMATIC ? 0 : w->bodies[ibody[b2]].invMass;
            for (int j = 0; j < 3; j++) ip[6 + j] = k * jp[6 + j];
            orc_mul0_331(ip + 9, S.invI + 12 * (size_t)b2, jp + 9);
        } else {
            for (int j = 6; j < 12; j++) ip[j] = 0;
        }
    }
    memset(lambda, 0, (size_t)m * sizeof(real));
    memset(fc, 0, (size_t)nb * 6 * sizeof(real));
    /* Ad = w / (diag(J iMJ) + cfm) */
    for (int i = 0; i < m; i++) {
        const real *ip = iMJ + 12 * (size_t)i, *jp = J + 12 * (size_t)i;
        real sum = 0;
        for (int j = 0; j < 6; j++) sum = FMA(ip[j], jp[j], sum);
        if (jb[2 * i + 1] >= 0)
            for (int j = 6; j < 12; j++) sum = FMA(ip[j], jp[j], sum);
        Ad[i] = w->sor_w / (sum + S.cfm[i]);
    }
    /* scale J and b by Ad; Ad *= cfm */
    for (int i = 0; i < m; i++) {
        real *jp = J + 12 * (size_t)i;
        for (int j = 0; j < 12; j++) jp[j] *= Ad[i];
        b[i] *= Ad[i];
        Ad[i] *= S.cfm[i];
    }
    /* rows with findex < 0 first (all of them here) */
    {
        int f = 0, k = 1;
        for (int i = 0; i < m; i++) {
            if (S.findex[i] < 0) S.order[f++] = i;
            else S.order[m - k++] = i;
        }
    }
    for (int it = 0; it < w->iters; it++) {
        if (w->row_order == ORC_ORDER_ODE && (it & 7) == 0) {
            for (int i = 1; i < m; i++) {
                int sw = orc_ode_rand_int(i + 1);
                int t = S.order[i]; S.order[i] = S.order[sw]; S.order[sw] = t;
            }
        }
        int last = (it == w->iters - 1);
        for (int i = 0; i < m; i++) {
            int idx = S.order[i];
            int b1 = jb[2 * idx], b2 = jb[2 * idx + 1];
            real *fc1 = fc + 6 * (size_t)b1;
            real *fc2 = b2 >= 0 ? fc + 6 * (size_t)b2 : NULL;
            real old_lambda = lambda[idx];
            real delta = FMA(-old_lambda, Ad[idx], b[idx]);
            const real *jp = J + 12 * (size_t)idx;
            delta -= FMA(fc1[5], jp[5], FMA(fc1[4], jp[4], FMA(fc1[3], jp[3],
                     FMA(fc1[2], jp[2], FMA(fc1[1], jp[1], fc1[0] * jp[0])))));
            if (fc2)
                delta -= FMA(fc2[5], jp[11], FMA(fc2[4], jp[10], FMA(fc2[3], jp[9],
                         FMA(fc2[2], jp[8], FMA(fc2[1], jp[7], fc2[0] * jp[6])))));
            real lo_act, hi_act;
            if (S.findex[idx] >= 0) {
                hi_act = orc_fabs(S.hi[idx] * lambda[S.findex[idx]]);
                lo_act = -hi_act;
            } else { hi_act = S.hi[idx]; lo_act = S.lo[idx]; }
            real new_lambda = old_lambda + delta;
            if (new_lambda < lo_act) { delta = lo_act - old_lambda; lambda[idx] = lo_act; }
            else if (new_lambda > hi_act) { delta = hi_act - old_lambda; lambda[idx] = hi_act; }
            else lambda[idx] = new_lambda;
            const real *ip = iMJ + 12 * (size_t)idx;
            for (int k = 0; k < 6; k++) fc1[k] = FMA(delta, ip[k], fc1[k]);
            if (fc2) for (int k = 0; k < 6; k++) fc2[k] = FMA(delta, ip[6 + k], fc2[k]);
            if (last) resid += (double)orc_fabs(delta);
        }
    }
    return resid;
}

/* ---------------------------------------------------------------------------------------------------------------
 * dWorldStep's solve [ODE-recall step.cpp dxStepIsland + lcp.cpp dSolveLCP]: the same rows as QuickStep, but the
 * island's system  A lambda = b + w,  A = J M^-1 J^T + diag(cfm / h),  lo <= lambda <= hi,  w_i >= 0 at lo, <= 0 at
 * hi, = 0 between, is solved to complementarity instead of swept 20 times.  With cfm > 0 A is positive definite, so
 * the solution is unique: any exact pivoting method reaches the lambda ODE's Dantzig solver reaches.  Here: block
 * principal pivoting (Judice & Pires) -- rows are free / at lo / at hi; solve the free block by Cholesky, flip every
 * row that violates its condition, and fall back to flipping only the highest violating row when the count of
 * violations has failed to shrink three times (Murty's rule, finite for P-matrices).
 * Every loop below has a fixed operation order; csrc/dmx_islands.hip (lcp_island_wg) follows the same order. */
static struct { real *A, *M, *r, *lam, *wv; int *state, *idx; size_t cap; } L;
enum { LCP_FREE = 0, LCP_LO = 1, LCP_HI = 2 };

static void lcp_need(size_t m)
{
    if (m <= L.cap) return;
    L.cap = m * 2;
    L.A = (real *)realloc(L.A, L.cap * L.cap * sizeof(real));
    L.M = (real *)realloc(L.M, L.cap * L.cap * sizeof(real));
    L.r = (real *)realloc(L.r, L.cap * sizeof(real));
    L.lam = (real *)realloc(L.lam, L.cap * sizeof(real));
    L.wv = (real *)realloc(L.wv, L.cap * sizeof(real));
    L.state = (int *)realloc(L.state, L.cap * sizeof(int));
    L.idx = (int *)realloc(L.idx, L.cap * sizeof(int));
}

static real dot6(const real *a, const real *b, real acc)
{
    for (int k = 0; k < 6; k++) acc = FMA(a[k], b[k], acc);
    return acc;
}

static double exact_lcp(orc_world *w, int m, int nb, const int *ibody)
{
    real *J = S.J, *iMJ = S.iMJ, *b = S.rhs, *lambda = S.lambda, *fc = S.fc;
    const int *jb = S.jb;
    lcp_need((size_t)m);
    /* iMJ = inv(M) J^T */
    for (int i = 0; i < m; i++) {
        real *ip = iMJ + 12 * (size_t)i; const real *jp = J + 12 * (size_t)i;
        int b1 = jb[2 * i], b2 = jb[2 * i + 1];
        real k = w->bodies[ibody[b1]].flags & ORC_BODY_KINEMATIC ? 0 : w->bodies[ibody[b1]].invMass;
        for (int j = 0; j < 3; j++) ip[j] = k * jp[j];
        orc_mul0_331(ip + 3, S.invI + 12 * (size_t)b1, jp + 3);
        if (b2 >= 0) {
            k = w->bodies[ibody[b2]].flags & ORC_BODY_KINEMATIC ? 0 : w->bodies[ibody[b2]].invMass;
            for (int j = 0; j < 3; j++) ip[6 + j] = k * jp[6 + j];
            orc_mul0_331(ip + 9, S.invI + 12 * (size_t)b2, jp + 9);
        } else {
            for (int j = 6; j < 12; j++) ip[j] = 0;
        }
    }
    /* A = J iMJ^T (only rows sharing a body couple) + diag(cfm) */
    real *A = L.A;
    for (int i = 0; i < m; i++) {
        const real *ji = J + 12 * (size_t)i;
        const int i1 = jb[2 * i], i2 = jb[2 * i + 1];
        for (int j = 0; j < m; j++) {
            const real *pj = iMJ + 12 * (size_t)j;
            const int j1 = jb[2 * j], j2 = jb[2 * j + 1];
            real a = 0;
            if (i1 == j1) a = dot6(ji, pj, a);
            if (j2 >= 0 && i1 == j2) a = dot6(ji, pj + 6, a);
            if (i2 >= 0 && i2 == j1) a = dot6(ji + 6, pj, a);
            if (i2 >= 0 && j2 >= 0 && i2 == j2) a = dot6(ji + 6, pj + 6, a);
            A[(size_t)i * m + j] = a;
        }
        A[(size_t)i * m + i] += S.cfm[i];
    }
    real bmax = 0;
    for (int i = 0; i < m; i++) if (orc_fabs(b[i]) > bmax) bmax = orc_fabs(b[i]);
#ifdef ORC_SINGLE
    const real tol = R(1e-5) * (R(1.0) + bmax);
#else
    const real tol = R(1e-11) * (R(1.0) + bmax);
#endif
    int *state = L.state, *idx = L.idx;
    real *lam = L.lam, *M = L.M, *r = L.r, *wv = L.wv;
    for (int i = 0; i < m; i++) { state[i] = LCP_FREE; lam[i] = 0; }
    int best = m + 1, patience = 3, rounds = 0;
    const int max_rounds = 20 * m + 100;
    for (;; rounds++) {
        int nf = 0;
        for (int i = 0; i < m; i++) {
            if (state[i] == LCP_FREE) idx[nf++] = i;
            else lam[i] = state[i] == LCP_LO ? S.lo[i] : S.hi[i];
        }
        /* free block and its right-hand side: r_F = b_F - A_F,clamped lambda_clamped */
        for (int a = 0; a < nf; a++) {
            const int i = idx[a];
            real s = b[i];
            for (int j = 0; j < m; j++) if (state[j] != LCP_FREE && lam[j] != 0) s -= A[(size_t)i * m + j] * lam[j];
            r[a] = s;
            for (int c = 0; c <= a; c++) M[(size_t)a * nf + c] = A[(size_t)i * m + idx[c]];
        }
        /* Cholesky, right-looking, lower triangle in place */
        for (int k = 0; k < nf; k++) {
            real d = M[(size_t)k * nf + k];
            d = orc_sqrt(d > 0 ? d : tol);
            M[(size_t)k * nf + k] = d;
            for (int i = k + 1; i < nf; i++) M[(size_t)i * nf + k] /= d;
            for (int i = k + 1; i < nf; i++) {
                const real lik = M[(size_t)i * nf + k];
                for (int j = k + 1; j <= i; j++) M[(size_t)i * nf + j] -= lik * M[(size_t)j * nf + k];
            }
        }
        /* L y = r (column oriented), L^T x = y */
        for (int k = 0; k < nf; k++) {
            r[k] /= M[(size_t)k * nf + k];
            for (int i = k + 1; i < nf; i++) r[i] -= M[(size_t)i * nf + k] * r[k];
        }
        for (int k = nf - 1; k >= 0; k--) {
            r[k] /= M[(size_t)k * nf + k];
            for (int i = 0; i < k; i++) r[i] -= M[(size_t)k * nf + i] * r[k];
        }
        for (int a = 0; a < nf; a++) lam[idx[a]] = r[a];
        /* w = A lambda - b */
        for (int i = 0; i < m; i++) {
            real s = -b[i];
            for (int j = 0; j < m; j++) s += A[(size_t)i * m + j] * lam[j];
            wv[i] = s;
        }
        /* violations */
        int nv = 0, top = -1;
        for (int i = 0; i < m; i++) {
            int v = 0;
            if (state[i] == LCP_FREE) v = (lam[i] < S.lo[i] - tol) ? 1 : (lam[i] > S.hi[i] + tol) ? 2 : 0;
            else if (state[i] == LCP_LO) v = wv[i] < -tol ? 3 : 0;
            else v = wv[i] > tol ? 3 : 0;
            idx[i] = v;                 /* (idx is rebuilt at the top of the next round) */
            if (v) { nv++; top = i; }
        }
        if (nv == 0 || rounds >= max_rounds) break;
        int all = 1;
        if (nv < best) { best = nv; patience = 3; }
        else if (patience > 0) patience--;
        else all = 0;
        for (int i = 0; i < m; i++) {
            if (!idx[i] || (!all && i != top)) continue;
            state[i] = idx[i] == 1 ? LCP_LO : idx[i] == 2 ? LCP_HI : LCP_FREE;
        }
    }
    w->lcp_rounds = rounds > w->lcp_rounds ? rounds : w->lcp_rounds;
    /* clamp what the tolerance let through, then cforce = inv(M) J^T lambda */
    for (int i = 0; i < m; i++) {
        if (state[i] == LCP_FREE) { if (lam[i] < S.lo[i]) lam[i] = S.lo[i]; if (lam[i] > S.hi[i]) lam[i] = S.hi[i]; }
        lambda[i] = lam[i];
    }
    memset(fc, 0, (size_t)nb * 6 * sizeof(real));
    for (int i = 0; i < m; i++) {
        const real *ip = iMJ + 12 * (size_t)i;
        real *f1 = fc + 6 * (size_t)jb[2 * i];
        for (int k = 0; k < 6; k++) f1[k] = FMA(lambda[i], ip[k], f1[k]);
        if (jb[2 * i + 1] >= 0) {
            real *f2 = fc + 6 * (size_t)jb[2 * i + 1];
            for (int k = 0; k < 6; k++) f2[k] = FMA(lambda[i], ip[6 + k], f2[k]);
        }
    }
    double resid = 0;      /* complementarity residual: what is left of the conditions */
    for (int i = 0; i < m; i++) {
        real v = state[i] == LCP_FREE ? orc_fabs(wv[i]) : (state[i] == LCP_LO ? (wv[i] < 0 ? -wv[i] : 0) : (wv[i] > 0 ? wv[i] : 0));
        resid += (double)v;
    }
    return resid;
}

static void step_island(orc_world *w, const int *ibody, int nb, const int *ijoint, int nj, real h)
{
    real stepsize1 = R(1.0) / h;
    need_bodies((size_t)nb);
    /* stage 0: gravity, world-frame inverse inertia, gyroscopic torque */
    for (int i = 0; i < nb; i++) {
        orc_body *b = &w->bodies[ibody[i]];
        if (!(b->flags & (ORC_BODY_NOGRAVITY | ORC_BODY_KINEMATIC)))
            for (int j = 0; j < 3; j++) b->facc[j] = FMA(b->mass, w->gravity[j], b->facc[j]);
        real tmp[12];
        if (b->flags & ORC_BODY_KINEMATIC) {
            memset(S.invI + 12 * (size_t)i, 0, 12 * sizeof(real));
        } else {
            orc_mul2_333(tmp, b->invI, b->R);
            orc_mul0_333(S.invI + 12 * (size_t)i, b->R, tmp);
            if (w->gyro_mode != ORC_GYRO_OFF) gyro_torque(w, b, h);
        }
    }
    /* rows */
    int m = 0;
    for (int k = 0; k < nj; k++) m += contact_m(&w->joints[ijoint[k]]);
    if (m > 0) {
        need_rows((size_t)m);
        int r = 0;
        for (int k = 0; k < nj; k++) {
            const orc_joint *j = &w->joints[ijoint[k]];
            int lb1 = S.local[j->b1];
            int lb2 = j->b2 >= 0 ? S.local[j->b2] : -1;
            contact_rows(w, j, lb1, lb2, r, stepsize1);
            r += contact_m(j);
        }
        /* rhs = c/h - J (v/h + invM fe) */
        for (int i = 0; i < nb; i++) {
            const orc_body *b = &w->bodies[ibody[i]];
            real im = (b->flags & ORC_BODY_KINEMATIC) ? 0 : b->invMass;
            real *t = S.tmp1 + 6 * (size_t)i;
            for (int j = 0; j < 3; j++) t[j] = FMA(b->facc[j], im, b->lvel[j] * stepsize1);
            orc_mul0_331(t + 3, S.invI + 12 * (size_t)i, b->tacc);
            for (int j = 0; j < 3; j++) t[3 + j] = FMA(b->avel[j], stepsize1, t[3 + j]);
        }
        for (int i = 0; i < m; i++) {
            const real *jp = S.J + 12 * (size_t)i;
            int b1 = S.jb[2 * i], b2 = S.jb[2 * i + 1];
            real sum = 0;
            const real *in = S.tmp1 + 6 * (size_t)b1;
            for (int j = 0; j < 6; j++) sum = FMA(jp[j], in[j], sum);
            if (b2 >= 0) {
                in = S.tmp1 + 6 * (size_t)b2;
                for (int j = 0; j < 6; j++) sum = FMA(jp[6 + j], in[j], sum);
            }
            S.rhs[i] = FMA(S.c[i], stepsize1, -sum);
            S.cfm[i] *= stepsize1;
        }
        w->last_residual += w->stepper == ORC_STEPPER_EXACT ? exact_lcp(w, m, nb, ibody) : sor_lcp(w, m, nb, ibody);
        for (int k = 0, rr = 0; k < nj; k++) {          /* diagnostics: the normal force of every contact */
            w->joints[ijoint[k]].lambda_n = S.lambda[rr];
            rr += contact_m(&w->joints[ijoint[k]]);
        }
        /* v += h * cforce */
        for (int i = 0; i < nb; i++) {
            orc_body *b = &w->bodies[ibody[i]];
            const real *cf = S.fc + 6 * (size_t)i;
            for (int j = 0; j < 3; j++) b->lvel[j] = FMA(h, cf[j], b->lvel[j]);
            for (int j = 0; j < 3; j++) b->avel[j] = FMA(h, cf[3 + j], b->avel[j]);
        }
    }
    /* v += h invM fe; integrate; clear accumulators */
    for (int i = 0; i < nb; i++) {
        orc_body *b = &w->bodies[ibody[i]];
        if (!(b->flags & ORC_BODY_KINEMATIC)) {
            real im = b->invMass;
            for (int j = 0; j < 3; j++) b->lvel[j] = FMA(h * im, b->facc[j], b->lvel[j]);
            for (int j = 0; j < 3; j++) b->tacc[j] *= h;
            orc_muladd0_331(b->avel, S.invI + 12 * (size_t)i, b->tacc);
        }
    }
    for (int i = 0; i < nb; i++) {
        orc_body *b = &w->bodies[ibody[i]];
        step_body(b, h);
        b->facc[0] = b->facc[1] = b->facc[2] = 0;
        b->tacc[0] = b->tacc[1] = b->tacc[2] = 0;
    }
}

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* dxProcessIslands: DFS over joints between bodies */
void orc_quickstep(orc_world *w, real h)
{
    int nb = w->nb, nj = w->nj;
    w->last_residual = 0;
    w->lcp_rounds = 0;
    if (nb == 0) return;
    /* adjacency: joints per body, in creation order */
    if ((size_t)nb + 1 > S.cap_adj_b) {
        S.cap_adj_b = (size_t)nb + 1;
        S.adj_off = (int *)realloc(S.adj_off, S.cap_adj_b * sizeof(int));
        S.adj_fill = (int *)realloc(S.adj_fill, S.cap_adj_b * sizeof(int));
        S.ibody = (int *)realloc(S.ibody, S.cap_adj_b * sizeof(int));
        S.stack = (int *)realloc(S.stack, S.cap_adj_b * sizeof(int));
        S.local = (int *)realloc(S.local, S.cap_adj_b * sizeof(int));
    }
    if ((size_t)2 * nj + 1 > S.cap_adj) {
        S.cap_adj = (size_t)2 * nj + 1;
        S.adj = (int *)realloc(S.adj, S.cap_adj * sizeof(int));
        S.ijoint = (int *)realloc(S.ijoint, S.cap_adj * sizeof(int));
    }
    memset(S.adj_off, 0, ((size_t)nb + 1) * sizeof(int));
    for (int j = 0; j < nj; j++) {
        S.adj_off[w->joints[j].b1 + 1]++;
        if (w->joints[j].b2 >= 0) S.adj_off[w->joints[j].b2 + 1]++;
        w->joints[j].tag = 0;
    }
    for (int i = 0; i < nb; i++) S.adj_off[i + 1] += S.adj_off[i];
    memcpy(S.adj_fill, S.adj_off, (size_t)nb * sizeof(int));
    for (int j = 0; j < nj; j++) {
        S.adj[S.adj_fill[w->joints[j].b1]++] = j;
        if (w->joints[j].b2 >= 0) S.adj[S.adj_fill[w->joints[j].b2]++] = j;
    }
    for (int i = 0; i < nb; i++) w->bodies[i].tag = 0;

    int ode = (w->row_order == ORC_ORDER_ODE);
    for (int s = 0; s < nb; s++) {
        /* ODE's body list is head-inserted: newest body first */
        int bb = ode ? nb - 1 - s : s;
        if (w->bodies[bb].tag) continue;
        w->bodies[bb].tag = 1;
        int bcount = 0, jcount = 0, sp = 0;
        int b = bb;
        for (;;) {
            S.local[b] = bcount;
            S.ibody[bcount++] = b;
            int lo = S.adj_off[b], hi = S.adj_off[b + 1];
            for (int t = 0; t < hi - lo; t++) {
                /* per-body joint list is head-inserted too in ODE */
                int j = ode ? S.adj[hi - 1 - t] : S.adj[lo + t];
                orc_joint *jt = &w->joints[j];
                if (jt->tag) continue;
                jt->tag = 1;
                S.ijoint[jcount++] = j;
                int other = (jt->b1 == b) ? jt->b2 : jt->b1;
                if (other >= 0 && !w->bodies[other].tag) {
                    w->bodies[other].tag = 1;
                    S.stack[sp++] = other;
                }
            }
            if (sp == 0) break;
            b = S.stack[--sp];
        }
        if (!ode) {
            /* ORC_ORDER_FIXED: bodies by index, joints (hence rows) in creation order */
            qsort(S.ibody, (size_t)bcount, sizeof(int), cmp_int);
            qsort(S.ijoint, (size_t)jcount, sizeof(int), cmp_int);
            for (int k = 0; k < bcount; k++) S.local[S.ibody[k]] = k;
        }
        step_island(w, S.ibody, bcount, S.ijoint, jcount, h);
    }
}
