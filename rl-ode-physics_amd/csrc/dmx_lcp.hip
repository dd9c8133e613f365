// dmx_lcp.hip -- dWorldStep's exact solve of a LARGE island, spread over the whole chip.
//
// The reference's tick is dSpaceCollide -> dWorldStep(world, 1/120) -> dJointGroupEmpty (/root/reference/src/main.c:211-215)
// with up to MAX_BODIES = 512 bodies (/root/reference/inc/body.h:6); piled into the pen they are ONE dynamics island of
// 2 000 - 2 600 constraint rows, and dWorldStep [ODE-recall step.cpp dxStepIsland + lcp.cpp dSolveLCP] solves that island's
//      A lambda = b + w,   A = J M^-1 J^T + diag(cfm / h),   lo <= lambda <= hi,   w complementary to lambda
// to the end.  With cfm > 0 A is positive definite and the solution unique, so any exact pivoting method reaches ODE's lambda;
// the CPU oracle's exact_lcp (oracle/, its step source) is the checker (north_star's 1e-5 on positions / quaternions; the order of operations is not
// the oracle's here, and need not be).
//
// Method (what ODE's Dantzig solver does with its `nub` leading unbounded rows, restated for a GPU):
//   1. the island's rows are built by the same phase functions as every other island kernel (dmx_island_rows.hpp);
//   2. rows are permuted: U = rows that can never clamp (lo = -inf, hi = +inf: both friction rows of a contact with mu = inf,
//      which is what the reference's NearCallback asks for, main.c:687) first, B = the rest (normal rows, bounded friction);
//   3. A (permuted, tiles of 64, column-major lower triangle, the right-hand side riding along as one extra ROW so that the
//      forward substitution comes for free) is assembled by a launch over tiles from the body-sharing structure;
//   4. a blocked right-looking Cholesky runs over U's panels only: per panel one launch that factors the 64 x 64 diagonal
//      block (every workgroup for itself, in LDS, one lane per row) and solves its own 64 rows of the panel against it, and one
//      launch of rank-64 updates of the trailing tiles on the matrix cores (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32;
//      one wavefront per 64 x 64 tile).  What is left in B's block is the Schur complement  S = A_BB - A_BU A_UU^-1 A_UB  and
//      the reduced right-hand side  b' = b_B - A_BU A_UU^-1 b_U:  the LCP in B's rows alone, a third of the size;
//   5. block principal pivoting (Judice & Pires; Murty's single flip once the violation count has stalled three times -- the
//      oracle's rule) on  S lambda_B = b' + w_B:  per round the free rows' block of S is gathered and factored by the same two
//      kernels, back-substituted, w_B = S lambda_B - b' and the violations are found by a launch, and the HOST flips (it reads
//      one small array per round).  The active set a contact's rows ended a tick with is where they start the next tick
//      (keyed by body pair and contact ordinal): a resting pile closes in one round;
//   6. lambda_U by back-substitution through U's factor, constraint forces, integration (finish_body).
// Cost per tick at m rows, |B| = m/3: one partial factorisation (m^3/3 less B's own block) + rounds x (|free B|^3/3).
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <unordered_map>

#include "dmx_lcp.hpp"
#include "dmx_island_rows.hpp"

namespace dmx {

int lcp_grid_threshold()
{
    static const int v = [] { const char *e = getenv("DMX_LCP_GRID_ROWS"); const int t = e ? atoi(e) : 1 << 30; return t < 1 ? 1 : t; }();
    return v;
}
int lcp_max_exact_rows()
{
    static const int v = [] { const char *e = getenv("DMX_MAX_EXACT_ROWS"); const int t = e ? atoi(e) : 16384; return t < 1 ? 1 : t; }();
    return v;
}

namespace {

constexpr int NB = 64;                  // tile / panel width
enum : int { ST_FREE = 0, ST_LO = 1, ST_HI = 2 };

// ---- the matrix cores' 16 x 16 x 4 forms: D = A B + C, lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; D's column is
//      l & 15, its row (l >> 4) * 4 + reg in f32 and (l >> 4) + 4 * reg in f64 (cdna_hip_programming.md section 3)
template <class T> struct MF;
template <> struct MF<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) * 4 + reg; }
};
template <> struct MF<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};

// value of `v` in lane `lane` (uniform), to every lane
__device__ __forceinline__ float bcast(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double bcast(double v, int lane)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// 1 / sqrt(x), x > 0: the hardware estimate and Newton steps to the format's precision (the pivots' square roots and
// reciprocals sit on the factorisation's one serial chain: IEEE sqrt + division sequences would triple it)
__device__ __forceinline__ float fast_rsqrt(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    y = y * fma_(-0.5f * x * y, y, 1.5f);
    return y;
}
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma_(-0.5 * x * y, y, 1.5);
    y = y * fma_(-0.5 * x * y, y, 1.5);
    return y;
}

// =========================================================================================================== 1. the island's rows
template <class T>
__global__ __launch_bounds__(512) void lcp_prepare(T *__restrict__ S, const uint8_t *__restrict__ bflags, int64_t stride, IslandSet<T> I,
                                                   StepParams<T> P, int isl, T *__restrict__ tol_out, T tol_rel)
{
    constexpr int WG = 512;
    const int tid = threadIdx.x;
    const T h = P.h, hinv = T(1) / h;
    const int b0 = I.body_off[isl], nb = I.body_off[isl + 1] - b0;
    const int c0 = I.con_off[isl], nc = I.con_off[isl + 1] - c0;
    const int r0 = I.row_off[isl];
    T *bs = I.bscr + (size_t)b0 * BW_COUNT;
    T *rows = I.rows + (size_t)r0 * RW_COUNT;
    int *jb = I.rowjb + 2 * (size_t)r0;
    const int m = nc > 0 ? I.crow[c0 + nc - 1] + contact_rpc(I, P, c0 + nc - 1) : 0;
    for (int k = tid; k < nb; k += WG) stage_body(S, bflags, stride, I, P, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], k);
    __syncthreads();
    for (int c = tid; c < nc; c += WG) contact_rows(S, stride, I, P, rows, jb, c0 + c, I.crow[c0 + c], hinv);
    for (int k = tid; k < nb; k += WG) body_tmp(S, stride, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], hinv);
    __syncthreads();
    T bmax = T(0);
    for (int i = tid; i < m; i += WG) {
        row_setup<T, false>(rows, jb, bs, i, hinv, P.sor_w);
        const T v = tabs(rows[(size_t)i * RW_COUNT + RW_RHS]);
        if (v > bmax) bmax = v;
    }
    __shared__ T red[WG];
    red[tid] = bmax;
    __syncthreads();
    for (int o = WG / 2; o > 0; o >>= 1) {
        if (tid < o && red[tid + o] > red[tid]) red[tid] = red[tid + o];
        __syncthreads();
    }
    if (tid == 0) tol_out[0] = tol_rel * (T(1) + red[0]);      // the oracle's tolerance (its exact_lcp)
}

// =========================================================================================================== 3. A, permuted, in tiles
// Element (r, c) of the matrix lives at A[c * ld + r]; tiles (tr, tc), tr >= tc, of 64 x 64; tile row nt holds ONE more row, the
// right-hand side (row nt * 64; the rest of that tile row is zero).  perm[p] = island row at permuted position p, -1 = padding
// (an identity row: its unknown is zero and touches nobody).  A(i, j) = J_i . (M^-1 J_j^T) over the bodies rows i and j share
// + cfm / h on the diagonal, as the CPU oracle builds it.
template <class T>
__global__ __launch_bounds__(256) void lcp_assemble(const T *__restrict__ rows, const int *__restrict__ jb, const int *__restrict__ perm,
                                                    int nt, T *__restrict__ A, int ld)
{
    const int tr = blockIdx.x, tc = blockIdx.y;
    if (tc > tr) return;
    __shared__ int pr[NB], pc[NB], r1[NB], r2[NB], c1[NB], c2[NB];
    const int tid = threadIdx.x;
    if (tid < NB) {
        const int p = tr < nt ? perm[tr * NB + tid] : (tid == 0 ? -2 : -3);
        pr[tid] = p;
        r1[tid] = p >= 0 ? jb[2 * p] : -7; r2[tid] = p >= 0 ? jb[2 * p + 1] : -7;
    } else if (tid < 2 * NB) {
        const int t = tid - NB, p = perm[tc * NB + t];
        pc[t] = p;
        c1[t] = p >= 0 ? jb[2 * p] : -8; c2[t] = p >= 0 ? jb[2 * p + 1] : -8;
    }
    __syncthreads();
    const int rl = tid & (NB - 1);
    for (int cl = tid >> 6; cl < NB; cl += 4) {
        const int i = pr[rl], j = pc[cl];
        T a = T(0);
        if (i == -2) a = j >= 0 ? rows[(size_t)j * RW_COUNT + RW_RHS] : T(0);
        else if (i == -3) a = T(0);
        else if (i < 0 || j < 0) a = (tr == tc && rl == cl) ? T(1) : T(0);
        else {
            const int i1 = r1[rl], i2 = r2[rl], j1 = c1[cl], j2 = c2[cl];
            const T *ji = rows + (size_t)i * RW_COUNT + RW_J, *pj = rows + (size_t)j * RW_COUNT + RW_IMJ;
            if (i1 == j1) { for (int q = 0; q < 6; q++) a = fma_(ji[q], pj[q], a); }
            if (j2 >= 0 && i1 == j2) { for (int q = 0; q < 6; q++) a = fma_(ji[q], pj[6 + q], a); }
            if (i2 >= 0 && i2 == j1) { for (int q = 0; q < 6; q++) a = fma_(ji[6 + q], pj[q], a); }
            if (i2 >= 0 && j2 >= 0 && i2 == j2) { for (int q = 0; q < 6; q++) a = fma_(ji[6 + q], pj[6 + q], a); }
            if (i == j) a += rows[(size_t)i * RW_COUNT + RW_AD];
        }
        A[(size_t)(tc * NB + cl) * ld + tr * NB + rl] = a;
    }
}

// =========================================================================================================== 4a. panel: factor + solve
constexpr int PANEL_WAVES = 16;          // one column of a 16-column sub-panel per wavefront in the left-looking updates

// Column 16 b + w of a 64 x 64 block held in LDS (own[c * NB + r], lane r = row r), minus what the columns to its left
// contribute:  own(r, 16 b + w) -= sum_{j < 16 b} own(r, j) * L(16 b + w, j),  L(., j) = Lb[j * NB + .] (one broadcast read)
template <class T>
__device__ __forceinline__ void subpanel_left(T *own, const T *Lb, int b, int w, int r)
{
    T p = own[(16 * b + w) * NB + r];
    const int c = 16 * b + w;
#pragma unroll 8
    for (int j = 0; j < 16 * b; j++) p = fma_(-own[j * NB + r], Lb[j * NB + c], p);
    own[c * NB + r] = p;
}

// One workgroup of 16 wavefronts per tile row below panel k.  Every workgroup factors the panel's 64 x 64 diagonal block for
// itself, in LDS, lane r = row r: sixteen columns at a time -- first each wavefront brings ONE of the sixteen columns up to date
// against the columns already factored (left-looking: a broadcast read and a multiply-add per earlier column), then wavefront 0
// factors the sixteen in registers, pivots handed round by v_readlane -- and solves ITS tile of the panel's rows (tile
// k + 1 + blockIdx.x; lane r = a row of that tile) against the factor the same way, one sub-panel behind:  X <- X L_kk^-T.  Workgroup 0 also leaves the
// factored block in Ldiag[k] (the block of A itself stays as it was: late workgroups are still reading it).
template <class T>
__global__ __launch_bounds__(64 * PANEL_WAVES) void lcp_panel(T *__restrict__ A, int ld, int k, T *__restrict__ Ldiag, const T *__restrict__ tolp)
{
    extern __shared__ __align__(16) unsigned char lcp_panel_raw[];
    T *Lb = reinterpret_cast<T *>(lcp_panel_raw), *Xb = Lb + NB * NB, *invd = Xb + NB * NB;
    const int r = threadIdx.x & 63, w = threadIdx.x >> 6;
    const T tol = tolp[0];
    const T *D = A + (size_t)k * NB * ld + (size_t)k * NB;
    const int t = k + 1 + blockIdx.x;
    T *Xg = A + (size_t)k * NB * ld + (size_t)t * NB;
#pragma unroll
    for (int c = w; c < NB; c += PANEL_WAVES) { Lb[c * NB + r] = D[(size_t)c * ld + r]; Xb[c * NB + r] = Xg[(size_t)c * ld + r]; }
    __syncthreads();
    // sixteen factored columns of the diagonal block, in registers (wavefront 0)
    auto factor16 = [&](int b) {
        T p[16];
#pragma unroll
        for (int c = 0; c < 16; c++) p[c] = Lb[(16 * b + c) * NB + r];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int gj = 16 * b + j;
            T ajj = bcast(p[j], gj);
            ajj = ajj > T(0) ? ajj : tol;               // (the oracle's guard: a pivot that rounding pushed below zero)
            const T inv = fast_rsqrt(ajj);
            p[j] = (r == gj) ? ajj * inv : p[j] * inv;
            if (r == gj) invd[gj] = inv;
#pragma unroll
            for (int c = j + 1; c < 16; c++) p[c] = fma_(-p[j], bcast(p[j], 16 * b + c), p[c]);
        }
#pragma unroll
        for (int c = 0; c < 16; c++) Lb[(16 * b + c) * NB + r] = (r >= 16 * b + c) ? p[c] : T(0);
    };
    // sixteen solved columns of this workgroup's tile of the panel (wavefront 1): forward substitution against factored columns
    auto solve16 = [&](int b) {
        T p[16];
#pragma unroll
        for (int c = 0; c < 16; c++) p[c] = Xb[(16 * b + c) * NB + r];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int gj = 16 * b + j;
            p[j] *= invd[gj];
            const T *lr = Lb + gj * NB + 16 * b;
#pragma unroll
            for (int c = j + 1; c < 16; c++) p[c] = fma_(-p[j], lr[c], p[c]);
        }
#pragma unroll
        for (int c = 0; c < 16; c++) Xb[(16 * b + c) * NB + r] = p[c];
    };
    // The tile's solve runs one sub-panel behind the factorisation, beside it: while wavefront 0 factors columns 16 b .. 16 b + 15 of the
    // diagonal block, wavefront 1 solves the tile's columns 16 (b - 1) .., and the left-looking updates of both (every wavefront a column)
    // share the phase before.  Ten barriers a panel; the 64 serial pivots of wavefront 0 are what is left on the chain.
    for (int b = 0; b <= 4; b++) {
        if (b >= 1 && b < 4) subpanel_left(Lb, Lb, b, w, r);
        if (b >= 2) subpanel_left(Xb, Lb, b - 1, w, r);
        if (b >= 1) __syncthreads();
        if (w == 0 && b < 4) factor16(b);
        if (w == 1 && b >= 1) solve16(b - 1);
        __syncthreads();
    }
#pragma unroll
    for (int c = w; c < NB; c += PANEL_WAVES) Xg[(size_t)c * ld + r] = Xb[c * NB + r];
    if (blockIdx.x == 0) {
        T *Lo = Ldiag + (size_t)k * NB * NB;
#pragma unroll
        for (int c = w; c < NB; c += PANEL_WAVES) Lo[c * NB + r] = Lb[c * NB + r];
    }
}

// =========================================================================================================== 4b. trailing update
// X(tr, tc) -= L(tr, k) L(tc, k)^T for every tile tr >= tc > k (tr up to the right-hand side's tile row nt): one workgroup of four
// wavefronts per 64 x 64 tile, each wavefront a strip of 16 of the tile's columns: 1 x 4 blocks of 16 x 16 accumulators, 16 steps
// of k = 4.  The matrix cores' "column" index (l & 15) runs along a tile's rows -- the contiguous direction of the storage -- so
// every operand fetch is 16 consecutive reals per quarter-wave, straight from L2 (a panel is a few hundred KB and every tile of
// a tile row reads the same slice).
template <class T>
__global__ __launch_bounds__(256) void lcp_syrk(T *__restrict__ A, int ld, int k)
{
    const int tr = k + 1 + blockIdx.x, tc = k + 1 + blockIdx.y;
    if (tc > tr) return;
    typedef typename MF<T>::acc_t acc_t;
    const int l = threadIdx.x & 63, ic = threadIdx.x >> 6, lc = l & 15, lk = l >> 4;
    T *X = A + (size_t)(tc * NB + ic * 16) * ld + (size_t)tr * NB;
    acc_t acc[4];
#pragma unroll
    for (int ir = 0; ir < 4; ir++)
#pragma unroll
        for (int g = 0; g < 4; g++) acc[ir][g] = X[(size_t)MF<T>::row(l, g) * ld + ir * 16 + lc];
    const T *Lr = A + (size_t)k * NB * ld + (size_t)tr * NB, *Lc = A + (size_t)k * NB * ld + (size_t)tc * NB + ic * 16;
    // every operand of the 16 steps is fetched before the first product: one round trip to L2, not one per step (the products
    // themselves are a microsecond; a tile's time is its memory latency)
    T fa[NB / 4], fb[NB / 4][4];
#pragma unroll
    for (int kk = 0; kk < NB / 4; kk++) {
        const size_t o = (size_t)(kk * 4 + lk) * ld + lc;
        fa[kk] = Lc[o];
#pragma unroll
        for (int i = 0; i < 4; i++) fb[kk][i] = Lr[o + i * 16];
    }
#pragma unroll
    for (int kk = 0; kk < NB / 4; kk++) {
#pragma unroll
        for (int ir = 0; ir < 4; ir++) acc[ir] = MF<T>::mfma(-fa[kk], fb[kk][ir], acc[ir]);
    }
#pragma unroll
    for (int ir = 0; ir < 4; ir++)
#pragma unroll
        for (int g = 0; g < 4; g++) X[(size_t)MF<T>::row(l, g) * ld + ir * 16 + lc] = acc[ir][g];
}

// =========================================================================================================== the reduced problem
// S (B's block of A after U's panels, lower tiles) -> a dense symmetric array Sd[c * lds + r] of its own
template <class T>
__global__ __launch_bounds__(256) void lcp_extract(const T *__restrict__ A, int ld, int nuP, T *__restrict__ Sd, int lds)
{
    const int tr = blockIdx.x, tc = blockIdx.y, tid = threadIdx.x, rl = tid & (NB - 1);
    const int r = tr * NB + rl;
    for (int cl = tid >> 6; cl < NB; cl += 4) {
        const int c = tc * NB + cl;
        Sd[(size_t)c * lds + r] = r >= c ? A[(size_t)(nuP + c) * ld + nuP + r] : A[(size_t)(nuP + r) * ld + nuP + c];
    }
}
// b' (the right-hand side's row after U's panels) and the bounds of B's rows
template <class T>
__global__ __launch_bounds__(256) void lcp_bvec(const T *__restrict__ A, int ld, int nuP, int mP, const int *__restrict__ perm,
                                                const T *__restrict__ rows, int nbdP, T *__restrict__ bprime, T *__restrict__ lo, T *__restrict__ hi)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nbdP) return;
    const int p = perm[nuP + i];
    bprime[i] = A[(size_t)(nuP + i) * ld + mP];
    lo[i] = p >= 0 ? rows[(size_t)p * RW_COUNT + RW_LO] : -Limits<T>::inf();
    hi[i] = p >= 0 ? rows[(size_t)p * RW_COUNT + RW_HI] : Limits<T>::inf();
}
// lambda_B at the start of a round: clamped rows at their bounds, free rows zero
template <class T>
__global__ __launch_bounds__(256) void lcp_clamped(const int *__restrict__ state, const T *__restrict__ lo, const T *__restrict__ hi, int n,
                                                   T *__restrict__ lam)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = state[i];
    lam[i] = s == ST_LO ? lo[i] : s == ST_HI ? hi[i] : T(0);
}
// the same with the rows listed in `except` (the volatile set: the LDS solve accounts for those itself) left at zero
template <class T>
__global__ __launch_bounds__(256) void lcp_clamped_except(const int *__restrict__ state, const T *__restrict__ lo, const T *__restrict__ hi, int n,
                                                          const int *__restrict__ except, int n_except, T *__restrict__ lam)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = state[i];
    T l = s == ST_LO ? lo[i] : s == ST_HI ? hi[i] : T(0);
    for (int e = 0; e < n_except; e++) if (except[e] == i) l = T(0);
    lam[i] = l;
}
// out = bvec - Sd v   (CLASSIFY = false: a round's right-hand side),   or
// out = Sd v - bvec = w and every row's verdict (CLASSIFY = true): 0 fine, 1 free row below lo, 2 free row above hi, 3 clamped
// row whose w has the wrong sign -- the oracle's tests and tolerance.  64 rows a workgroup, the columns in four fixed segments.
template <class T, bool CLASSIFY>
__global__ __launch_bounds__(256) void lcp_gemv(const T *__restrict__ Sd, int lds, int n, const T *__restrict__ v, const T *__restrict__ bvec,
                                                T *__restrict__ out, const int *__restrict__ state, const T *__restrict__ lo,
                                                const T *__restrict__ hi, const T *__restrict__ tolp, int *__restrict__ viol)
{
    __shared__ T part[4][NB];
    const int tid = threadIdx.x, il = tid & (NB - 1), seg = tid >> 6;
    const int i = blockIdx.x * NB + il;
    const int per = (n + 3) / 4, j0 = seg * per, j1 = (j0 + per < n) ? j0 + per : n;
    T s = T(0);
    if (i < n)
        for (int j = j0; j < j1; j++) s = fma_(Sd[(size_t)j * lds + i], v[j], s);
    part[seg][il] = s;
    __syncthreads();
    if (seg != 0 || i >= n) return;
    s = ((part[0][il] + part[1][il]) + part[2][il]) + part[3][il];
    if (!CLASSIFY) { out[i] = bvec[i] - s; return; }
    const T w = s - bvec[i], tol = tolp[0], lam = v[i];
    out[i] = w;
    const int st = state[i];
    int vi = 0;
    if (st == ST_FREE) vi = (lam < lo[i] - tol) ? 1 : (lam > hi[i] + tol) ? 2 : 0;
    else if (st == ST_LO) vi = w < -tol ? 3 : 0;
    else vi = w > tol ? 3 : 0;
    viol[i] = vi;
}
// the free rows' block of S and their right-hand side, in the factorisation's layout (fidx: the free rows in order, -1 = padding)
template <class T>
__global__ __launch_bounds__(256) void lcp_gather(const T *__restrict__ Sd, int lds, const int *__restrict__ fidx, int nft,
                                                  const T *__restrict__ rr, T *__restrict__ Mw, int ldw)
{
    const int tr = blockIdx.x, tc = blockIdx.y;
    if (tc > tr) return;
    __shared__ int fr[NB], fc[NB];
    const int tid = threadIdx.x;
    if (tid < NB) fr[tid] = tr < nft ? fidx[tr * NB + tid] : (tid == 0 ? -2 : -3);
    else if (tid < 2 * NB) fc[tid - NB] = fidx[tc * NB + tid - NB];
    __syncthreads();
    const int rl = tid & (NB - 1);
    for (int cl = tid >> 6; cl < NB; cl += 4) {
        const int i = fr[rl], j = fc[cl];
        T a;
        if (i == -2) a = j >= 0 ? rr[j] : T(0);
        else if (i == -3) a = T(0);
        else if (i < 0 || j < 0) a = (tr == tc && rl == cl) ? T(1) : T(0);
        else a = Sd[(size_t)j * lds + i];
        Mw[(size_t)(tc * NB + cl) * ldw + tr * NB + rl] = a;
    }
}

// =========================================================================================================== back-substitution
// L^T x = y over nt panels of a factored matrix (the strictly lower tiles in A, the diagonal blocks in Ldiag), one workgroup:
// per panel, last to first, the columns' dot products with the part of x already known (a wavefront per column, lanes along the
// rows: contiguous), then the 64 x 64 triangle by one wavefront (lane = unknown).  y[c] = yv[c * ystride].
// scatter != null: x goes to out[scatter[a]] for the entries with scatter[a] >= 0 (a round's free rows back into lambda_B);
// otherwise out[a] = x[a].
template <class T>
__global__ __launch_bounds__(1024) void lcp_backsolve(const T *__restrict__ A, int ld, const T *__restrict__ Ldiag, int nt,
                                                      const T *__restrict__ yv, size_t ystride, const int *__restrict__ scatter,
                                                      T *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char lcp_bs_raw[];
    constexpr int LP = NB + 1;
    T *xs = reinterpret_cast<T *>(lcp_bs_raw);              // [nt * NB]: y, overwritten by x panel by panel
    T *Lb0 = xs + (size_t)nt * NB;                          // two diagonal blocks [NB * LP]: the one in use, the next one arriving
    T *v = Lb0 + 2 * NB * LP;                               // [NB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = nt * NB;
    for (int a = tid; a < n; a += 1024) xs[a] = yv[(size_t)a * ystride];
    {
        const T *Ld = Ldiag + (size_t)(nt - 1) * NB * NB;
        for (int e = tid; e < NB * NB; e += 1024) Lb0[(e >> 6) * LP + (e & 63)] = Ld[e];     // L(r, c) at Lb[c * LP + r]
    }
    __syncthreads();
    int cur = 0;
    for (int kp = nt - 1; kp >= 0; kp--) {
        T *Lb = Lb0 + cur * NB * LP, *Ln = Lb0 + (cur ^ 1) * NB * LP;
        T pre[4];
        if (kp > 0) {
            const T *Ld = Ldiag + (size_t)(kp - 1) * NB * NB;
#pragma unroll
            for (int q = 0; q < 4; q++) pre[q] = Ld[tid + 1024 * q];
        }
        {   // four columns a wavefront, side by side (four independent chains of loads)
            const T *c0 = A + (size_t)(kp * NB + wave) * ld, *c1 = c0 + (size_t)16 * ld, *c2 = c1 + (size_t)16 * ld, *c3 = c2 + (size_t)16 * ld;
            T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
#pragma unroll 2
            for (int r = (kp + 1) * NB + lane; r < n; r += 64) {
                const T x = xs[r];
                s0 = fma_(c0[r], x, s0); s1 = fma_(c1[r], x, s1); s2 = fma_(c2[r], x, s2); s3 = fma_(c3[r], x, s3);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); s3 += __shfl_xor(s3, o, 64);
            }
            if (lane == 0) {
                v[wave] = xs[kp * NB + wave] - s0; v[wave + 16] = xs[kp * NB + wave + 16] - s1;
                v[wave + 32] = xs[kp * NB + wave + 32] - s2; v[wave + 48] = xs[kp * NB + wave + 48] - s3;
            }
        }
        if (kp > 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) { const int e = tid + 1024 * q; Ln[(e >> 6) * LP + (e & 63)] = pre[q]; }
        }
        __syncthreads();
        if (wave == 0) {
            T acc = v[lane];
            const T invd = T(1) / Lb[lane * LP + lane];
            T mine = T(0);
            for (int rr = NB - 1; rr >= 0; rr--) {
                const T xr = bcast(acc * invd, rr);
                if (lane == rr) mine = xr;
                if (lane < rr) acc = fma_(-Lb[lane * LP + rr], xr, acc);
            }
            xs[kp * NB + lane] = mine;
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int a = tid; a < n; a += 1024) {
        if (scatter) { const int d = scatter[a]; if (d >= 0) out[d] = xs[a]; }
        else out[a] = xs[a];
    }
}

// z = y_U - L_BU^T lambda_B, the right-hand side of U's back-substitution: a wavefront per column of U
template <class T>
__global__ __launch_bounds__(64) void lcp_zvec(const T *__restrict__ A, int ld, int nuP, int mP, int nbd, const T *__restrict__ lamB,
                                               T *__restrict__ z)
{
    const int c = blockIdx.x, lane = threadIdx.x;
    const T *col = A + (size_t)c * ld;
    T s = T(0);
    for (int i = lane; i < nbd; i += 64) s = fma_(col[nuP + i], lamB[i], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) z[c] = col[mP] - s;
}

// =========================================================================================================== 6. forces, integration
// lambda (U's from the back-substitution, B's from the round) into the rows, then the constraint force of every body
//   cforce = M^-1 J^T lambda   (rows in creation order per body: bodyrows lists them, entry = 2 row + side)
// and then either (FINISH = false) every bounded row's  w = J cforce + (cfm / h) lambda - rhs  from the rows themselves -- the
// quantity the complementarity conditions are about, free of the cancellation the Schur complement's entries carry when cfm is
// tiny -- and its verdict (lcp_gemv's codes), or (FINISH = true) the integration of the island's bodies (finish_body).
template <class T, bool FINISH>
__global__ __launch_bounds__(1024) void lcp_forces(T *__restrict__ S, const uint8_t *__restrict__ bflags, int64_t stride, IslandSet<T> I,
                                                   StepParams<T> P, int isl, const int *__restrict__ perm, int nuP, int mP, int nbd,
                                                   const T *__restrict__ xU, const T *__restrict__ lamB, T *__restrict__ wB,
                                                   const int *__restrict__ state, const int *__restrict__ boff, const int *__restrict__ bodyrows,
                                                   const T *__restrict__ tolp, int *__restrict__ viol, StepDiag *__restrict__ diag)
{
    constexpr int WG = 1024;
    const int tid = threadIdx.x;
    const int b0 = I.body_off[isl], nb = I.body_off[isl + 1] - b0;
    const int c0 = I.con_off[isl], nc = I.con_off[isl + 1] - c0;
    const int r0 = I.row_off[isl];
    T *bs = I.bscr + (size_t)b0 * BW_COUNT;
    T *rows = I.rows + (size_t)r0 * RW_COUNT;
    const int *jb = I.rowjb + 2 * (size_t)r0;
    double resid = 0.0;
    for (int p = tid; p < mP; p += WG) {
        const int i = perm[p];
        if (i < 0) continue;
        T l;
        if (p < nuP) l = xU[p];
        else {
            const int q = p - nuP;
            l = lamB[q];
            if (FINISH) {
                const int st = state[q];
                const T w = wB[q];
                if (st == ST_FREE) {      // clamp what the tolerance let through (the oracle does)
                    const T lo = rows[(size_t)i * RW_COUNT + RW_LO], hi = rows[(size_t)i * RW_COUNT + RW_HI];
                    if (l < lo) l = lo;
                    if (l > hi) l = hi;
                    resid += (double)tabs(w);
                } else resid += (double)(st == ST_LO ? (w < T(0) ? -w : T(0)) : (w > T(0) ? w : T(0)));
            }
        }
        rows[(size_t)i * RW_COUNT + RW_LAM] = l;
    }
    __syncthreads();
    for (int k = tid; k < nb; k += WG) {
        T f[6] = { T(0), T(0), T(0), T(0), T(0), T(0) };
        for (int e = boff[k]; e < boff[k + 1]; e++) {
            const int code = bodyrows[e], i = code >> 1, side = code & 1;
            const T *ip = rows + (size_t)i * RW_COUNT + RW_IMJ + 6 * side;
            const T lam = rows[(size_t)i * RW_COUNT + RW_LAM];
#pragma unroll
            for (int q = 0; q < 6; q++) f[q] = fma_(lam, ip[q], f[q]);
        }
        T *b = bs + (size_t)k * BW_COUNT;
#pragma unroll
        for (int q = 0; q < 6; q++) b[BW_FC + q] = f[q];
        if (FINISH) finish_body(S, bflags, stride, b, I.bodies[b0 + k], true, P.h);
    }
    if (FINISH) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) resid += __shfl_xor(resid, o, 64);
        if ((tid & 63) == 0 && resid != 0.0) atomicAdd(&diag->residual, resid);
        if (tid == 0) atomicAdd(&diag->contacts, (unsigned long long)nc);
        return;
    }
    __syncthreads();
    const T tol = tolp[0];
    for (int q = tid; q < nbd; q += WG) {
        const int i = perm[nuP + q];
        const T *row = rows + (size_t)i * RW_COUNT;
        const int l1 = jb[2 * i], l2 = jb[2 * i + 1];
        const T *f1 = bs + (size_t)l1 * BW_COUNT + BW_FC;
        const T lam = row[RW_LAM];
        T w = fma_(row[RW_AD], lam, -row[RW_RHS]);
#pragma unroll
        for (int k = 0; k < 6; k++) w = fma_(row[RW_J + k], f1[k], w);
        if (l2 >= 0) {
            const T *f2 = bs + (size_t)l2 * BW_COUNT + BW_FC;
#pragma unroll
            for (int k = 0; k < 6; k++) w = fma_(row[RW_J + 6 + k], f2[k], w);
        }
        wB[q] = w;
        const int st = state[q];
        int vi = 0;
        if (st == ST_FREE) vi = (lam < row[RW_LO] - tol) ? 1 : (lam > row[RW_HI] + tol) ? 2 : 0;
        else if (st == ST_LO) vi = w < -tol ? 3 : 0;
        else vi = w > tol ? 3 : 0;
        viol[q] = vi;
    }
}

// =========================================================================================================== small and medium islands
// One workgroup per island, the whole solve in LDS: the same method as the grid solve above (never-clamping rows first, eliminated
// once; block principal pivoting on the Schur complement in the bounded rows; lambda_U by back-substitution), as an LDL^T on a
// packed lower triangle -- no square roots and ONE barrier per pivot: at step k every thread reads the pivot M(k,k) and the column
// below it, which the previous step's barrier made final, and updates its share of the trailing triangle
//   M(i,j) -= M(i,k) M(j,k) / M(k,k),   i > k, k < j <= i;
// column k stays as it is (c_ik = l_ik d_k), the right-hand side rides along as the last row (it ends as y = L^-1 b).
// Islands of up to a few hundred rows in the reference's pen while the pile is still loose (/root/reference/src/main.c:213).
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }      // j <= i

// steps [k0, k1) of the LDL^T of the packed matrix M whose rows 0 .. last are stored (row `last` = the right-hand side; the
// unknowns are rows / columns 0 .. last - 1); dinv[k] = 1 / pivot.  Ends with a barrier.
// FOUR pivots a pass (round 4; one a pass before: a 93-row solve was 93 passes of a few hundred cycles' work between two barriers,
// 0.2 us each whatever the work, and the volatile rows' solve refactors ten times a call).  A pass: (A) every thread eliminates the
// 4 x 4 pivot block for itself (ten values, read before anybody writes them), and the thread that owns a row below the block brings
// the row's four panel entries up to date -- entry q by the pivots before q, exactly the products the one-pivot passes made, in
// their order -- leaves them in the matrix and in a small buffer; barrier; (B) the trailing triangle takes the four updates of
// every element in one read-modify-write; barrier.  Same operands, same order per element: same bits.
template <class T> constexpr int LDLT_MAXN = sizeof(T) == 8 ? 192 : 288;       // rows (the right-hand side's included) an LDS solve holds at most
template <class T, int WG>
__device__ __forceinline__ void ldlt_steps(T *M, int last, int k0, int k1, T *dinv, T tol, int tid)
{
    constexpr int NBK = 4;
    __shared__ T pn[LDLT_MAXN<T>][NBK];
    const int ty = tid >> 4, tx = tid & 15;
    for (int k = k0; k < k1; k += NBK) {
        const int nb = k1 - k < NBK ? k1 - k : NBK;
        // ---- (A) the pivot block, by everybody
        T A[NBK][NBK], inv[NBK];
#pragma unroll
        for (int p = 0; p < NBK; p++)
#pragma unroll
            for (int q = 0; q <= p; q++) A[p][q] = p < nb ? M[tri(k + p, k + q)] : T(0);
#pragma unroll
        for (int q = 0; q < NBK; q++) {
            T d = A[q][q];
            d = d > T(0) ? d : tol;               // (the oracle's guard: a pivot that rounding pushed below zero)
            inv[q] = T(1) / d;
#pragma unroll
            for (int p = q + 1; p < NBK; p++) {
                const T c = A[p][q] * inv[q];
#pragma unroll
                for (int j = q + 1; j <= p; j++) A[p][j] = fma_(-c, A[j][q], A[p][j]);
            }
        }
        if (tid < nb) dinv[k + tid] = tid == 0 ? inv[0] : tid == 1 ? inv[1] : tid == 2 ? inv[2] : inv[3];
        // the rows below the block: their panel entries
        for (int i = k + nb + tid; i <= last; i += WG) {
            T *Mi = M + tri(i, k);
            T r[NBK];
#pragma unroll
            for (int q = 0; q < NBK; q++) r[q] = q < nb ? Mi[q] : T(0);
#pragma unroll
            for (int q = 0; q < NBK; q++) {
                const T c = r[q] * inv[q];
#pragma unroll
                for (int j = q + 1; j < NBK; j++) r[j] = fma_(-c, A[j][q], r[j]);
            }
#pragma unroll
            for (int q = 0; q < NBK; q++) { pn[i][q] = r[q]; if (q >= 1 && q < nb) Mi[q] = r[q]; }
        }
        __syncthreads();
        // ---- (B) the block's own rows keep their final values; the trailing triangle
        if (tid >= 1 && tid < nb) {
            T *Mp = M + tri(k + tid, k);
#pragma unroll
            for (int p = 1; p < NBK; p++)
                if (tid == p) {
#pragma unroll
                    for (int q = 1; q <= p; q++) Mp[q] = A[p][q];
                }
        }
        for (int i = k + nb + ty; i <= last; i += WG / 16) {
            T c[NBK];
#pragma unroll
            for (int q = 0; q < NBK; q++) c[q] = pn[i][q] * inv[q];
            T *Mi = M + tri(i, 0);
            const int jmax = i < last ? i : last - 1;
            for (int j = k + nb + tx; j <= jmax; j += 16) {
                T v = Mi[j];
#pragma unroll
                for (int q = 0; q < NBK; q++)
                    if (q < nb) v = fma_(-c[q], pn[j][q], v);
                Mi[j] = v;
            }
        }
        __syncthreads();
    }
}
// L^T x = z in place over unknowns [0, n) of a packed LDL^T (L(i,k) = M(i,k) dinv[k]); z in LDS.  Ends with a barrier.
// Every z[k] takes its updates in descending i, one product each: 64 unknowns at a time, their own triangle by one wavefront
// (a lane an unknown, the solved value handed round by a shuffle: no barrier per unknown), then everybody below takes the 64.
template <class T, int WG>
__device__ __forceinline__ void ldlt_backsub(const T *M, int n, const T *dinv, T *z, int tid)
{
    const int lane = tid & 63;
    for (int e = n; e > 0; e -= 64) {
        const int s = e > 64 ? e - 64 : 0;
        if (tid < 64) {
            const int k = s + lane;
            T zk = k < e ? z[k] : T(0);
            const T dk = k < e ? dinv[k] : T(0);
            for (int i = e - 1; i > s; i--) {
                const T xi = __shfl(zk, i - s, 64);
                if (k < i) zk = fma_(-M[tri(i, k)] * dk, xi, zk);
            }
            if (k < e) z[k] = zk;
        }
        __syncthreads();
        for (int k = tid; k < s; k += WG) {
            T zk = z[k];
            const T dk = dinv[k];
            for (int i = e - 1; i >= s; i--) zk = fma_(-M[tri(i, k)] * dk, z[i], zk);
            z[k] = zk;
        }
        __syncthreads();
    }
}

// Block principal pivoting on the bounded rows' problem held in LDS: S = rows / columns nu .. nu + nbd - 1 of the packed matrix M,
// b' = its row m from column nu.  state: in = where the active set starts, out = where it ended; lam / wv: the solution and
// w = S lam - b'.  W: room for a packed (nbd + 1)-row matrix; rd: 2 nbd reals.  The oracle's rule: every violating row flips;
// when the count of violations has failed to shrink three times, only the highest violating row does (Murty).  Returns the
// number of rounds.  Ends after a barrier.
template <class T, int WG>
__device__ __forceinline__ int lds_pivot_rounds(const T *M, int m, int nu, int nbd, T *W, T *lam, T *wv, const T *lo, const T *hi, int *state,
                                                int *fidx, int *viol, T *rd, T tol, int murty_only, int max_rounds, int tid)
{
    __shared__ int s_cnt[2][WG / 64 + 1];
    __shared__ int s_nv, s_top;
    const int lane = tid & 63, wave = tid >> 6;
    int round = 0;
    int best = m + 1, patience = murty_only ? 0 : 3;
    if (nbd <= 0) return 0;
    for (;; round++) {
        // the free rows in order
        int nf = 0;
        {
            int fbase = 0;
            for (int base = 0; base < nbd; base += WG) {
                const int q = base + tid;
                const bool f = q < nbd && state[q] == ST_FREE;
                const unsigned long long bf = __ballot(f);
                if (lane == 0) s_cnt[0][wave] = __popcll(bf);
                __syncthreads();
                int off = fbase;
                for (int e = 0; e < wave; e++) off += s_cnt[0][e];
                if (f) fidx[off + __popcll(bf & ((1ull << lane) - 1ull))] = q;
                else if (q < nbd) lam[q] = state[q] == ST_LO ? lo[q] : hi[q];
                for (int e = 0; e < WG / 64; e++) fbase += s_cnt[0][e];
                __syncthreads();
            }
            nf = fbase;
        }
        // W = S[F, F] with the right-hand side b'_F - S_FC lambda_C as its last row
        for (int a = tid; a < nf; a += WG) {
            T *Wa = W + tri(a, 0);
            const int qa = fidx[a];
            for (int c = 0; c <= a; c++) Wa[c] = M[tri(nu + qa, nu + fidx[c])];
        }
        for (int c = tid; c <= nf; c += WG) {
            T r = T(0);
            if (c < nf) {
                const int qc = fidx[c];
                r = M[tri(m, nu + qc)];
                for (int e = 0; e < nbd; e++)
                    if (state[e] != ST_FREE && lam[e] != T(0)) r = fma_(-(e >= qc ? M[tri(nu + e, nu + qc)] : M[tri(nu + qc, nu + e)]), lam[e], r);
            }
            W[tri(nf, c)] = r;
        }
        __syncthreads();
        ldlt_steps<T, WG>(W, nf, 0, nf, rd, tol, tid);
        T *x = rd + nbd;          // (z has room for max(nu, 2 nbd))
        for (int a = tid; a < nf; a += WG) x[a] = W[tri(nf, a)] * rd[a];
        __syncthreads();
        ldlt_backsub<T, WG>(W, nf, rd, x, tid);
        for (int a = tid; a < nf; a += WG) lam[fidx[a]] = x[a];
        __syncthreads();
        // w_B = S lambda_B - b', verdicts
        if (tid == 0) { s_nv = 0; s_top = -1; }
        __syncthreads();
        for (int q = tid; q < nbd; q += WG) {
            T sacc = -M[tri(m, nu + q)];
            for (int e = 0; e < nbd; e++) sacc = fma_(e <= q ? M[tri(nu + q, nu + e)] : M[tri(nu + e, nu + q)], lam[e], sacc);
            wv[q] = sacc;
            const int st = state[q];
            int vi = 0;
            if (st == ST_FREE) vi = (lam[q] < lo[q] - tol) ? 1 : (lam[q] > hi[q] + tol) ? 2 : 0;
            else if (st == ST_LO) vi = sacc < -tol ? 3 : 0;
            else vi = sacc > tol ? 3 : 0;
            viol[q] = vi;
            if (vi) { atomicAdd(&s_nv, 1); atomicMax(&s_top, q); }
        }
        __syncthreads();
        const int nviol = s_nv, top = s_top;
        if (nviol == 0 || round >= max_rounds) break;
        bool all = true;
        if (nviol < best) { best = nviol; if (!murty_only) patience = 3; }
        else if (patience > 0) patience--;
        else all = false;
        if (murty_only) all = false;
        for (int q = tid; q < nbd; q += WG) {
            const int vi = viol[q];
            if (!vi || (!all && q != top)) continue;
            state[q] = vi == 1 ? ST_LO : vi == 2 ? ST_HI : ST_FREE;
        }
        __syncthreads();
    }
    return round + 1;
}

// LDS of lcp_island_lds: M (m + 1 packed rows), W (nbd + 1), lam / w / lo / hi [nbd], dinv [nu], z [max(nu, 2 nbd)]; then the ints
__host__ __device__ inline size_t lcp_lds_reals(int m, int nbd)
{
    const int nu = m - nbd;
    return (size_t)(m + 1) * (m + 2) / 2 + (size_t)(nbd + 1) * (nbd + 2) / 2 + (size_t)4 * nbd + (size_t)nu + (size_t)(nu > 2 * nbd ? nu : 2 * nbd) + 8;
}
template <class T> __host__ __device__ inline size_t lcp_lds_bytes(int m, int nbd)
{
    return ((lcp_lds_reals(m, nbd) * sizeof(T) + 15) / 16) * 16 + ((size_t)3 * m + (size_t)3 * nbd + 16) * sizeof(int);
}

template <class T, int WG>
__global__ __launch_bounds__(WG) void lcp_island_lds(T *__restrict__ S, const uint8_t *__restrict__ bflags, int64_t stride, IslandSet<T> I,
                                                     StepParams<T> P, StepDiag *__restrict__ diag, int murty_only, T tol_rel)
{
    const int isl = I.big_list[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T h = P.h, hinv = T(1) / h;
    const int b0 = I.body_off[isl], nb = I.body_off[isl + 1] - b0;
    const int c0 = I.con_off[isl], nc = I.con_off[isl + 1] - c0;
    const int r0 = I.row_off[isl];
    T *bs = I.bscr + (size_t)b0 * BW_COUNT;
    T *rows = I.rows + (size_t)r0 * RW_COUNT;
    int *jb = I.rowjb + 2 * (size_t)r0;
    const int m = nc > 0 ? I.crow[c0 + nc - 1] + contact_rpc(I, P, c0 + nc - 1) : 0;

    for (int k = tid; k < nb; k += WG) stage_body(S, bflags, stride, I, P, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], k);
    __syncthreads();
    for (int c = tid; c < nc; c += WG) contact_rows(S, stride, I, P, rows, jb, c0 + c, I.crow[c0 + c], hinv);
    for (int k = tid; k < nb; k += WG) body_tmp(S, stride, bs + (size_t)k * BW_COUNT, I.bodies[b0 + k], hinv);
    __syncthreads();
    __shared__ int s_cnt[2][WG / 64 + 1];
    __shared__ T s_red[WG / 64];
    // rows: setup, the largest |rhs| (the tolerance's scale), and who can never clamp
    T bmax = T(0);
    for (int i = tid; i < m; i += WG) {
        row_setup<T, false>(rows, jb, bs, i, hinv, P.sor_w);
        const T v = tabs(rows[(size_t)i * RW_COUNT + RW_RHS]);
        if (v > bmax) bmax = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const T v = __shfl_xor(bmax, o, 64); if (v > bmax) bmax = v; }
    if (lane == 0) s_red[wave] = bmax;
    __syncthreads();
    bmax = s_red[0];
    for (int q = 1; q < WG / 64; q++) if (s_red[q] > bmax) bmax = s_red[q];
    const T tol = tol_rel * (T(1) + bmax);

    extern __shared__ __align__(16) unsigned char lcp_lds_raw[];
    // ---- the permutation: unbounded rows first (m <= a few hundred: passes of WG rows, ballot ranks)
    int nu = 0;
    {
        // count first, then place: two passes over the rows, each with per-wave ballots
        int mine_u = 0;
        for (int base = 0; base < m; base += WG) {
            const int i = base + tid;
            const bool u = i < m && rows[(size_t)i * RW_COUNT + RW_LO] == -Limits<T>::inf() && rows[(size_t)i * RW_COUNT + RW_HI] == Limits<T>::inf();
            const unsigned long long bal = __ballot(u);
            if (lane == 0) mine_u += __popcll(bal);
        }
        if (lane == 0) s_cnt[0][wave] = mine_u;
        __syncthreads();
        for (int q = 0; q < WG / 64; q++) nu += s_cnt[0][q];
        __syncthreads();
    }
    const int nbd = m - nu;
    T *M = reinterpret_cast<T *>(lcp_lds_raw);
    T *W = M + (size_t)(m + 1) * (m + 2) / 2;
    T *lam = W + (size_t)(nbd + 1) * (nbd + 2) / 2, *wv = lam + nbd, *lo = wv + nbd, *hi = lo + nbd;
    T *dinv = hi + nbd, *z = dinv + nu;
    int *perm = reinterpret_cast<int *>(lcp_lds_raw + ((lcp_lds_reals(m, nbd) * sizeof(T) + 15) / 16) * 16);
    int *pb1 = perm + m, *pb2 = pb1 + m, *state = pb2 + m, *fidx = state + nbd, *viol = fidx + nbd;
    {
        int ubase = 0, bbase = 0;
        for (int base = 0; base < m; base += WG) {
            const int i = base + tid;
            const bool in = i < m;
            const bool u = in && rows[(size_t)i * RW_COUNT + RW_LO] == -Limits<T>::inf() && rows[(size_t)i * RW_COUNT + RW_HI] == Limits<T>::inf();
            const unsigned long long bu = __ballot(u), bb = __ballot(in && !u);
            if (lane == 0) { s_cnt[0][wave] = __popcll(bu); s_cnt[1][wave] = __popcll(bb); }
            __syncthreads();
            int offu = ubase, offb = bbase;
            for (int q = 0; q < wave; q++) { offu += s_cnt[0][q]; offb += s_cnt[1][q]; }
            const unsigned long long lt = (1ull << lane) - 1ull;
            if (u) perm[offu + __popcll(bu & lt)] = i;
            else if (in) perm[nu + offb + __popcll(bb & lt)] = i;
            for (int q = 0; q < WG / 64; q++) { ubase += s_cnt[0][q]; bbase += s_cnt[1][q]; }
            __syncthreads();
        }
    }
    for (int p = tid; p < m; p += WG) {
        const int i = perm[p];
        pb1[p] = jb[2 * i]; pb2[p] = jb[2 * i + 1];
        if (p >= nu) {
            lo[p - nu] = rows[(size_t)i * RW_COUNT + RW_LO]; hi[p - nu] = rows[(size_t)i * RW_COUNT + RW_HI];
            state[p - nu] = ST_FREE; lam[p - nu] = T(0);
        }
    }
    __syncthreads();
    // ---- A, permuted, packed lower, with the right-hand side as row m
    for (int p = tid; p <= m; p += WG) {
        T *Mp = M + tri(p, 0);
        if (p == m) { for (int q = 0; q < m; q++) Mp[q] = rows[(size_t)perm[q] * RW_COUNT + RW_RHS]; Mp[m] = T(0); continue; }
        const int i = perm[p], i1 = pb1[p], i2 = pb2[p];
        const T *ji = rows + (size_t)i * RW_COUNT + RW_J;
        for (int q = 0; q <= p; q++) {
            const int j1 = pb1[q], j2 = pb2[q];
            T a = T(0);
            if (i1 == j1 || i1 == j2 || (i2 >= 0 && (i2 == j1 || i2 == j2))) {
                const T *pj = rows + (size_t)perm[q] * RW_COUNT + RW_IMJ;
                if (i1 == j1) { for (int e = 0; e < 6; e++) a = fma_(ji[e], pj[e], a); }
                if (j2 >= 0 && i1 == j2) { for (int e = 0; e < 6; e++) a = fma_(ji[e], pj[6 + e], a); }
                if (i2 >= 0 && i2 == j1) { for (int e = 0; e < 6; e++) a = fma_(ji[6 + e], pj[e], a); }
                if (i2 >= 0 && j2 >= 0 && i2 == j2) { for (int e = 0; e < 6; e++) a = fma_(ji[6 + e], pj[6 + e], a); }
            }
            if (q == p) a += rows[(size_t)i * RW_COUNT + RW_AD];
            Mp[q] = a;
        }
    }
    __syncthreads();
    // ---- U eliminated: what is left in rows / columns nu.. is the Schur complement and the reduced right-hand side
    ldlt_steps<T, WG>(M, m, 0, nu, dinv, tol, tid);
    const int max_rounds = 20 * m + 100;
    (void)lds_pivot_rounds<T, WG>(M, m, nu, nbd, W, lam, wv, lo, hi, state, fidx, viol, z, tol, murty_only, max_rounds, tid);
    // ---- lambda_U: L_UU^T x = D^-1 y_U - L_BU^T lambda_B
    for (int k = tid; k < nu; k += WG) {
        T acc = M[tri(m, k)];
        for (int e = 0; e < nbd; e++) acc = fma_(-M[tri(nu + e, k)], lam[e], acc);
        z[k] = acc * dinv[k];
    }
    __syncthreads();
    ldlt_backsub<T, WG>(M, nu, dinv, z, tid);
    // lambda into the rows (free rows clamped to their bounds as the oracle does), residual, forces, integration
    double resid = 0.0;
    for (int p = tid; p < m; p += WG) {
        const int i = perm[p];
        T l;
        if (p < nu) l = z[p];
        else {
            const int q = p - nu;
            l = lam[q];
            const T w = wv[q];
            if (state[q] == ST_FREE) { if (l < lo[q]) l = lo[q]; if (l > hi[q]) l = hi[q]; resid += (double)tabs(w); }
            else resid += (double)(state[q] == ST_LO ? (w < T(0) ? -w : T(0)) : (w > T(0) ? w : T(0)));
        }
        rows[(size_t)i * RW_COUNT + RW_LAM] = l;
    }
    __syncthreads();
    for (int k = tid; k < nb; k += WG) {
        T f[6] = { T(0), T(0), T(0), T(0), T(0), T(0) };
        for (int i = 0; i < m; i++) {
            const int2 bb = *reinterpret_cast<const int2 *>(jb + 2 * (size_t)i);
            if (bb.x != k && bb.y != k) continue;
            const T *ip = rows + (size_t)i * RW_COUNT + RW_IMJ;
            const T l = rows[(size_t)i * RW_COUNT + RW_LAM];
            if (bb.x == k) { for (int q = 0; q < 6; q++) f[q] = fma_(l, ip[q], f[q]); }
            if (bb.y == k) { for (int q = 0; q < 6; q++) f[q] = fma_(l, ip[6 + q], f[q]); }
        }
        T *b = bs + (size_t)k * BW_COUNT;
        for (int q = 0; q < 6; q++) b[BW_FC + q] = f[q];
        finish_body(S, bflags, stride, b, I.bodies[b0 + k], m > 0, h);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) resid += __shfl_xor(resid, o, 64);
    if (lane == 0 && resid != 0.0) atomicAdd(&diag->residual, resid);
    if (tid == 0) atomicAdd(&diag->contacts, (unsigned long long)nc);
}

// =========================================================================================================== the volatile rows' problem
// Second level of the same idea.  After a pivoting round on the bounded rows B most of them are settled -- free with lambda well
// inside its bounds, or clamped with w well on its side -- and a few are not (the violators, and rows close to changing sides).
// The settled FREE rows Fs are eliminated from S like U was from A (gather [Fs | V], panels over Fs: the same kernels), which
// leaves the LCP in the volatile rows V alone, a hundred or two: small enough for ONE workgroup to pivot to the end in LDS
// (lds_pivot_rounds), however many rounds -- Murty's single flips included -- that takes.  lambda_Fs follows by
// back-substitution, and a product with S checks EVERY bounded row: if a settled row turns out to violate, the host flips and
// goes round again.
template <class T, int WG>
__global__ __launch_bounds__(WG) void lcp_reduced_lds(const T *__restrict__ Sv, int ldv, const T *__restrict__ bv, int nv, const T *__restrict__ lo_g,
                                                      const T *__restrict__ hi_g, int *__restrict__ state_g, T *__restrict__ lam_g,
                                                      const T *__restrict__ tolp, int murty_only, int *__restrict__ rounds_out)
{
    extern __shared__ __align__(16) unsigned char lcp_red_raw[];
    const int tid = threadIdx.x;
    const size_t tsz = (size_t)(nv + 1) * (nv + 2) / 2;
    T *M = reinterpret_cast<T *>(lcp_red_raw), *W = M + tsz;
    T *lam = W + tsz, *wv = lam + nv, *lo = wv + nv, *hi = lo + nv, *rd = hi + nv;
    int *state = reinterpret_cast<int *>(lcp_red_raw + (((2 * tsz + (size_t)6 * nv + 8) * sizeof(T) + 15) / 16) * 16);
    int *fidx = state + nv, *viol = fidx + nv;
    for (int p = tid; p <= nv; p += WG) {
        T *Mp = M + tri(p, 0);
        if (p == nv) { for (int q = 0; q < nv; q++) Mp[q] = bv[q]; Mp[nv] = T(0); }
        else for (int q = 0; q <= p; q++) Mp[q] = Sv[(size_t)q * ldv + p];
    }
    for (int q = tid; q < nv; q += WG) { lo[q] = lo_g[q]; hi[q] = hi_g[q]; state[q] = state_g[q]; lam[q] = T(0); }
    __syncthreads();
    const int rounds = lds_pivot_rounds<T, WG>(M, nv, 0, nv, W, lam, wv, lo, hi, state, fidx, viol, rd, tolp[0], murty_only, 20 * nv + 100, tid);
    for (int q = tid; q < nv; q += WG) { lam_g[q] = lam[q]; state_g[q] = state[q]; }
    if (tid == 0) rounds_out[0] = rounds;
}
template <class T> size_t reduced_lds_bytes(int nv)
{
    const size_t tsz = (size_t)(nv + 1) * (nv + 2) / 2;
    return (((2 * tsz + (size_t)6 * nv + 8) * sizeof(T) + 15) / 16) * 16 + ((size_t)3 * nv + 8) * sizeof(int);
}
template <class T> int reduced_lds_cap()
{
    static const int cap = [] { int n = 16; while (reduced_lds_bytes<T>(n + 8) <= (size_t)144 * 1024 && n + 9 <= LDLT_MAXN<T>) n += 8; return n; }();
    return cap;
}
// the volatile rows' data out of the level-2 matrix: b'' = its right-hand-side row behind Fs's columns; bounds and states by row
// (vidx[i] = the bounded row at position i of V, -1 = padding)
template <class T>
__global__ __launch_bounds__(256) void lcp_l2_setup(const T *__restrict__ Mw, int ldw, int fsP, int rhs_row, const int *__restrict__ vidx, int nvP,
                                                    const T *__restrict__ lo, const T *__restrict__ hi, const int *__restrict__ state,
                                                    T *__restrict__ b2, T *__restrict__ lo2, T *__restrict__ hi2, int *__restrict__ state2, T *__restrict__ lam2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nvP) return;
    const int q = vidx[i];
    b2[i] = Mw[(size_t)(fsP + i) * ldw + rhs_row];
    lo2[i] = q >= 0 ? lo[q] : -Limits<T>::inf();
    hi2[i] = q >= 0 ? hi[q] : Limits<T>::inf();
    state2[i] = q >= 0 ? state[q] : (int)ST_FREE;
    lam2[i] = T(0);
}
template <class T>
__global__ __launch_bounds__(256) void lcp_l2_scatter(const int *__restrict__ vidx, int nv, const T *__restrict__ lam2, const int *__restrict__ state2,
                                                      T *__restrict__ lamB, int *__restrict__ state)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nv) return;
    const int q = vidx[i];
    if (q < 0) return;
    lamB[q] = lam2[i];
    state[q] = state2[i];
}

// =========================================================================================================== host side
struct LcpGrid {
    dmxBatch::DevBuf A, Ldiag, Sd, Mw, Mdiag, vec, ints;
    void *pin = nullptr; size_t pin_bytes = 0;
    void *pin_real = nullptr; size_t pin_real_bytes = 0;
    dmxBatch::DevBuf Sv, vec2, ints2;
    int l2_at = 0;               // DMX_LCP_L2_AT (0: 16 rows in f32, 32 in f64)
    int level2 = -1;             // DMX_LCP_LEVEL2: 1 = the volatile rows' problem is pivoted in LDS between checks of all rows, 0 = every pivoting
                                 // round refactors the whole free block; -1: by precision (see lcp_grid_solve)
    std::vector<int> rank_f, rank_c, in_v, state0;
    std::unordered_map<uint64_t, uint8_t> warm_prev, warm_next;
    std::vector<int> perm, state;
    int64_t stats[8] = { 0 };
    int64_t l2_solves = 0;       // volatile-row problems solved in LDS (level 2)
    double flops = 0;            // algorithmic: the partial factorisation (nu^3/3 + nu^2 nb + nu nb^2) and nf^3/3 per pivoting round
    bool warm = true, murty_only = false;
    int w_mode = -1;             // DMX_LCP_W: 0 = w from the Schur complement (one product per round), 1 = from the rows through lambda_U; -1: by precision
    double tol_rel = 0;          // DMX_LCP_TOL: the complementarity tolerance relative to 1 + max |rhs| (0: the oracle's, 1e-5 f32 / 1e-11 f64)
};

LcpGrid *grid_of(dmxBatch *b)
{
    if (!b->lcp_grid) {
        LcpGrid *g = new LcpGrid;
        const char *e = getenv("DMX_LCP_WARM");
        g->warm = !(e && atoi(e) == 0);
        e = getenv("DMX_LCP_MURTY");
        g->murty_only = e && atoi(e) != 0;
        e = getenv("DMX_LCP_LEVEL2");
        if (e) g->level2 = atoi(e);
        e = getenv("DMX_LCP_L2_AT");
        if (e) g->l2_at = atoi(e);
        e = getenv("DMX_LCP_W");
        if (e) g->w_mode = atoi(e) != 0 ? 1 : 0;
        e = getenv("DMX_LCP_TOL");
        if (e) g->tol_rel = atof(e);
        b->lcp_grid = g;
    }
    return (LcpGrid *)b->lcp_grid;
}

// the bounded rows outside `skip`, by how close they are to changing sides: free rows by lambda - lo (a normal row's lo is 0; a
// bounded friction row's bounds are not known on the host: it counts as closest), clamped rows by |w|
template <class T>
void rank_margins(int nbd, const std::vector<int> &state, const std::vector<int> &skip, const T *h_lam, const T *h_w, const LcpIslandRows &R,
                  const std::vector<int> &perm, int nuP, std::vector<int> &rf, std::vector<int> &rc)
{
    rf.clear(); rc.clear();
    for (int q = 0; q < nbd; q++) {
        if (skip[(size_t)q]) continue;
        if (state[(size_t)q] == ST_FREE) rf.push_back(q); else rc.push_back(q);
    }
    auto margin_f = [&](int q) { return (R.key[(size_t)perm[(size_t)(nuP + q)]] & 3u) != 0u ? (T)0 : h_lam[q]; };
    std::sort(rf.begin(), rf.end(), [&](int a, int c) { const T ma = margin_f(a), mc = margin_f(c); return ma < mc || (ma == mc && a < c); });
    std::sort(rc.begin(), rc.end(), [&](int a, int c) { const T wa = h_w[a] < 0 ? -h_w[a] : h_w[a], wc = h_w[c] < 0 ? -h_w[c] : h_w[c];
                                                        return wa < wc || (wa == wc && a < c); });
}

template <class T> size_t panel_lds() { return (size_t)(2 * NB * NB + NB) * sizeof(T); }

// Cholesky of the first `np` panels of an augmented matrix of `nt` tiles (+ the right-hand side's tile row): per panel the
// factor-and-solve launch over the tile rows below it and the matrix-core update of everything to its right
template <class T>
hipError_t factor_panels(T *A, int ld, int nt, int np, T *Ldiag, const T *tol, hipStream_t st)
{
    const size_t lds = panel_lds<T>();
    for (int k = 0; k < np; k++) {
        hipLaunchKernelGGL((lcp_panel<T>), dim3((unsigned)(nt - k)), dim3(64 * PANEL_WAVES), lds, st, A, ld, k, Ldiag, tol);
        if (nt - k - 1 > 0)
            hipLaunchKernelGGL((lcp_syrk<T>), dim3((unsigned)(nt - k), (unsigned)(nt - k - 1)), dim3(256), 0, st, A, ld, k);
    }
    return hipGetLastError();
}

template <class T> size_t backsolve_lds(int nt) { return ((size_t)nt * NB + 2 * NB * (NB + 1) + NB) * sizeof(T); }

}  // namespace

void lcp_grid_begin_tick(dmxBatch *b)
{
    if (!b->lcp_grid) return;
    LcpGrid *g = (LcpGrid *)b->lcp_grid;
    g->warm_next.clear();
}
void lcp_grid_end_tick(dmxBatch *b)
{
    if (!b->lcp_grid) return;
    LcpGrid *g = (LcpGrid *)b->lcp_grid;
    g->warm_prev.swap(g->warm_next);
    g->warm_next.clear();
}
void lcp_grid_free(dmxBatch *b)
{
    if (!b->lcp_grid) return;
    LcpGrid *g = (LcpGrid *)b->lcp_grid;
    if (getenv("DMX_LCP_REPORT"))
        fprintf(stderr, "libode_mi355 lcp grid: solves=%lld rounds=%lld max_rounds=%lld last_m=%lld last_nu=%lld last_nbd=%lld single=%lld fallback=%lld gflop=%.4f level2=%lld\n",
                (long long)g->stats[0], (long long)g->stats[1], (long long)g->stats[2], (long long)g->stats[3], (long long)g->stats[4],
                (long long)g->stats[5], (long long)g->stats[6], (long long)g->stats[7], g->flops * 1e-9, (long long)g->l2_solves);
    for (dmxBatch::DevBuf *d : { &g->A, &g->Ldiag, &g->Sd, &g->Mw, &g->Mdiag, &g->vec, &g->ints, &g->Sv, &g->vec2, &g->ints2 })
        if (d->p) (void)hipFree(d->p);
    if (g->pin) (void)hipHostFree(g->pin);
    if (g->pin_real) (void)hipHostFree(g->pin_real);
    delete g;
    b->lcp_grid = nullptr;
}

template <class T>
int lcp_grid_solve(dmxBatch *b, const IslandSet<T> &I, const StepParams<T> &P, const LcpIslandRows &R)
{
    LcpGrid *g = grid_of(b);
    hipStream_t st = b->stream;
    const int m = R.m, isl = R.isl;
    // ---- the permutation: unbounded rows, padding, bounded rows, padding
    int nu = 0;
    for (int i = 0; i < m; i++) nu += R.unbounded[(size_t)i] ? 1 : 0;
    const int nbd = m - nu;
    const int nuT = (nu + NB - 1) / NB, nbT = (nbd + NB - 1) / NB;
    const int nuP = nuT * NB, nbdP = nbT * NB, mP = nuP + nbdP, nt = nuT + nbT;
    const int ld = mP + NB, lds = nbdP > 0 ? nbdP : NB, ldw = nbdP + 3 * NB;      // (the level-2 matrix pads Fs and V separately)
    std::vector<int> &perm = g->perm;
    perm.assign((size_t)mP, -1);
    {
        int au = 0, ab = nuP;
        for (int i = 0; i < m; i++) { if (R.unbounded[(size_t)i]) perm[(size_t)au++] = i; else perm[(size_t)ab++] = i; }
    }
    // ---- buffers
    int rc;
    if ((rc = dmx_ensure_dev(g->A, (size_t)mP * ld * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(g->Ldiag, (size_t)nt * NB * NB * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(g->Sd, (size_t)lds * lds * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(g->Mw, (size_t)(nbdP + 2 * NB) * ldw * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(g->Mdiag, (size_t)(nbT + 2) * NB * NB * sizeof(T))) != DMX_OK) return rc;
    const int vcap = reduced_lds_cap<T>(), vcapP = ((vcap + NB - 1) / NB) * NB;
    if ((rc = dmx_ensure_dev(g->Sv, (size_t)vcapP * vcapP * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(g->vec2, (size_t)4 * (vcapP + NB) * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(g->ints2, ((size_t)nbdP + 3 * NB + 2 * (vcapP + NB) + 16) * sizeof(int))) != DMX_OK) return rc;
    {
        const size_t want = ((size_t)2 * (nbdP + NB) * sizeof(T) + ((size_t)nbdP + 3 * NB + vcapP + 64) * sizeof(int)) + 256;
        if (g->pin_real_bytes < want) {
            if (g->pin_real) HIP_TRY(hipHostFree(g->pin_real));
            g->pin_real = nullptr; g->pin_real_bytes = 0;
            HIP_TRY(hipHostMalloc(&g->pin_real, want * 2));
            g->pin_real_bytes = want * 2;
        }
    }
    // vectors: tol[4] bprime lo hi lamB wB rr (nbdP each) z xU (nuP each)
    const size_t nv = 4 + (size_t)6 * (nbdP + NB) + (size_t)2 * (nuP + NB);
    if ((rc = dmx_ensure_dev(g->vec, nv * sizeof(T))) != DMX_OK) return rc;
    // ints: perm[mP] state[nbdP] fidx[nbdP] viol[nbdP] boff[nb + 1] bodyrows[2 m]
    const size_t n_csr = R.boff.size() + R.bodyrows.size();
    const size_t ni = (size_t)mP + (size_t)3 * (nbdP + NB) + n_csr + 8;
    if ((rc = dmx_ensure_dev(g->ints, ni * sizeof(int))) != DMX_OK) return rc;
    if (g->pin_bytes < ni * sizeof(int)) {
        if (g->pin) HIP_TRY(hipHostFree(g->pin));
        g->pin = nullptr; g->pin_bytes = 0;
        const size_t want = ni * sizeof(int) * 2 + 4096;
        HIP_TRY(hipHostMalloc(&g->pin, want));
        g->pin_bytes = want;
    }
    T *A = (T *)g->A.p, *Ldiag = (T *)g->Ldiag.p, *Sd = (T *)g->Sd.p, *Mw = (T *)g->Mw.p, *Mdiag = (T *)g->Mdiag.p;
    T *tol = (T *)g->vec.p, *bprime = tol + 4, *lo = bprime + (nbdP + NB), *hi = lo + (nbdP + NB), *lamB = hi + (nbdP + NB),
      *wB = lamB + (nbdP + NB), *rr = wB + (nbdP + NB), *z = rr + (nbdP + NB), *xU = z + (nuP + NB);
    int *d_perm = (int *)g->ints.p, *d_state = d_perm + mP, *d_fidx = d_state + (nbdP + NB), *d_viol = d_fidx + (nbdP + NB);
    int *h_perm = (int *)g->pin, *h_state = h_perm + mP, *h_fidx = h_state + (nbdP + NB), *h_viol = h_fidx + (nbdP + NB);
    int *d_boff = d_viol + (nbdP + NB), *d_bodyrows = d_boff + R.boff.size();
    int *h_boff = h_viol + (nbdP + NB), *h_bodyrows = h_boff + R.boff.size();
    const bool sparse_w = g->w_mode > 0;
    T *Sv = (T *)g->Sv.p, *b2 = (T *)g->vec2.p, *lo2 = b2 + (vcapP + NB), *hi2 = lo2 + (vcapP + NB), *lam2 = hi2 + (vcapP + NB);
    int *d_fidx2 = (int *)g->ints2.p, *d_state2 = d_fidx2 + (nbdP + 3 * NB), *d_rounds2 = d_state2 + (vcapP + NB);
    T *h_lam = (T *)g->pin_real, *h_w = h_lam + (nbdP + NB);
    int *h_fidx2 = (int *)(h_w + (nbdP + NB)), *h_state2 = h_fidx2 + (nbdP + 3 * NB), *h_rounds2 = h_state2 + vcapP + 8;

    // (the pinned staging is reused by the next island / tick: every copy below is followed by a synchronisation before the
    //  host writes it again -- the rounds' read-back)
    memcpy(h_perm, perm.data(), (size_t)mP * sizeof(int));
    HIP_TRY(hipMemcpyAsync(d_perm, h_perm, (size_t)mP * sizeof(int), hipMemcpyHostToDevice, st));
    memcpy(h_boff, R.boff.data(), R.boff.size() * sizeof(int));
    memcpy(h_bodyrows, R.bodyrows.data(), R.bodyrows.size() * sizeof(int));
    HIP_TRY(hipMemcpyAsync(d_boff, h_boff, n_csr * sizeof(int), hipMemcpyHostToDevice, st));

    // rows of this island inside the flat arrays: row_off is a device array, but by construction row_off[isl] = 3 * con_off[isl],
    // which the caller passes as R.row_base
    T *rows = I.rows + (size_t)R.row_base * RW_COUNT;
    const int *jb = I.rowjb + 2 * (size_t)R.row_base;

    hipLaunchKernelGGL((lcp_prepare<T>), dim3(1), dim3(512), 0, st, (T *)b->slab, b->bflags, b->stride, I, P, isl, tol,
                       (T)(g->tol_rel > 0 ? g->tol_rel : (sizeof(T) == 4 ? 1e-5 : 1e-11)));
    hipLaunchKernelGGL((lcp_assemble<T>), dim3((unsigned)(nt + 1), (unsigned)nt), dim3(256), 0, st, rows, jb, d_perm, nt, A, ld);
    {
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&lcp_panel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)panel_lds<T>());
        if (ea != hipSuccess) HIP_TRY(ea);
    }
    HIP_TRY(factor_panels<T>(A, ld, nt, nuT, Ldiag, tol, st));
    g->flops += (double)nu * nu * nu / 3.0 + (double)nu * nu * nbd + (double)nu * nbd * nbd;

    // ---- the reduced problem in B's rows
    std::vector<int> &state = g->state;
    state.assign((size_t)nbd, ST_FREE);
    int rounds = 0, single_rounds = 0, l2_solves = 0;
    if (nbd > 0) {
        hipLaunchKernelGGL((lcp_extract<T>), dim3((unsigned)nbT, (unsigned)nbT), dim3(256), 0, st, A, ld, nuP, Sd, lds);
        hipLaunchKernelGGL((lcp_bvec<T>), dim3((unsigned)((nbdP + 255) / 256)), dim3(256), 0, st, A, ld, nuP, mP, d_perm, rows, nbdP, bprime, lo, hi);
        // where the rows' active set starts: what the same contact's row ended the previous tick with; and which rows are
        // expected to move (they were close to changing sides then, or are new): the first pass's volatile set
        bool any_bounded_friction = false;
        std::vector<int> &in_v = g->in_v, &rf = g->rank_f, &rc2 = g->rank_c, &state0 = g->state0;
        in_v.assign((size_t)nbd, 0);
        // (f64, ODE's cfm = 1e-10: the bounded rows' problem is close to degenerate and plain block pivoting cycles -- 40 to 70 rounds in
        //  the pen, the oracle's own count -- until Murty's single flips end it; pivoting the volatile rows in LDS ends that.  f32,
        //  cfm = 1e-5: five plain rounds or so.)
        // DMX_LCP_LEVEL2 = 2 (f32's default): plain rounds first -- each cuts the violators severalfold -- and the volatile rows' solve
        // once no more than DMX_LCP_L2_AT (32) rows violate: the last two or three plain rounds, which chase a handful of rows through
        // a full refactorisation each, become one pass.
        // (measured, 400 / 512 bodies in the pen, ms per tick -- f32: plain rounds 2.46 / 3.12, this 2.36 / 3.05 at 16 rows, 3.26 / 3.45 at 64;
        //  f64: the volatile rows' solve from the first pass on 3.84 / 4.06, this 3.41 / 3.61 at 32 rows, plain rounds 5.5 / 5.9)
        const int l2_mode = g->level2 >= 0 ? g->level2 : 2;
        const int l2_at = g->l2_at > 0 ? g->l2_at : (sizeof(T) == 8 ? 32 : 16);
        const bool use_l2 = l2_mode != 0;
        bool classical = !use_l2 || g->murty_only || l2_mode == 2;
        const bool hybrid = use_l2 && !g->murty_only && l2_mode == 2;
        int nv_pred = 0;
        for (int q = 0; q < nbd; q++) {
            const int i = perm[(size_t)(nuP + q)];
            const uint64_t key = R.key[(size_t)i];
            if ((key & 3u) != 0u) any_bounded_friction = true;
            bool known = false;
            if (g->warm && key != 0) {
                auto it = g->warm_prev.find(key);
                if (it != g->warm_prev.end()) { state[(size_t)q] = it->second & 3; known = true; if (it->second & 4) { in_v[(size_t)q] = 1; nv_pred++; } }
            }
            if (!known && g->warm && !g->warm_prev.empty()) { in_v[(size_t)q] = 1; nv_pred++; }      // a new contact
        }
        if (classical || nv_pred > vcap) { in_v.assign((size_t)nbd, 0); nv_pred = 0; }
        state0 = state;
        int best = m + 1, patience = g->murty_only ? 0 : 3, passes = 0, l2_passes = 0;
        bool subset_mode = false;
        const int max_rounds = 20 * m + 100;
        {
            const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&lcp_backsolve<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      (int)backsolve_lds<T>((nuT > nbT ? nuT : nbT) + 1));
            if (ea != hipSuccess) HIP_TRY(ea);
            const hipError_t eb = hipFuncSetAttribute(reinterpret_cast<const void *>(&lcp_reduced_lds<T, 256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      (int)reduced_lds_bytes<T>(vcap));
            if (eb != hipSuccess) HIP_TRY(eb);
        }
        // One pass: the settled free rows Fs (free and not in V) are eliminated from S (gather [Fs | V], panels over Fs), V's own LCP is
        // pivoted to the end by one workgroup in LDS, lambda_Fs follows by back-substitution, and EVERY bounded row is checked.  With V
        // empty a pass is a plain pivoting round.  Violators join V (with the rows closest to changing sides) and the pass is
        // repeated; should V outgrow a workgroup's LDS, the violators are flipped on the host instead (the classical block-pivoting
        // round, with Murty's single flip once the violation count has stalled three times) and V starts empty again.
        for (;; rounds++) {
            int nfs = 0, nv2 = 0;
            for (int q = 0; q < nbd; q++) {
                h_state[q] = state[(size_t)q];
                if (!in_v[(size_t)q] && state[(size_t)q] == ST_FREE) h_fidx2[nfs++] = q;
            }
            const int fsT = (nfs + NB - 1) / NB, fsP = fsT * NB;
            for (int a = nfs; a < fsP; a++) h_fidx2[a] = -1;
            for (int q = 0; q < nbd; q++) if (in_v[(size_t)q]) h_fidx2[fsP + nv2++] = q;
            const int vT = (nv2 + NB - 1) / NB, vP = vT * NB, nt2 = fsT + vT;
            for (int a = nv2; a < vP; a++) h_fidx2[fsP + a] = -1;
            HIP_TRY(hipMemcpyAsync(d_state, h_state, (size_t)nbd * sizeof(int), hipMemcpyHostToDevice, st));
            if (nt2 > 0) HIP_TRY(hipMemcpyAsync(d_fidx2, h_fidx2, (size_t)(fsP + vP) * sizeof(int), hipMemcpyHostToDevice, st));
            // right-hand side: b' less the settled clamped rows' part (non-zero only for bounded friction rows at a bound)
            const T *rhs2 = bprime;
            if (any_bounded_friction) {
                bool nz = false;
                for (int q = 0; q < nbd; q++)
                    if (!in_v[(size_t)q] && state[(size_t)q] != ST_FREE && (R.key[(size_t)perm[(size_t)(nuP + q)]] & 3u) != 0u) nz = true;
                if (nz) {
                    // (V's rows count as free here: their part is the LDS solve's own business)
                    hipLaunchKernelGGL((lcp_clamped_except<T>), dim3((unsigned)((nbd + 255) / 256)), dim3(256), 0, st, d_state, lo, hi, nbd, d_fidx2 + fsP, nv2, lamB);
                    hipLaunchKernelGGL((lcp_gemv<T, false>), dim3((unsigned)nbT), dim3(256), 0, st, Sd, lds, nbd, lamB, bprime, rr,
                                       (const int *)nullptr, (const T *)nullptr, (const T *)nullptr, (const T *)nullptr, (int *)nullptr);
                    rhs2 = rr;
                }
            }
            if (nt2 > 0) {
                hipLaunchKernelGGL((lcp_gather<T>), dim3((unsigned)(nt2 + 1), (unsigned)nt2), dim3(256), 0, st, Sd, lds, d_fidx2, nt2, rhs2, Mw, ldw);
                HIP_TRY(factor_panels<T>(Mw, ldw, nt2, fsT, Mdiag, tol, st));
                g->flops += (double)nfs * nfs * nfs / 3.0 + (double)nfs * nfs * nv2 + (double)nfs * nv2 * nv2;
            }
            if (nv2 > 0) {
                hipLaunchKernelGGL((lcp_extract<T>), dim3((unsigned)vT, (unsigned)vT), dim3(256), 0, st, Mw, ldw, fsP, Sv, vP);
                hipLaunchKernelGGL((lcp_l2_setup<T>), dim3((unsigned)((vP + 255) / 256)), dim3(256), 0, st, Mw, ldw, fsP, nt2 * NB, d_fidx2 + fsP, vP, lo, hi,
                                   d_state, b2, lo2, hi2, d_state2, lam2);
                hipLaunchKernelGGL((lcp_reduced_lds<T, 256>), dim3(1), dim3(256), reduced_lds_bytes<T>(nv2), st, Sv, vP, b2, nv2, lo2, hi2, d_state2, lam2, tol, 0,
                                   d_rounds2);
                hipLaunchKernelGGL((lcp_l2_scatter<T>), dim3((unsigned)((nv2 + 255) / 256)), dim3(256), 0, st, d_fidx2 + fsP, nv2, lam2, d_state2, lamB, d_state);
                l2_solves++;
            }
            // lambda_B: clamped rows at their bounds, V from the LDS solve, Fs by back-substitution
            hipLaunchKernelGGL((lcp_clamped<T>), dim3((unsigned)((nbd + 255) / 256)), dim3(256), 0, st, d_state, lo, hi, nbd, lamB);
            if (nv2 > 0)
                hipLaunchKernelGGL((lcp_l2_scatter<T>), dim3((unsigned)((nv2 + 255) / 256)), dim3(256), 0, st, d_fidx2 + fsP, nv2, lam2, d_state2, lamB, d_state);
            if (fsT > 0) {
                // (rr is free again: the gather has consumed it)
                hipLaunchKernelGGL((lcp_zvec<T>), dim3((unsigned)fsP), dim3(64), 0, st, Mw, ldw, fsP, nt2 * NB, vP, lam2, rr);
                hipLaunchKernelGGL((lcp_backsolve<T>), dim3(1), dim3(1024), backsolve_lds<T>(fsT), st, Mw, ldw, Mdiag, fsT, rr, (size_t)1, d_fidx2, lamB);
            }
            if (sparse_w) {
                if (nuT > 0) {
                    hipLaunchKernelGGL((lcp_zvec<T>), dim3((unsigned)nuP), dim3(64), 0, st, A, ld, nuP, mP, nbd, lamB, z);
                    hipLaunchKernelGGL((lcp_backsolve<T>), dim3(1), dim3(1024), backsolve_lds<T>(nuT), st, A, ld, Ldiag, nuT, z, (size_t)1,
                                       (const int *)nullptr, xU);
                }
                hipLaunchKernelGGL((lcp_forces<T, false>), dim3(1), dim3(1024), 0, st, (T *)b->slab, b->bflags, b->stride, I, P, isl, d_perm, nuP, mP,
                                   nbd, xU, lamB, wB, d_state, d_boff, d_bodyrows, tol, d_viol, b->diag_isl);
            } else
                hipLaunchKernelGGL((lcp_gemv<T, true>), dim3((unsigned)nbT), dim3(256), 0, st, Sd, lds, nbd, lamB, bprime, wB, d_state, lo, hi, tol, d_viol);
            HIP_TRY(hipMemcpyAsync(h_viol, d_viol, (size_t)nbd * sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_lam, lamB, (size_t)nbd * sizeof(T), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_w, wB, (size_t)nbd * sizeof(T), hipMemcpyDeviceToHost, st));
            if (nv2 > 0) {
                HIP_TRY(hipMemcpyAsync(h_state, d_state, (size_t)nbd * sizeof(int), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(h_rounds2, d_rounds2, sizeof(int), hipMemcpyDeviceToHost, st));
            }
            HIP_TRY(hipStreamSynchronize(st));
            passes++;
            if (nv2 > 0) l2_passes++;
            if (nv2 > 0) {
                for (int q = 0; q < nbd; q++) state[(size_t)q] = h_state[q];
                g->stats[1] += h_rounds2[0];
            }
            int nviol = 0, top = -1;
            for (int q = 0; q < nbd; q++) if (h_viol[q]) { nviol++; top = q; }
            static const int trace = [] { const char *e = getenv("DMX_LCP_TRACE"); return e ? atoi(e) : 0; }();
            if (trace > 0 && rounds >= trace)
                fprintf(stderr, "lcp trace: solve %lld round %d pass %d nviol %d nv %d nfs %d %s\n", (long long)g->stats[0], rounds, passes, nviol, nv2, nfs,
                        classical ? "classical" : "level2");
            if (nviol == 0 || rounds >= max_rounds) break;
            // ... or once block pivoting has stalled (the violation count has failed to shrink three times: what would follow is Murty's
            // single flips, a full refactorisation per flipped row -- 500 rounds in one f64 tick of a 1 500-tick run, 0.2 s) and the
            // violators with their neighbourhood fit a workgroup's LDS: there a flip costs microseconds.
            const bool stalled = !g->murty_only && nviol >= best && patience == 0;
            // (More violators than the LDS level holds -- 176 of 679 rows in the worst tick of that run: the level takes them a
            //  workgroup-full at a time, the highest rows first as Murty's rule takes its one; each pass settles its share exactly
            //  against the rest as it stands.  No proof of termination comes with that: forty passes, then the single flips.)
            if (hybrid && classical && (nviol <= l2_at || stalled) && l2_passes < 40) {
                classical = false; in_v.assign((size_t)nbd, 0);
                subset_mode = stalled && 2 * nviol > vcap;
            }
            if (!classical && l2_passes < 40) {
                // the violators join V, and so do the rows closest to changing sides, while a workgroup's LDS has room
                int nvv = 0;
                for (int q = 0; q < nbd; q++) { if (h_viol[q]) in_v[(size_t)q] = 1; nvv += in_v[(size_t)q]; }
                if (subset_mode && nvv > 3 * vcap / 4) {
                    int seen = 0;
                    for (int q = nbd - 1; q >= 0; q--)
                        if (in_v[(size_t)q]) { if (seen >= 3 * vcap / 4 || !h_viol[q]) in_v[(size_t)q] = 0; else seen++; }
                    nvv = seen;
                }
                if (nvv <= vcap) {
                    rank_margins<T>(nbd, state, in_v, h_lam, h_w, R, perm, nuP, rf, rc2);
                    const int want = std::min(std::min(vcap, nbd), std::max(nvv + 2 * nviol + 16, 64));
                    for (size_t a = 0, c = 0; nvv < want && (a < rf.size() || c < rc2.size());) {
                        if (a < rf.size()) { in_v[(size_t)rf[a++]] = 1; nvv++; }
                        if (nvv < want && c < rc2.size()) { in_v[(size_t)rc2[c++]] = 1; nvv++; }
                    }
                    continue;
                }
                in_v.assign((size_t)nbd, 0);           // too many for one workgroup: a classical round on all of them, V starts again
            }
            bool all = true;
            if (nviol < best) { best = nviol; if (!g->murty_only) patience = 3; }
            else if (patience > 0) patience--;
            else all = false;
            if (g->murty_only) all = false;
            if (!all) single_rounds++;
            for (int q = 0; q < nbd; q++) {
                if (!h_viol[q] || (!all && q != top)) continue;
                state[(size_t)q] = h_viol[q] == 1 ? ST_LO : h_viol[q] == 2 ? ST_HI : ST_FREE;
            }
        }
        {
            static const int trace = [] { const char *e = getenv("DMX_LCP_TRACE"); return e ? atoi(e) : 0; }();
            if (trace < 0) fprintf(stderr, "lcp solve: %lld m %d nbd %d passes %d l2_passes %d single %d\n", (long long)g->stats[0], m, nbd, passes, l2_passes, single_rounds);
        }
        // remember the active set, and who is likely to move next tick: the rows that changed sides in this solve and the ones
        // closest to doing so
        {
            std::vector<int> &vol = g->in_v;
            int changed = 0;
            for (int q = 0; q < nbd; q++) { vol[(size_t)q] = state[(size_t)q] != state0[(size_t)q] ? 1 : 0; changed += vol[(size_t)q]; }
            if (!classical && !hybrid) {
                rank_margins<T>(nbd, state, vol, h_lam, h_w, R, perm, nuP, rf, rc2);
                int nvv = changed;
                const int want = std::min(std::min(3 * vcap / 4, nbd), std::max(3 * changed + 16, 48));
                for (size_t a = 0, c = 0; nvv < want && (a < rf.size() || c < rc2.size());) {
                    if (a < rf.size()) { vol[(size_t)rf[a++]] = 1; nvv++; }
                    if (nvv < want && c < rc2.size()) { vol[(size_t)rc2[c++]] = 1; nvv++; }
                }
            }
            for (int q = 0; q < nbd; q++) {
                const uint64_t key = R.key[(size_t)perm[(size_t)(nuP + q)]];
                if (key != 0) g->warm_next[key] = (uint8_t)(state[(size_t)q] | (vol[(size_t)q] ? 4 : 0));
            }
        }
    } else {
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&lcp_backsolve<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)backsolve_lds<T>(nuT));
        if (ea != hipSuccess) HIP_TRY(ea);
    }
    // ---- lambda_U, forces, integration
    if (nuT > 0 && !(sparse_w && nbd > 0)) {        // (the last round of the accurate form has left lambda_U behind)
        hipLaunchKernelGGL((lcp_zvec<T>), dim3((unsigned)nuP), dim3(64), 0, st, A, ld, nuP, mP, nbd, lamB, z);
        hipLaunchKernelGGL((lcp_backsolve<T>), dim3(1), dim3(1024), backsolve_lds<T>(nuT), st, A, ld, Ldiag, nuT, z, (size_t)1,
                           (const int *)nullptr, xU);
    }
    hipLaunchKernelGGL((lcp_forces<T, true>), dim3(1), dim3(1024), 0, st, (T *)b->slab, b->bflags, b->stride, I, P, isl, d_perm, nuP, mP, nbd, xU,
                       lamB, wB, d_state, d_boff, d_bodyrows, tol, d_viol, b->diag_isl);
    HIP_TRY(hipGetLastError());
    g->stats[0] += 1; g->stats[1] += rounds + 1; if (rounds + 1 > g->stats[2]) g->stats[2] = rounds + 1;
    g->stats[3] = m; g->stats[4] = nu; g->stats[5] = nbd; g->stats[6] += single_rounds; g->l2_solves += l2_solves;
    return DMX_OK;
}

template int lcp_grid_solve<float>(dmxBatch *, const IslandSet<float> &, const StepParams<float> &, const LcpIslandRows &);
template int lcp_grid_solve<double>(dmxBatch *, const IslandSet<double> &, const StepParams<double> &, const LcpIslandRows &);

// can island (m rows, nbd of them bounded) be solved by one workgroup in LDS?
bool lcp_lds_fits(int real_bytes, int m, int nbd)
{
    static const int lim = [] { const char *e = getenv("DMX_LCP_LDS_BYTES"); return e ? atoi(e) : 150 * 1024; }();
    if (m + 1 > (real_bytes == 4 ? LDLT_MAXN<float> : LDLT_MAXN<double>)) return false;
    const size_t need = real_bytes == 4 ? lcp_lds_bytes<float>(m, nbd) : lcp_lds_bytes<double>(m, nbd);
    return need <= (size_t)lim;
}
template <class T>
hipError_t launch_lcp_lds(T *S, const uint8_t *bflags, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P, StepDiag *diag,
                          size_t lds_bytes, hipStream_t st)
{
    if (I.n_big <= 0) return hipSuccess;
    static const int murty = [] { const char *e = getenv("DMX_LCP_MURTY"); return e && atoi(e) != 0 ? 1 : 0; }();
    static const double tol_env = [] { const char *e = getenv("DMX_LCP_TOL"); return e ? atof(e) : 0.0; }();
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&lcp_island_lds<T, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((lcp_island_lds<T, 256>), dim3((unsigned)I.n_big), dim3(256), lds_bytes, st, S, bflags, stride, I, P, diag, murty,
                       (T)(tol_env > 0 ? tol_env : (sizeof(T) == 4 ? 1e-5 : 1e-11)));
    return hipGetLastError();
}
template hipError_t launch_lcp_lds<float>(float *, const uint8_t *, int64_t, const IslandSet<float> &, const StepParams<float> &, StepDiag *, size_t, hipStream_t);
template hipError_t launch_lcp_lds<double>(double *, const uint8_t *, int64_t, const IslandSet<double> &, const StepParams<double> &, StepDiag *, size_t, hipStream_t);
size_t lcp_lds_need(int real_bytes, int m, int nbd) { return real_bytes == 4 ? lcp_lds_bytes<float>(m, nbd) : lcp_lds_bytes<double>(m, nbd); }

void lcp_grid_stats(dmxBatch *b, int64_t out[8])
{
    for (int k = 0; k < 8; k++) out[k] = 0;
    if (!b->lcp_grid) return;
    const LcpGrid *g = (const LcpGrid *)b->lcp_grid;
    for (int k = 0; k < 8; k++) out[k] = g->stats[k];
}
void lcp_grid_count_fallback(dmxBatch *b) { grid_of(b)->stats[7] += 1; }

hipError_t dmx_touch_lcp(int real_bytes)
{
    hipFuncAttributes a;
    hipError_t e = hipSuccess;
    auto touch = [&](const void *k) { const hipError_t r = hipFuncGetAttributes(&a, k); if (r != hipSuccess) e = r; };
    if (real_bytes == 4) { touch((const void *)&lcp_panel<float>); touch((const void *)&lcp_syrk<float>); }
    else { touch((const void *)&lcp_panel<double>); touch((const void *)&lcp_syrk<double>); }
    return e;
}

}  // namespace dmx
