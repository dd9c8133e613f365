// scripts/ubench_pk_chain.hip -- would step_plane gain from two bodies per lane held as float2 (v_pk_fma_f32)?
// Its SOR sweep is latency-bound: a row update is a 6-deep dependent dot-product chain, a clamp, then 6 independent accumulator
// updates, and the next row's chain starts from those.  This measures exactly that shape on gfx950:
//   scalar: one body per lane, v_fma_f32 (what step_plane does), W waves per SIMD;
//   packed: two bodies per lane as float2, v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, W waves per SIMD.
// Output: cycles per ROW UPDATE per SIMD and body-row-updates per cycle per SIMD (the figure that decides).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o ubench_pk_chain scripts/ubench_pk_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
template <class V> struct Ops;
template <> struct Ops<float> {
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float clamp0(float nl, float old, float &delta) { const bool b = nl < 0.f; delta = b ? -old : delta; return b ? 0.f : nl; }
    static __device__ __forceinline__ float splat(float x) { return x; }
    static __device__ __forceinline__ float sum(float x) { return x; }
    static constexpr int bodies = 1;
};
template <> struct Ops<f2> {
    static __device__ __forceinline__ f2 fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ __forceinline__ f2 clamp0(f2 nl, f2 old, f2 &delta) { const i2 b = nl < (f2){ 0.f, 0.f }; delta = b ? -old : delta; return b ? (f2){ 0.f, 0.f } : nl; }
    static __device__ __forceinline__ f2 splat(float x) { return (f2){ x, x * 1.0001f }; }
    static __device__ __forceinline__ float sum(f2 x) { return x.x + x.y; }
    static constexpr int bodies = 2;
};

// ROWS rows in registers (J 6, iM 6, rhs, adcfm, lam), swept `sweeps` times: step_plane's inner loop
template <class V, int ROWS> __global__ __launch_bounds__(64) void sweep(float *out, int sweeps, float seed)
{
    using O = Ops<V>;
    V J[ROWS][6], iM[ROWS][6], rhs[ROWS], adcfm[ROWS], lam[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
#pragma unroll
        for (int j = 0; j < 6; j++) { J[r][j] = O::splat(seed * (float)(r + j + 1 + threadIdx.x % 7) * 1e-3f); iM[r][j] = O::splat(seed * (float)(r * 3 + j + 2) * 1e-3f); }
        rhs[r] = O::splat(seed * (float)(r + 1)); adcfm[r] = O::splat(1e-6f * seed); lam[r] = O::splat(0.f);
    }
    V f[6];
#pragma unroll
    for (int j = 0; j < 6; j++) f[j] = O::splat(0.f);
    for (int it = 0; it < sweeps; it++) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const V old = lam[r];
            V delta = O::fma(-old, adcfm[r], rhs[r]);
            delta -= O::fma(f[5], J[r][5], O::fma(f[4], J[r][4], O::fma(f[3], J[r][3], O::fma(f[2], J[r][2], O::fma(f[1], J[r][1], f[0] * J[r][0])))));
            V nl = old + delta;
            if (r % 3 == 0) nl = O::clamp0(nl, old, delta);            // normal rows clamp at zero, friction rows are unbounded
            lam[r] = nl;
#pragma unroll
            for (int j = 0; j < 6; j++) f[j] = O::fma(delta, iM[r][j], f[j]);
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) s += O::sum(f[j]);
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class V, int ROWS> static void run(const char *name, int waves_per_simd, int simds, double clock_ghz)
{
    const int sweeps = 2000;
    const int blocks = simds * waves_per_simd;
    float *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    sweep<V, ROWS><<<blocks, 64>>>(out, sweeps, 1.0f);
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0, 0));
        sweep<V, ROWS><<<blocks, 64>>>(out, sweeps, 1.0f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    int nregs = 0;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void *)sweep<V, ROWS>) == hipSuccess) nregs = fa.numRegs;
    const double row_updates_per_simd = (double)sweeps * ROWS * waves_per_simd;          // wave-level row updates issued by one SIMD
    const double cycles = best * 1e-3 * clock_ghz * 1e9;
    const double body_rows = row_updates_per_simd * 64 * Ops<V>::bodies;
    printf("  %-6s rows %2d  waves/SIMD %d  regs %3d: %8.3f ms  %6.1f cycles per row update per SIMD, %5.2f body-row-updates per cycle per SIMD\n", name, ROWS,
           waves_per_simd, nregs, best, cycles / row_updates_per_simd, body_rows / cycles);
    CHECK(hipFree(out));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int simds = p.multiProcessorCount * 4;
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d SIMDs, clock %.2f GHz.  One row update = 14 FMA-class instructions + a clamp on every third row.\n", p.name, simds, ghz);
    for (int w : { 1, 2 }) run<float, 12>("scalar", w, simds, ghz);
    for (int w : { 3, 4 }) run<float, 6>("scalar", w, simds, ghz);           // (half the rows so that 3-4 waves fit: what more waves would buy)
    for (int w : { 1, 2 }) run<f2, 12>("packed", w, simds, ghz);
    for (int w : { 1, 2 }) run<f2, 6>("packed", w, simds, ghz);
    return 0;
}
