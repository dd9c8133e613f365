// scripts/ubench_fma_chain.hip -- what bounds step_plane?  Its SOR sweep is a chain of dependent v_fma_f32 (each row update
// feeds the next through the body's accumulators).  This measures the VALU issue ceiling that shape can reach on gfx950:
// N dependent FMAs per lane, C independent chains per lane (ILP), W resident waves per SIMD.  Output: cycles per FMA
// instruction per SIMD at the measured clock, to hold against step_plane's SQ counters (profiles/r01_sq_counters.txt).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_fma_chain scripts/ubench_fma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int C> __global__ __launch_bounds__(64) void chain(float *out, int n, float a, float b)
{
    float x[C];
#pragma unroll
    for (int c = 0; c < C; c++) x[c] = (float)(threadIdx.x + c);
#pragma unroll 32
    for (int i = 0; i < n; i++) {                   // unrolled: the loop's own branch must not be what is measured
#pragma unroll
        for (int c = 0; c < C; c++) x[c] = __builtin_fmaf(x[c], a, b);      // each chain depends on its own previous value only
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < C; c++) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int C> static void run(int waves_per_simd, int simds, double clock_ghz)
{
    const int n = 1 << 16;
    const int blocks = simds * waves_per_simd;              // one 64-lane workgroup = one wave; the dispatcher spreads them over the SIMDs
    float *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    chain<C><<<blocks, 64>>>(out, n, 0.999f, 0.001f);
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0, 0));
        chain<C><<<blocks, 64>>>(out, n, 0.999f, 0.001f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double instr_per_simd = (double)n * C * waves_per_simd;          // wave-instructions issued by one SIMD
    const double cycles = best * 1e-3 * clock_ghz * 1e9;
    printf("  chains/lane %d  waves/SIMD %d: %8.3f ms  %.2f cycles per FMA per SIMD  (%.1f TFLOP/s f32 over the chip)\n", C, waves_per_simd, best,
           cycles / instr_per_simd, 2.0 * 64 * instr_per_simd * simds / (best * 1e-3) / 1e12);
    CHECK(hipFree(out));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int simds = p.multiProcessorCount * 4;
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %d SIMDs, clock %.2f GHz (peak f32 FMA: 2 x 64 lanes x 0.5 instr/cycle x SIMDs x clock = %.1f TFLOP/s)\n", p.name,
           p.multiProcessorCount, simds, ghz, 2.0 * 64 * 0.5 * simds * ghz * 1e9 / 1e12);
    for (int w : { 1, 2, 4, 8 }) run<1>(w, simds, ghz);
    for (int w : { 1, 2, 4 }) run<2>(w, simds, ghz);
    for (int w : { 1, 2, 4 }) run<4>(w, simds, ghz);
    return 0;
}
