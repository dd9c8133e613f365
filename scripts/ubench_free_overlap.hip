// Microbenchmark: how much of integrate_free's distance from the bare tile pattern (scripts/ubench_inplace.hip) is its ARITHMETIC,
// and which launch shape overlaps it best with the memory stream.  The product's pass reads 17 components and writes 13 of a
// 64-body tile per wavefront with ~450 vector instructions in between; here the same memory pattern carries K fused
// multiply-adds per lane (independent chains over the 13 state values, so the stream is issue-bound like the product's).
//   one-shot grid, 256- and 64-lane workgroups, occupancy capped by an LDS allocation; and a persistent grid whose waves
//   fetch tile t+1 into a second register set before working on tile t.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_free_overlap scripts/ubench_free_overlap.hip ; run: ./ubench_free_overlap [side]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int K> __device__ __forceinline__ void work(float (&x)[17])
{
    const float c = x[13] * 1e-9f + x[14] * 1e-9f + x[15] * 1e-9f + x[16] * 1e-9f;
#pragma unroll
    for (int k = 0; k < K; k++) x[k % 13] = __builtin_fmaf(x[(k + 5) % 13], c, x[k % 13]);
}

// what is it about arithmetic on real data?  MODE 0: 13 multiplies only (the data changes every pass); 1: K fused multiply-adds whose
// result decides nothing but a never-true predicate, the loaded values go back unchanged; 2: K integer multiply-adds on the bits
template <int K, int MODE> __global__ __launch_bounds__(256) void k_probe(const float *S, float *out, int n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n) return;
    size_t t = i >> 6, j = i & 63;
    const float *p = S + t * 30 * 64 + j;
    float x[17], y[17];
#pragma unroll
    for (int c = 0; c < 17; c++) { x[c] = p[c * 64]; y[c] = x[c]; }
    if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < 13; c++) y[c] = x[c] * 1.0000001f;
    } else if (MODE == 1) {
        work<K>(x);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 13; c++) s += x[c];
        if (s == 12345.678f) y[0] = s;
    } else {
        unsigned u[13];
#pragma unroll
        for (int c = 0; c < 13; c++) u[c] = __float_as_uint(x[c]);
        const unsigned m = __float_as_uint(x[13]) | 1u;
#pragma unroll
        for (int k = 0; k < K; k++) u[k % 13] = u[(k + 5) % 13] * m + u[k % 13];
        unsigned s = 0;
#pragma unroll
        for (int c = 0; c < 13; c++) s ^= u[c];
        if (s == 0x12345678u) y[0] = 1.f;
    }
    float *o = out + t * 30 * 64 + j;
#pragma unroll
    for (int c = 0; c < 13; c++) o[c * 64] = y[c];
}
// arithmetic alone: the clock under K-long chains on zeros or on real data
template <int K> __global__ __launch_bounds__(256) void k_alu(const float *S, float *out, int n, int loops)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n) return;
    size_t t = i >> 6, j = i & 63;
    const float *p = S + t * 30 * 64 + j;
    float x[17];
#pragma unroll
    for (int c = 0; c < 17; c++) x[c] = p[c * 64];
    for (int l = 0; l < loops; l++) work<K>(x);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 13; c++) s += x[c];
    if (s == 12345.678f) out[i] = s;
}

template <int K, int BS> __global__ __launch_bounds__(BS) void k_oneshot(const float *S, float *out, int n)
{
    extern __shared__ float lds_cap[];
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    if (i >= (size_t)n) return;
    size_t t = i >> 6, j = i & 63;
    const float *p = S + t * 30 * 64 + j;
    float x[17];
#pragma unroll
    for (int c = 0; c < 17; c++) x[c] = p[c * 64];
    work<K>(x);
    float *o = out + t * 30 * 64 + j;
#pragma unroll
    for (int c = 0; c < 13; c++) o[c * 64] = x[c];
}

// persistent: every wave walks tiles w, w + nw, ...; the next tile's loads are issued before the current tile's arithmetic
template <int K> __global__ __launch_bounds__(256) void k_pipe(const float *S, float *out, int ntiles)
{
    const int nw = gridDim.x * 4, w0 = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
    if (w0 >= ntiles) return;
    float x[17], y[17];
    {
        const float *p = S + (size_t)w0 * 30 * 64 + j;
#pragma unroll
        for (int c = 0; c < 17; c++) x[c] = p[c * 64];
    }
    for (int t = w0; t < ntiles; t += nw) {
        const int tn = t + nw;
        if (tn < ntiles) {
            const float *p = S + (size_t)tn * 30 * 64 + j;
#pragma unroll
            for (int c = 0; c < 17; c++) y[c] = p[c * 64];
        }
        work<K>(x);
        float *o = out + (size_t)t * 30 * 64 + j;
#pragma unroll
        for (int c = 0; c < 13; c++) o[c * 64] = x[c];
#pragma unroll
        for (int c = 0; c < 17; c++) x[c] = y[c];
    }
}
// persistent without the prefetch (control for the grid shape alone)
template <int K> __global__ __launch_bounds__(256) void k_persist(const float *S, float *out, int ntiles)
{
    const int nw = gridDim.x * 4, w0 = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
    for (int t = w0; t < ntiles; t += nw) {
        const float *p = S + (size_t)t * 30 * 64 + j;
        float x[17];
#pragma unroll
        for (int c = 0; c < 17; c++) x[c] = p[c * 64];
        work<K>(x);
        float *o = out + (size_t)t * 30 * 64 + j;
#pragma unroll
        for (int c = 0; c < 13; c++) o[c * 64] = x[c];
    }
}

__global__ void k_fill(float *a, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    a[i] = 0.5f + (float)(h & 0xffffff) * (1.0f / 16777216.0f);
}

template <class F> static double time_us(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) launch();
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; i++) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e3 / reps < best) best = ms * 1e3 / reps;
    }
    CHECK(hipGetLastError());
    return best;
}
static void report(const char *name, size_t n, double us)
{
    printf("  %-74s %9.2f us  frac %.3f\n", name, us, n * 120.0 / us * 1e-3 / 8000.0);
    fflush(stdout);
}

template <int K> static void run_k(float *A, size_t n, int reps)
{
    char nm[160];
    const unsigned g = (unsigned)((n + 255) / 256);
    snprintf(nm, sizeof nm, "K=%3d one-shot, 256-lane groups", K);
    report(nm, n, time_us([&] { k_oneshot<K, 256><<<g, 256>>>(A, A, (int)n); }, reps));
    snprintf(nm, sizeof nm, "K=%3d one-shot, 64-lane groups", K);
    report(nm, n, time_us([&] { k_oneshot<K, 64><<<(unsigned)((n + 63) / 64), 64>>>(A, A, (int)n); }, reps));
    for (int per_cu : { 2, 4, 6, 7 }) {
        const size_t lds = (size_t)(160 * 1024 / per_cu) & ~(size_t)255;
        CHECK(hipFuncSetAttribute((const void *)k_oneshot<K, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        snprintf(nm, sizeof nm, "K=%3d one-shot, 256-lane groups, at most %d waves per SIMD", K, per_cu);
        report(nm, n, time_us([&] { k_oneshot<K, 256><<<g, 256, lds>>>(A, A, (int)n); }, reps));
    }
    const int ntiles = (int)(n / 64);
    for (int per_cu : { 2, 4, 6, 8 }) {
        snprintf(nm, sizeof nm, "K=%3d persistent grid, %d groups per CU", K, per_cu);
        report(nm, n, time_us([&] { k_persist<K><<<256 * per_cu, 256>>>(A, A, ntiles); }, reps));
        snprintf(nm, sizeof nm, "K=%3d persistent grid, %d groups per CU, next tile fetched before the arithmetic", K, per_cu);
        report(nm, n, time_us([&] { k_pipe<K><<<256 * per_cu, 256>>>(A, A, ntiles); }, reps));
    }
}

int main(int argc, char **argv)
{
    const int side = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t n = (size_t)side * side;
    float *A;
    CHECK(hipMalloc(&A, n * 30 * sizeof(float)));
    CHECK(hipMemset(A, 0, n * 30 * sizeof(float)));
    const bool rnd = argc > 2 && atoi(argv[2]) != 0;       // 0: the slab holds zeros; 1: pseudo-random values in [0.5, 1.5)
    if (rnd) { k_fill<<<(unsigned)((n * 30 + 255) / 256), 256>>>(A, n * 30); CHECK(hipDeviceSynchronize()); }
    printf("slab contents: %s\n", rnd ? "pseudo-random" : "zeros");
    const int reps = 40;
    printf("n = %zu bodies, tile pattern 17 read / 13 written in place, K fused multiply-adds per lane in between\n", n);
    const unsigned g = (unsigned)((n + 255) / 256);
    if (argc > 3 && atoi(argv[3]) != 0) {           // the probes only
        report("K=  0 (bare pattern)", n, time_us([&] { k_oneshot<0, 256><<<g, 256>>>(A, A, (int)n); }, reps));
        report("K=  8", n, time_us([&] { k_oneshot<8, 256><<<g, 256>>>(A, A, (int)n); }, reps));
        report("K= 16", n, time_us([&] { k_oneshot<16, 256><<<g, 256>>>(A, A, (int)n); }, reps));
        report("K= 32", n, time_us([&] { k_oneshot<32, 256><<<g, 256>>>(A, A, (int)n); }, reps));
        report("K= 64", n, time_us([&] { k_oneshot<64, 256><<<g, 256>>>(A, A, (int)n); }, reps));
        report("K=128", n, time_us([&] { k_oneshot<128, 256><<<g, 256>>>(A, A, (int)n); }, reps));
        report("13 multiplies, every value changes each pass", n, time_us([&] { k_probe<0, 0><<<g, 256>>>(A, A, (int)n); }, reps));
        report("K=128 fused multiply-adds, the LOADED values written back", n, time_us([&] { k_probe<128, 1><<<g, 256>>>(A, A, (int)n); }, reps));
        report("K=448 fused multiply-adds, the LOADED values written back", n, time_us([&] { k_probe<448, 1><<<g, 256>>>(A, A, (int)n); }, reps));
        report("K=128 integer multiply-adds, the loaded values written back", n, time_us([&] { k_probe<128, 2><<<g, 256>>>(A, A, (int)n); }, reps));
        report("K=  0 (bare pattern) again", n, time_us([&] { k_oneshot<0, 256><<<g, 256>>>(A, A, (int)n); }, reps));
        const double a1 = time_us([&] { k_alu<416><<<g, 256>>>(A, A + n * 20, (int)n, 8); }, reps);
        printf("  arithmetic alone, 8 x 416 fused multiply-adds per lane: %.2f us = %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", a1,
               a1 * 1e-6 * 2.4e9 / ((double)n / 64 / 1024 * 8 * 416));
        return 0;
    }
    run_k<0>(A, n, reps);
    run_k<128>(A, n, reps);
    run_k<256>(A, n, reps);
    run_k<448>(A, n, reps);
    return 0;
}
