// scripts/ubench_inplace.hip -- at an HBM-resident size (16 Mi bodies x 120 B = 2 GB per pass), what separates the
// free-flight pass's access pattern (per wave: 17 x 256 B read, 13 x 256 B written back IN PLACE) from a float4 copy?
// Variants: copy out of place / in place; the tile pattern in place / out of place (into a compact 13-component tile
// slab) / with the stores of tile t issued by the wave that reads tile t+G (software-pipelined in-place).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_inplace scripts/ubench_inplace.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_copy(const float4 *a, float4 *b, size_t n4)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) b[i] = a[i];
}
// pure streams: what the memory system gives reads alone and writes alone (the pass is 57 % reads)
__global__ void k_read(const float4 *a, float *out, size_t n4)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 v = i < n4 ? a[i] : make_float4(0, 0, 0, 0);
    float s = v.x + v.y + v.z + v.w;
    if (s == 12345.678f) out[i & 1023] = s;                 // (never true: keeps the load alive without a store stream)
}
__global__ void k_write(float4 *b, size_t n4)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) b[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// r of every 8 float4 are read-modify-written, the rest only read: a read share between 50 % and 100 %
template <int W> __global__ void k_mix(float4 *a, size_t n4)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = a[i];
    if ((blockIdx.x & 7) < W) { v.x *= 1.0000001f; a[i] = v; }
    else if (v.x == 12345.678f) a[i] = v;
}
__global__ void k_scale_inplace(float4 *a, size_t n4)      // read X, write X
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { float4 v = a[i]; v.x *= 1.0000001f; a[i] = v; }
}
// tile pattern: 30 comps x 64 bodies per tile; read comps 0..16, write comps 0..12 to `out` with out tile stride OC comps
template <int OC> __global__ __launch_bounds__(256) void k_tile(const float *S, float *out, int n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n) return;
    size_t t = i >> 6, j = i & 63;
    const float *p = S + t * 30 * 64 + j;
    float x[17];
#pragma unroll
    for (int c = 0; c < 17; c++) x[c] = p[c * 64];
    float s = x[13] * 1e-9f + x[14] * 1e-9f + x[15] * 1e-9f + x[16] * 1e-9f;
    float *o = out + t * OC * 64 + j;
#pragma unroll
    for (int c = 0; c < 13; c++) o[c * 64] = x[c] * (1.0f + s);
}
// same, 4 tiles per wave iteration (more bytes in flight per wave)
template <int OC, int U> __global__ __launch_bounds__(256) void k_tile_u(const float *S, float *out, int n)
{
    size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, j = threadIdx.x & 63;
    size_t t0 = w * U;
    if (t0 * 64 >= (size_t)n) return;
    float x[U][17];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const float *p = S + (t0 + u) * 30 * 64 + j;
#pragma unroll
        for (int c = 0; c < 17; c++) x[u][c] = p[c * 64];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        float s = x[u][13] * 1e-9f + x[u][14] * 1e-9f + x[u][15] * 1e-9f + x[u][16] * 1e-9f;
        float *o = out + (t0 + u) * OC * 64 + j;
#pragma unroll
        for (int c = 0; c < 13; c++) o[c * 64] = x[u][c] * (1.0f + s);
    }
}

// split layout: state tiles (13 comps x 64, dense) in X, constant tiles (4 comps x 64, dense) in C; new state to Y
__global__ __launch_bounds__(256) void k_split(const float *X, const float *C, float *Y, int n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n) return;
    size_t t = i >> 6, j = i & 63;
    const float *p = X + t * 13 * 64 + j, *q = C + t * 4 * 64 + j;
    float x[17];
#pragma unroll
    for (int c = 0; c < 13; c++) x[c] = p[c * 64];
#pragma unroll
    for (int c = 0; c < 4; c++) x[13 + c] = q[c * 64];
    float s = x[13] * 1e-9f + x[14] * 1e-9f + x[15] * 1e-9f + x[16] * 1e-9f;
    float *o = Y + t * 13 * 64 + j;
#pragma unroll
    for (int c = 0; c < 13; c++) o[c * 64] = x[c] * (1.0f + s);
}
// 17-comp dense tiles (state + constants together), new state to a 13-comp dense slab
__global__ __launch_bounds__(256) void k_dense17(const float *X, float *Y, int n, int oc)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n) return;
    size_t t = i >> 6, j = i & 63;
    const float *p = X + t * 17 * 64 + j;
    float x[17];
#pragma unroll
    for (int c = 0; c < 17; c++) x[c] = p[c * 64];
    float s = x[13] * 1e-9f + x[14] * 1e-9f + x[15] * 1e-9f + x[16] * 1e-9f;
    float *o = Y + t * oc * 64 + j;
#pragma unroll
    for (int c = 0; c < 13; c++) o[c * 64] = x[c] * (1.0f + s);
}

// slab contents other than zeros: the memory system moves zeros measurably faster (see main)
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
    return best;
}
static void report(const char *name, size_t n, double us)
{
    printf("  %-58s %9.2f us  %7.0f GB/s  frac %.3f\n", name, us, n * 120.0 / us * 1e-3, n * 120.0 / us * 1e-3 / 8000.0);
    fflush(stdout);
}
int main(int argc, char **argv)
{
    const int side = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t n = (size_t)side * side;
    float *A, *B;
    CHECK(hipMalloc(&A, n * 30 * sizeof(float)));
    CHECK(hipMalloc(&B, n * 30 * sizeof(float)));
    CHECK(hipMemset(A, 0, n * 30 * sizeof(float)));
    CHECK(hipMemset(B, 0, n * 30 * sizeof(float)));
    const bool rnd = argc > 2 && atoi(argv[2]) != 0;       // 0: both slabs hold zeros; 1: pseudo-random values in [0.5, 1.5)
    if (rnd) {
        k_fill<<<(unsigned)((n * 30 + 255) / 256), 256>>>(A, n * 30);
        k_fill<<<(unsigned)((n * 30 + 255) / 256), 256>>>(B, n * 30);
        CHECK(hipDeviceSynchronize());
    }
    printf("slab contents: %s\n", rnd ? "pseudo-random" : "zeros");
    const int reps = 40;
    printf("n = %zu bodies, %.0f MB algorithmic per pass (120 B per body)\n", n, n * 120.0 / 1e6);
    const size_t n4 = n * 60 / 16;
    report("float4 copy, out of place (60 B read + 60 B written / body)", n, time_us([&] { k_copy<<<(n4 + 255) / 256, 256>>>((float4 *)A, (float4 *)B, n4); }, reps));
    report("float4 read-modify-write IN PLACE (same bytes)", n, time_us([&] { k_scale_inplace<<<(n4 + 255) / 256, 256>>>((float4 *)A, n4); }, reps));
    const unsigned g = (unsigned)((n + 255) / 256);
    report("tile pattern 17r/13w, in place (the product's pass)", n, time_us([&] { k_tile<30><<<g, 256>>>(A, A, (int)n); }, reps));
    report("tile pattern, out of place into a 30-comp slab", n, time_us([&] { k_tile<30><<<g, 256>>>(A, B, (int)n); }, reps));
    report("tile pattern, out of place into a compact 13-comp slab", n, time_us([&] { k_tile<13><<<g, 256>>>(A, B, (int)n); }, reps));
    report("tile pattern in place, 2 tiles per wave", n, time_us([&] { k_tile_u<30, 2><<<(g + 1) / 2, 256>>>(A, A, (int)n); }, reps));
    report("tile pattern in place, 4 tiles per wave", n, time_us([&] { k_tile_u<30, 4><<<(g + 3) / 4, 256>>>(A, A, (int)n); }, reps));
    report("tile pattern out of place (13-comp), 4 tiles per wave", n, time_us([&] { k_tile_u<13, 4><<<(g + 3) / 4, 256>>>(A, B, (int)n); }, reps));
    float *Cn = B + n * 14;          // constants: 4 comps per body, behind a 13-comp state region in B
    report("split layout: dense state X + dense consts -> dense state Y (out of place)", n, time_us([&] { k_split<<<g, 256>>>(A, Cn, B, (int)n); }, reps));
    report("split layout, in place (Y = X)", n, time_us([&] { k_split<<<g, 256>>>(A, Cn, A, (int)n); }, reps));
    report("17-comp dense tiles -> 13-comp dense slab (out of place)", n, time_us([&] { k_dense17<<<g, 256>>>(A, B, (int)n, 13); }, reps));
    report("17-comp dense tiles in place", n, time_us([&] { k_dense17<<<g, 256>>>(A, A, (int)n, 17); }, reps));
    {
        auto rep2 = [&](const char *name, double bytes, double us) { printf("  %-58s %9.2f us  %7.0f GB/s  frac %.3f\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0); fflush(stdout); };
        const double b60 = (double)n * 60.0;
        rep2("pure read  (float4, 60 B / body)", b60, time_us([&] { k_read<<<(n4 + 255) / 256, 256>>>((float4 *)A, B, n4); }, reps));
        rep2("pure write (float4, 60 B / body)", b60, time_us([&] { k_write<<<(n4 + 255) / 256, 256>>>((float4 *)B, n4); }, reps));
        rep2("in place, 2 of 8 blocks written back (read share 80 %)", b60 * 1.25, time_us([&] { k_mix<2><<<(n4 + 255) / 256, 256>>>((float4 *)A, n4); }, reps));
        rep2("in place, 4 of 8 blocks written back (read share 67 %)", b60 * 1.5, time_us([&] { k_mix<4><<<(n4 + 255) / 256, 256>>>((float4 *)A, n4); }, reps));
        rep2("in place, 6 of 8 blocks written back (read share 57 %)", b60 * 1.75, time_us([&] { k_mix<6><<<(n4 + 255) / 256, 256>>>((float4 *)A, n4); }, reps));
    }
    report("float4 copy again", n, time_us([&] { k_copy<<<(n4 + 255) / 256, 256>>>((float4 *)A, (float4 *)B, n4); }, reps));
    return 0;
}
