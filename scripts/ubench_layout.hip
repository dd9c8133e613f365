// scripts/ubench_layout.hip -- layout microbenchmark for the free-flight integrate kernel: how close to the
// float4-copy rate (~6.3 TB/s on MI355X) does a 17-read / 13-write-per-body pass get under
//   (a) SoA with component stride = n (power of two), (b) SoA with a padded stride,
//   (c) AoSoA tiles of T bodies (30*T contiguous reals per tile), at 4 / 8 / 16 bytes per lane?
// The arithmetic is the free-body step's shape (position, quaternion, velocity), not its exact bits.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_layout scripts/ubench_layout.hip ; run: ./ubench_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NC = 30;   // 13 state + 4 const + 13 padding-to-match (sides, force, torque, bp): only 17 are read

template <int V> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

template <int V> __device__ inline void ld(const float *p, float (&o)[V])
{
    typename Vec<V>::T v = *reinterpret_cast<const typename Vec<V>::T *>(p);
    const float *f = reinterpret_cast<const float *>(&v);
#pragma unroll
    for (int k = 0; k < V; k++) o[k] = f[k];
}
template <int V> __device__ inline void st(float *p, const float (&o)[V])
{
    typename Vec<V>::T v;
    float *f = reinterpret_cast<float *>(&v);
#pragma unroll
    for (int k = 0; k < V; k++) f[k] = o[k];
    *reinterpret_cast<typename Vec<V>::T *>(p) = v;
}

// comp(c) -> pointer to component c of the V bodies this lane owns
template <int V, class Addr> __device__ inline void body_step(float *S, Addr comp, float h)
{
    float x[17][V];
#pragma unroll
    for (int c = 0; c < 17; c++) ld<V>(S + comp(c), x[c]);
#pragma unroll
    for (int k = 0; k < V; k++) {
        float *p[17];
        // 0-2 pos, 3-6 quat, 7-9 lvel, 10-12 avel, 13 mass, 14-16 inertia
        float im = 1.0f / x[13][k];
        x[8][k] = fmaf(h, -9.8f * x[13][k] * im, x[8][k]);
        x[0][k] = fmaf(h, x[7][k], x[0][k]); x[1][k] = fmaf(h, x[8][k], x[1][k]); x[2][k] = fmaf(h, x[9][k], x[2][k]);
        float w0 = x[10][k], w1 = x[11][k], w2 = x[12][k];
        float q0 = x[3][k], q1 = x[4][k], q2 = x[5][k], q3 = x[6][k];
        float d0 = 0.5f * (-w0 * q1 - w1 * q2 - w2 * q3), d1 = 0.5f * (w0 * q0 + w1 * q3 - w2 * q2);
        float d2 = 0.5f * (-w0 * q3 + w1 * q0 + w2 * q1), d3 = 0.5f * (w0 * q2 - w1 * q1 + w2 * q0);
        q0 = fmaf(h, d0, q0); q1 = fmaf(h, d1, q1); q2 = fmaf(h, d2, q2); q3 = fmaf(h, d3, q3);
        float l = 1.0f / sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        x[3][k] = q0 * l; x[4][k] = q1 * l; x[5][k] = q2 * l; x[6][k] = q3 * l;
        x[10][k] = w0 * (1.0f + 1e-9f * x[14][k]); x[11][k] = w1 * (1.0f + 1e-9f * x[15][k]); x[12][k] = w2 * (1.0f + 1e-9f * x[16][k]);
        (void)p;
    }
#pragma unroll
    for (int c = 0; c < 13; c++) st<V>(S + comp(c), x[c]);
}

template <int V> __global__ __launch_bounds__(256) void k_soa(float *S, size_t stride, int n, float h)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i >= (size_t)n) return;
    body_step<V>(S, [=](int c) { return (size_t)c * stride + i; }, h);
}

// AoSoA: tile of T bodies = NC*T contiguous reals; component c of body j in tile t at t*NC*T + c*T + j
template <int V, int T> __global__ __launch_bounds__(256) void k_aosoa(float *S, int n, float h)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i >= (size_t)n) return;
    size_t t = i / T, j = i % T;
    size_t base = t * (size_t)NC * T + j;
    body_step<V>(S, [=](int c) { return base + (size_t)c * T; }, h);
}

// AoSoA with the 17 live components first in a tile of 17*T (state+const) and nothing else: the lower bound
template <int V, int T> __global__ __launch_bounds__(256) void k_aosoa17(float *S, int n, float h)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i >= (size_t)n) return;
    size_t t = i / T, j = i % T;
    size_t base = t * (size_t)17 * T + j;
    body_step<V>(S, [=](int c) { return base + (size_t)c * T; }, h);
}

__global__ void k_copy(const float4 *a, float4 *b, size_t n4)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) b[i] = a[i];
}

template <class F> static double time_us(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) launch();
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

static void report(const char *name, int n, double us)
{
    double bytes = (double)n * 120.0;
    printf("  %-34s %9.2f us  %7.0f GB/s  %6.2f Gbs/s\n", name, us, bytes / us * 1e-3, n / us * 1e-3);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    int sides[] = { 1024, 2048, 4096 };
    for (int side : sides) {
        int n = side * side;
        size_t max_stride = (size_t)n + 8192;
        size_t elems = (size_t)NC * max_stride;
        float *S;
        CHECK(hipMalloc(&S, elems * sizeof(float)));
        std::vector<float> host(elems);
        for (size_t i = 0; i < elems; i++) host[i] = 0.5f + (float)((i * 2654435761u) & 1023) / 1024.0f;
        CHECK(hipMemcpy(S, host.data(), elems * sizeof(float), hipMemcpyHostToDevice));
        int reps = side == 1024 ? 400 : side == 2048 ? 200 : 60;
        float h = 1e-6f;
        printf("n = %d (%.1f MB algorithmic per pass)\n", n, n * 120.0 / 1e6);
        {   // copy reference: n*60 B read + n*60 B written
            size_t n4 = (size_t)n * 60 / 16;
            float4 *a = (float4 *)S, *b = (float4 *)(S + (size_t)n * 15);
            report("float4 copy (same bytes)", n, time_us([&] { k_copy<<<(n4 + 255) / 256, 256>>>(a, b, n4); }, reps));
        }
        size_t pads[] = { 0, 64, 256, 1024 + 256, 4096 + 256 + 64 };
        for (size_t pad : pads) {
            size_t stride = (size_t)n + pad;
            char nm[96];
            snprintf(nm, sizeof nm, "SoA stride=n+%zu V=1", pad);
            report(nm, n, time_us([&] { k_soa<1><<<(n + 255) / 256, 256>>>(S, stride, n, h); }, reps));
            if (pad % 2 == 0) {
                snprintf(nm, sizeof nm, "SoA stride=n+%zu V=2", pad);
                report(nm, n, time_us([&] { k_soa<2><<<(n / 2 + 255) / 256, 256>>>(S, stride, n, h); }, reps));
            }
            if (pad % 4 == 0) {
                snprintf(nm, sizeof nm, "SoA stride=n+%zu V=4", pad);
                report(nm, n, time_us([&] { k_soa<4><<<(n / 4 + 255) / 256, 256>>>(S, stride, n, h); }, reps));
            }
        }
        report("AoSoA T=64  V=1 (30 comps/tile)", n, time_us([&] { k_aosoa<1, 64><<<(n + 255) / 256, 256>>>(S, n, h); }, reps));
        report("AoSoA T=128 V=2", n, time_us([&] { k_aosoa<2, 128><<<(n / 2 + 255) / 256, 256>>>(S, n, h); }, reps));
        report("AoSoA T=256 V=1", n, time_us([&] { k_aosoa<1, 256><<<(n + 255) / 256, 256>>>(S, n, h); }, reps));
        report("AoSoA T=256 V=4", n, time_us([&] { k_aosoa<4, 256><<<(n / 4 + 255) / 256, 256>>>(S, n, h); }, reps));
        report("AoSoA T=1024 V=4", n, time_us([&] { k_aosoa<4, 1024><<<(n / 4 + 255) / 256, 256>>>(S, n, h); }, reps));
        report("AoSoA17 T=64 V=1 (17 comps/tile)", n, time_us([&] { k_aosoa17<1, 64><<<(n + 255) / 256, 256>>>(S, n, h); }, reps));
        report("AoSoA17 T=256 V=4", n, time_us([&] { k_aosoa17<4, 256><<<(n / 4 + 255) / 256, 256>>>(S, n, h); }, reps));
        CHECK(hipFree(S));
    }
    return 0;
}
