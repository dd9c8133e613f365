// What clock does a kernel that occupies ONE compute unit run at?  A chain of dependent adds timed with the constant 100 MHz
// counter (wall_clock64); one workgroup, then the same chain on every compute unit.  Also: the cost of a workgroup barrier and
// of an LDS read-modify-write hand-over between waves (the level step of solve_island_wg), in nanoseconds.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_clock scripts/ubench_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_chain(unsigned long long *out, int n, float seed)
{
    const unsigned long long t0 = wall_clock64();
    float x = seed + threadIdx.x;
    for (int i = 0; i < n; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);       // dependent chain: n x (issue + latency)
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)x; }
}
__global__ void k_barrier(unsigned long long *out, int n)
{
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; i++) __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
// n level steps: each thread reads 6 floats of "its body" from LDS, runs a 12-deep dependent chain, writes them back; barrier
__global__ void k_level(unsigned long long *out, int n, int active)
{
    __shared__ float fc[6 * 1024];
    for (int i = threadIdx.x; i < 6 * 1024; i += blockDim.x) fc[i] = 1.0f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    const int b = (threadIdx.x * 37) & 1023;
    for (int i = 0; i < n; i++) {
        if ((int)threadIdx.x < active) {
            float a[6];
#pragma unroll
            for (int j = 0; j < 6; j++) a[j] = fc[6 * b + j];
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) d = __builtin_fmaf(a[j], 0.5f, d);
#pragma unroll
            for (int j = 0; j < 6; j++) d = __builtin_fmaf(d, 0.5f, a[j]);
#pragma unroll
            for (int j = 0; j < 6; j++) fc[6 * b + j] = __builtin_fmaf(d, 1e-6f, a[j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main()
{
    unsigned long long *d, h[2];
    CHECK(hipMalloc(&d, 16));
    const int n = 200000;
    for (int blocks : { 1, 1, 256, 2048 }) {
        k_chain<<<blocks, 64>>>(d, n, 1.0f);
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("dependent fma chain, %4d workgroup(s) of one wave: %.2f ns per fma (4 cycles at 2.4 GHz = 1.67 ns)\n", blocks, h[0] * 10.0 / n);
    }
    for (int wg : { 64, 256, 1024 }) {
        k_barrier<<<1, wg>>>(d, 20000);
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("__syncthreads, one workgroup of %4d: %.1f ns each\n", wg, h[0] * 10.0 / 20000);
    }
    for (int wg : { 64, 256 })
        for (int active : { 8, 40, wg }) {
            k_level<<<1, wg>>>(d, 20000, active);
            CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            printf("level step (LDS read 6, 12-deep chain, LDS write 6, barrier), workgroup of %3d, %3d lanes with a row: %.1f ns each\n", wg, active, h[0] * 10.0 / 20000);
        }
    return 0;
}
