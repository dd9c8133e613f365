// What does one kernel launch cost the host?  A trivial kernel enqueued 4096 times back to back on one stream, with 16 bytes and with
// 480 bytes of arguments (StepParams is that size); host time per launch and total time per launch.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_launch scripts/ubench_launch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float v[120]; };
__global__ void k_small(float *p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0 && n < 0) p[0] = 1.f; }
__global__ void k_big(float *p, Big b, int n) { if (threadIdx.x == 0 && blockIdx.x == 0 && n < 0) p[0] = b.v[5]; }
int main()
{
    float *d; hipMalloc(&d, 256);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    Big b{};
    const int N = 4096;
    for (int variant = 0; variant < 2; variant++) {
        for (int rep = 0; rep < 3; rep++) {
            hipStreamSynchronize(st);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                if (variant == 0) hipLaunchKernelGGL(k_small, dim3(512), dim3(256), 0, st, d, i);
                else hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, st, d, b, i);
            }
            const auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(st);
            const auto t2 = std::chrono::steady_clock::now();
            if (rep == 2)
                printf("%s arguments: host %.2f us per launch, %.2f us per launch until the stream is idle\n", variant == 0 ? " 16 bytes of" : "496 bytes of",
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
    }
    return 0;
}
