// Host cost of hipLaunchKernelGGL on gfx950 / ROCm 7 by kernel-argument size and dynamic LDS: an empty kernel launched 20,000
// times back to back (the GPU keeps up), host time per launch.   hipcc --offload-arch=gfx950 -O2 tools/launch_cost.hip -o build/tmp/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Blob { unsigned v[N]; };
template <int N> __global__ void k_args(Blob<N> b, float *p) { if (b.v[0] == 12345u && threadIdx.x == 0) p[0] = (float)b.v[N - 1]; }
__global__ void k_lds(float *p) { extern __shared__ float s[]; if (threadIdx.x == 999) p[0] = s[0]; }
template <class F> double per_launch(F f, hipStream_t st, int n = 20000)
{
    for (int i = 0; i < 200; ++i) f();
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(st);
    return 1e6 * std::chrono::duration<double>(t1 - t0).count() / n;
}
int main()
{
    float *p; hipMalloc(&p, 64);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    Blob<4> b4{}; Blob<64> b64{}; Blob<256> b256{}; Blob<640> b640{}; Blob<1000> b1000{};
    printf("args   16 B: %.2f us\n", per_launch([&] { hipLaunchKernelGGL(k_args<4>, dim3(1), dim3(64), 0, st, b4, p); }, st));
    printf("args  256 B: %.2f us\n", per_launch([&] { hipLaunchKernelGGL(k_args<64>, dim3(1), dim3(64), 0, st, b64, p); }, st));
    printf("args 1024 B: %.2f us\n", per_launch([&] { hipLaunchKernelGGL(k_args<256>, dim3(1), dim3(64), 0, st, b256, p); }, st));
    printf("args 2560 B: %.2f us\n", per_launch([&] { hipLaunchKernelGGL(k_args<640>, dim3(1), dim3(64), 0, st, b640, p); }, st));
    printf("args 4000 B: %.2f us\n", per_launch([&] { hipLaunchKernelGGL(k_args<1000>, dim3(1), dim3(64), 0, st, b1000, p); }, st));
    hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    printf("lds 32 KB dynamic: %.2f us\n", per_launch([&] { hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 32768, st, p); }, st));
    printf("grid 2640 x 256 threads, args 2560 B: %.2f us\n",
           per_launch([&] { hipLaunchKernelGGL(k_args<640>, dim3(2640), dim3(256), 0, st, b640, p); }, st, 5000));
    return 0;
}
