// What an event on a stream's chain costs on gfx950 / ROCm 7: a chain of short kernels on stream A with, between two of
// them, (0) nothing, (1) hipEventRecord, (2) the event riding on the previous launch (hipExtLaunchKernelGGL stopEvent),
// each with a second stream waiting on the event and launching a kernel of its own; (3) a hipStreamWaitEvent on A for an
// event of stream B.   hipcc --offload-arch=gfx950 -O2 tools/event_cost.hip -o build/tmp/event_cost && build/tmp/event_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void k_spin(float *p, int iters)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
int main()
{
    float *a, *b;
    hipMalloc(&a, 4096); hipMalloc(&b, 4096);
    hipStream_t A, B;
    hipStreamCreateWithFlags(&A, hipStreamNonBlocking); hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    hipEvent_t ev, evb, t0, t1;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming); hipEventCreateWithFlags(&evb, hipEventDisableTiming);
    hipEventCreate(&t0); hipEventCreate(&t1);
    const int N = 2000, IT = 2000;  // ~10 us kernels
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto h0 = std::chrono::steady_clock::now();
            hipEventRecord(t0, A);
            for (int i = 0; i < N; ++i) {
                if (mode == 2 || mode == 5) hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, A, nullptr, ev, 0, a, IT);
                else hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, A, a, IT);
                if (mode == 1 || mode == 4) hipEventRecord(ev, A);
                if (mode == 1 || mode == 2) { hipStreamWaitEvent(B, ev, 0); hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, B, b, 100); }
                if (mode == 3) {  // B produces, A waits
                    hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, B, b, 100);
                    hipEventRecord(evb, B);
                    hipStreamWaitEvent(A, evb, 0);
                }
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, A, a, IT);
            }
            hipEventRecord(t1, A);
            auto h1 = std::chrono::steady_clock::now();
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, t0, t1);
            if (rep) printf("mode %d: %.2f us per pair of kernels on A (host enqueue %.2f us)\n", mode, 1e3 * ms / N,
                            1e6 * std::chrono::duration<double>(h1 - h0).count() / N);
        }
    }
    printf("modes: 0 plain, 1 record + other stream waits, 2 stopEvent on the launch + other stream waits, 3 A waits for B's event, "
           "4 record only, 5 stopEvent only\n");
    return 0;
}
