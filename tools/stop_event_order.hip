// Does an event that rides on a launch (hipExtLaunchKernelGGL stopEvent) order a kernel on ANOTHER stream behind the launch's
// COMPLETION?  Stream A: a kernel that spins ~20 us and then writes slot i; stream B waits for the event and checks the slot.
//   hipcc --offload-arch=gfx950 -O2 tools/stop_event_order.hip -o build/tmp/stop_event_order && build/tmp/stop_event_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_produce(int *slots, int i, int spin)
{
    float v = (float)i;
    for (int k = 0; k < spin; ++k) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0 && blockIdx.x == 0) slots[i] = v > -1.f ? i : -i;
}
__global__ void k_check(const int *slots, int *bad, int i)
{
    if (threadIdx.x == 0 && slots[i] != i) atomicAdd(bad, 1);
}
int main()
{
    const int N = 3000;
    int *slots, *bad;
    (void)hipMalloc(&slots, (N + 1) * sizeof(int)); (void)hipMemset(slots, 0xff, (N + 1) * sizeof(int));
    (void)hipMalloc(&bad, 2 * sizeof(int)); (void)hipMemset(bad, 0, 2 * sizeof(int));
    hipStream_t A, B;
    (void)hipStreamCreateWithFlags(&A, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    hipEvent_t ev[8];
    for (auto &e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (int mode = 0; mode < 2; ++mode) {   // 0: hipEventRecord behind the launch, 1: the event rides on the launch
        (void)hipMemset(slots, 0xff, (N + 1) * sizeof(int));
        (void)hipDeviceSynchronize();
        for (int i = 1; i <= N; ++i) {
            hipEvent_t e = ev[i & 7];
            if (mode == 0) {
                hipLaunchKernelGGL(k_produce, dim3(64), dim3(256), 0, A, slots, i, 4000);
                (void)hipEventRecord(e, A);
            } else {
                hipExtLaunchKernelGGL(k_produce, dim3(64), dim3(256), 0, A, nullptr, e, 0, slots, i, 4000);
            }
            (void)hipStreamWaitEvent(B, e, 0);
            hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, B, slots, bad + mode, i);
        }
        (void)hipDeviceSynchronize();
    }
    int h[2];
    (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("consumer saw a stale slot: hipEventRecord %d of %d, stopEvent on the launch %d of %d\n", h[0], N, h[1], N);
    return 0;
}
