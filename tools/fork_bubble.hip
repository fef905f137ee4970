// Where the idle time behind a fork on the training step's chain comes from (gfx950 / ROCm 7).  The step's timeline
// (profiles/r05_timeline_3steps_*.csv) shows ~18 us between the end of the data-gradient kernel and the start of the table
// backward when the weight-gradient kernels are forked off behind the former, ~7 us when they are not.  This program
// rebuilds that piece of the chain out of memory-streaming kernels of the same sizes and reads the gaps from device
// timestamps (s_memrealtime, 100 MHz) written by the kernels themselves -- no profiler in the way:
//
//   main   : K0 (fork F) Kmid Kmid .... K1 (fork E) [wait J] K2 ............ [wait JW]
//   helper1:            wait F, Kbin (record J) ...... wait E, K3 ........ (record JW)
//   helper2:                                            wait E, K4 -> helper1 waits for it
//
//   hipcc --offload-arch=gfx950 -O2 tools/fork_bubble.hip -o build/tmp/fork_bubble && build/tmp/fork_bubble
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned long long u64;

__global__ void k_stream(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4, int passes, u64 *ts)
{
    if (threadIdx.x == 0) atomicMin(&ts[0], (u64)wall_clock64());
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            float4 v = in[i];
            v.x += 1.f + p;
            out[i] = v;
        }
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&ts[1], (u64)wall_clock64());
}

struct Scenario {
    const char *name;
    int fork;          // 0 none, 1 hipEventRecord, 2 the event rides on K1
    bool join_bin;     // main waits for helper1's Kbin before K2
    unsigned ev_flags; // extra event flags
    bool owner_first;  // K2 issued before the helpers' waits
    bool small_k1;     // K1 writes 1 MB instead of 40 MB
    bool one_helper;   // no helper2
};

int main(int argc, char **argv)
{
    const int ITERS = argc > 1 ? atoi(argv[1]) : 300;
    const size_t MB = 1 << 20;
    float4 *a, *b, *c, *d, *e, *f;
    hipMalloc(&a, 64 * MB); hipMalloc(&b, 64 * MB); hipMalloc(&c, 64 * MB); hipMalloc(&d, 64 * MB); hipMalloc(&e, 64 * MB); hipMalloc(&f, 64 * MB);
    hipMemset(a, 0, 64 * MB); hipMemset(c, 0, 64 * MB); hipMemset(e, 0, 64 * MB);
    u64 *ts;  // [ITERS][8 kernels][2]
    const size_t n_ts = (size_t)ITERS * 8 * 2;
    hipMalloc(&ts, n_ts * sizeof(u64));
    std::vector<u64> init(n_ts), host(n_ts);
    for (size_t i = 0; i < n_ts; ++i) init[i] = (i & 1) ? 0ull : ~0ull;
    hipStream_t M, H1, H2;
    hipStreamCreateWithFlags(&M, hipStreamNonBlocking); hipStreamCreateWithFlags(&H1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&H2, hipStreamNonBlocking);
    const Scenario sc[] = {
        {"0 no fork behind K1 (K3/K4 not launched), join for Kbin", 0, true, 0, false, false, false},
        {"1 fork by hipEventRecord", 1, true, 0, false, false, false},
        {"2 fork riding on K1 (the step's form)", 2, true, 0, false, false, false},
        {"3 riding, no join for Kbin in front of K2", 2, false, 0, false, false, false},
        {"4 riding, events hipEventReleaseToDevice", 2, true, hipEventReleaseToDevice, false, false, false},
        {"5 riding, events hipEventDisableSystemFence", 2, true, hipEventDisableSystemFence, false, false, false},
        {"6 riding, K2 issued before the helpers' waits", 2, true, 0, true, false, false},
        {"7 riding, K1 writes 1 MB", 2, true, 0, false, true, false},
        {"8 riding, one helper stream", 2, true, 0, false, false, true},
        {"9 no fork, no join (plain chain)", 0, false, 0, false, false, false},
    };
    for (const Scenario &s : sc) {
        hipEvent_t F, E, J, JW, H2D;
        const unsigned fl = hipEventDisableTiming | s.ev_flags;
        hipEventCreateWithFlags(&F, fl); hipEventCreateWithFlags(&E, fl); hipEventCreateWithFlags(&J, fl);
        hipEventCreateWithFlags(&JW, fl); hipEventCreateWithFlags(&H2D, fl);
        double best_total = 1e30;
        double g12 = 0, g13 = 0, g01 = 0, k1 = 0, k2 = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpy(ts, init.data(), n_ts * sizeof(u64), hipMemcpyHostToDevice);
            hipDeviceSynchronize();
            for (int it = 0; it < ITERS; ++it) {
                u64 *t = ts + (size_t)it * 16;
                // K0: ~15 us of streaming (the compositing pair), fork F rides on it
                hipExtLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, M, nullptr, F, 0, a, b, 24 * MB / 16, 1, t + 0);
                hipStreamWaitEvent(H1, F, 0);
                hipLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, H1, c, d, 16 * MB / 16, 2, t + 2);  // Kbin ~40 us
                hipEventRecord(J, H1);
                // two kernels of the chain's middle (colour MLP, compositing): Kbin is long done when K1 ends, as in the step
                hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, M, a, b, 24 * MB / 16, 1, t + 12);
                hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, M, b, a, 24 * MB / 16, 1, t + 14);
                // K1: the data-gradient kernel: reads 24 MB, writes 40 MB
                const size_t n1 = (s.small_k1 ? 1 : 40) * MB / 16;
                if (s.fork == 2) hipExtLaunchKernelGGL(k_stream, dim3(768), dim3(256), 0, M, nullptr, E, 0, b, a, n1, 1, t + 4);
                else hipLaunchKernelGGL(k_stream, dim3(768), dim3(256), 0, M, b, a, n1, 1, t + 4);
                if (s.fork == 1) hipEventRecord(E, M);
                auto helpers = [&]() {
                    if (!s.fork) return;
                    hipStreamWaitEvent(H1, E, 0);
                    hipLaunchKernelGGL(k_stream, dim3(128), dim3(256), 0, H1, a, d, 16 * MB / 16, 2, t + 8);  // K3
                    if (!s.one_helper) {
                        hipStreamWaitEvent(H2, E, 0);
                        hipLaunchKernelGGL(k_stream, dim3(128), dim3(256), 0, H2, a, f, 16 * MB / 16, 2, t + 10);  // K4
                        hipEventRecord(H2D, H2);
                        hipStreamWaitEvent(H1, H2D, 0);
                    }
                };
                if (!s.owner_first) helpers();
                if (s.join_bin) hipStreamWaitEvent(M, J, 0);
                // K2: the table backward: ~100 us
                hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, M, e, c, 64 * MB / 16, 4, t + 6);
                if (s.owner_first) helpers();
                hipEventRecord(JW, H1);
                hipStreamWaitEvent(M, JW, 0);
            }
            hipDeviceSynchronize();
            hipMemcpy(host.data(), ts, n_ts * sizeof(u64), hipMemcpyDeviceToHost);
            double s12 = 0, s13 = 0, s01 = 0, d1 = 0, d2 = 0;
            int n = 0;
            for (int it = 20; it < ITERS; ++it) {
                const u64 *t = host.data() + (size_t)it * 16;
                s12 += (double)(t[6] - t[5]);
                s01 += (double)(t[12] - t[1]);
                if (s.fork) s13 += (double)(t[8] - t[5]);
                d1 += (double)(t[5] - t[4]);
                d2 += (double)(t[7] - t[6]);
                ++n;
            }
            const double total = (double)(host[(size_t)(ITERS - 1) * 16 + 7] - host[(size_t)20 * 16 + 0]) / (ITERS - 21);
            if (total < best_total) {
                best_total = total;
                g12 = s12 / n; g13 = s13 / n; g01 = s01 / n; k1 = d1 / n; k2 = d2 / n;
            }
        }
        // 100 MHz ticks -> us
        printf("%-58s iter %.1f us | K0 end -> Kmid start %.2f | K1 end -> K2 start %.2f | K1 end -> K3 start %.2f | K1 %.1f K2 %.1f us\n",
               s.name, best_total / 100.0, g01 / 100.0, g12 / 100.0, g13 / 100.0, k1 / 100.0, k2 / 100.0);
        hipEventDestroy(F); hipEventDestroy(E); hipEventDestroy(J); hipEventDestroy(JW); hipEventDestroy(H2D);
    }
    return 0;
}
