// Microbenchmark: LDS atomic throughput on gfx950 (lane-ops per clock per CU) for the op types a gradient
// accumulator could use.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o /tmp/ldsb tools/lds_atomics_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int N = 8192;  // LDS dwords
constexpr int K = 2048;  // ops per lane

template <int OP, int PATTERN>
__global__ void __launch_bounds__(256) k(float *out, uint32_t seed)
{
    __shared__ uint64_t lds64[N / 2];
    float *ldsf = reinterpret_cast<float *>(lds64);
    uint32_t *ldsu = reinterpret_cast<uint32_t *>(lds64);
    for (int i = threadIdx.x; i < N; i += 256) ldsu[i] = 0;
    __syncthreads();
    uint32_t h = threadIdx.x * 2654435761u + seed;
    float acc = 0.f;
    for (int i = 0; i < K; ++i) {
        uint32_t a;
        if (PATTERN == 0) a = (threadIdx.x + i * 256) & (N - 1);          // conflict-free, consecutive lanes
        else if (PATTERN == 1) { h = h * 1664525u + 1013904223u; a = (h >> 8) & (N - 1); }  // random
        else a = ((threadIdx.x >> 3) + i * 32) & (N - 1);                 // 8 lanes share an address
        if (OP == 0) atomicAdd(&ldsf[a], 1.0f);
        else if (OP == 1) atomicAdd(&ldsu[a], 1u);
        else if (OP == 2) atomicAdd(reinterpret_cast<unsigned long long *>(&lds64[a >> 1]), 1ull);
        else if (OP == 3) ldsf[a] = (float)i;
        else if (OP == 4) acc += ldsf[a];
        else if (OP == 5) ldsf[a] += 1.0f;  // non-atomic rmw
    }
    __syncthreads();
    if (OP == 4) out[blockIdx.x * 256 + threadIdx.x] = acc;
    else out[blockIdx.x * 256 + threadIdx.x] = ldsf[threadIdx.x];
}

template <int OP, int PATTERN>
void run(const char *name, float *out)
{
    const int blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<OP, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 1u + r);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double laneops = (double)blocks * 256 * K;
    // 256 CUs at ~2.4 GHz
    printf("%-34s %8.1f us  %6.2f lane-ops/clk/CU\n", name, ms * 1e3, laneops / (ms * 1e-3) / 256 / 2.4e9);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 4 * 256 * 4);
    run<0, 0>("ds_add_f32 conflict-free", out);
    run<1, 0>("ds_add_u32 conflict-free", out);
    run<2, 0>("ds_add_u64 conflict-free", out);
    run<3, 0>("ds_write_b32 conflict-free", out);
    run<4, 0>("ds_read_b32 conflict-free", out);
    run<5, 0>("read+add+write conflict-free", out);
    run<0, 1>("ds_add_f32 random", out);
    run<1, 1>("ds_add_u32 random", out);
    run<2, 1>("ds_add_u64 random", out);
    run<3, 1>("ds_write_b32 random", out);
    run<0, 2>("ds_add_f32 8 lanes/address", out);
    run<1, 2>("ds_add_u32 8 lanes/address", out);
    run<2, 2>("ds_add_u64 8 lanes/address", out);
    return 0;
}
