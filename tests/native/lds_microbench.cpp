// Microbenchmark: cost of LDS float atomics vs plain read-add-write on gfx950, per wave instruction,
// for 64 / 8 / 1 active lanes, 1 and 20 single-wave workgroups per CU.  Development tool.
// Build: hipcc --offload-arch=gfx950 -O3 tests/native/lds_microbench.cpp -o /tmp/lds_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int MODE>   // 0: ds_add_f32 atomic   1: plain RMW   2: ds_read only  3: ds_write only
__global__ void __launch_bounds__(64) bench(const unsigned *__restrict__ slots, int iters, int active, float *sink,
                                            unsigned long long *cycles) {
    extern __shared__ float acc[];
    const int lane = threadIdx.x;
    for (int x = lane; x < 2048; x += 64) acc[x] = 0.f;
    __syncthreads();
    unsigned s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = slots[(blockIdx.x * 8 + u) * 64 + lane] & 2047u;
    float v = 1.0f + lane;
    float keep = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (lane < active) {
                if (MODE == 0) {
                    (void)__hip_atomic_fetch_add(&acc[s[u]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (MODE == 1) {
                    acc[s[u]] = acc[s[u]] + v;
                } else if (MODE == 2) {
                    keep += acc[s[u]];
                } else {
                    acc[s[u]] = v;
                }
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = acc[lane] + keep;
}

template <int MODE>
void run(const char *name, const unsigned *d_slots, int blocks, int active, float *sink, unsigned long long *d_cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(64), 8192, 0, d_slots, iters, active, sink, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(64), 8192, 0, d_slots, iters, active, sink, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : c) avg += (double)x; avg /= blocks;
    printf("%-10s blocks %5d active %2d: %.1f clk per wave-instr (per wave), kernel %.3f ms => %.2f wave-instr/clk/CU-equivalent\n",
           name, blocks, active, avg / (iters * 8.0), ms, (double)blocks * iters * 8.0 / (ms * 1e-3 * 2.4e9) / 256.0);
}

int main() {
    const int maxb = 256 * 20;
    std::vector<unsigned> h((size_t)maxb * 8 * 64);
    srand(3);
    for (auto &x : h) x = rand();
    unsigned *d_slots; float *sink; unsigned long long *d_cyc;
    hipMalloc(&d_slots, h.size() * 4); hipMalloc(&sink, (size_t)maxb * 64 * 4); hipMalloc(&d_cyc, maxb * 8);
    hipMemcpy(d_slots, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 256 * 20})
        for (int active : {64, 8, 1}) {
            run<0>("atomic", d_slots, blocks, active, sink, d_cyc);
            run<1>("plainRMW", d_slots, blocks, active, sink, d_cyc);
            run<2>("read", d_slots, blocks, active, sink, d_cyc);
            run<3>("write", d_slots, blocks, active, sink, d_cyc);
        }
    return 0;
}
