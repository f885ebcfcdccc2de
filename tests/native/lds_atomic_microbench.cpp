// Microbenchmark: LDS integer atomics on gfx950 (ds_add_u32 with and without return) next to the float
// atomic and the plain read-add-write, per wave instruction, for 64 / 16 / 4 active lanes; single-wave
// workgroups (20 per CU, 8 KB table each) and 16-wave workgroups sharing one 128 KB table.
// Development tool.  Build: hipcc --offload-arch=gfx950 -O3 tests/native/lds_atomic_microbench.cpp -o /tmp/lds_amb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// MODE 0: ds_add_f32   1: plain RMW (u32)   2: ds_add_u32 (no return)   3: ds_add_rtn_u32 (result consumed)
template <int MODE>
__global__ void __launch_bounds__(1024) bench(const unsigned *__restrict__ slots, int iters, int active, unsigned mask,
                                              unsigned *sink, unsigned long long *cycles) {
    extern __shared__ unsigned tab[];
    const int lane = threadIdx.x & 63;
    for (unsigned x = threadIdx.x; x <= mask; x += blockDim.x) tab[x] = 0;
    __syncthreads();
    unsigned s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = slots[((size_t)blockIdx.x * 8 + u) * blockDim.x + threadIdx.x] & mask;
    unsigned keep = 0;
    const unsigned v = 1u + lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (lane < active) {
                if (MODE == 0) {
                    (void)__hip_atomic_fetch_add((float *)&tab[s[u]], (float)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (MODE == 1) {
                    tab[s[u]] = tab[s[u]] + v;
                } else if (MODE == 2) {
                    (void)__hip_atomic_fetch_add(&tab[s[u]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    keep += __hip_atomic_fetch_add(&tab[s[u]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = tab[threadIdx.x & mask] + keep;
}

template <int MODE>
void run(const char *name, const unsigned *d_slots, int blocks, int threads, int lds_bytes, int active, unsigned *sink,
         unsigned long long *d_cyc) {
    const int iters = 1000;
    const unsigned mask = lds_bytes / 4 - 1;
    hipFuncSetAttribute((const void *)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), lds_bytes, 0, d_slots, iters, active, mask, sink, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), lds_bytes, 0, d_slots, iters, active, mask, sink, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * (threads / 64) * iters * 8.0;
    printf("%-8s wg %4d thr x %5d, lds %6d B, active %2d: kernel %8.3f ms => %6.1f clk per wave-instr per CU (%.2f clk per active lane)\n",
           name, threads, blocks, lds_bytes, active, ms, ms * 1e-3 * 2.4e9 * 256.0 / wave_instr,
           ms * 1e-3 * 2.4e9 * 256.0 / wave_instr / active);
}

int main() {
    const size_t n = (size_t)256 * 20 * 8 * 64;
    std::vector<unsigned> h(n);
    srand(3);
    for (auto &x : h) x = (unsigned)rand();
    unsigned *d_slots, *sink; unsigned long long *d_cyc;
    hipMalloc(&d_slots, n * 4); hipMalloc(&sink, n * 4); hipMalloc(&d_cyc, 256 * 20 * 8);
    hipMemcpy(d_slots, h.data(), n * 4, hipMemcpyHostToDevice);
    struct Cfg { int blocks, threads, lds; } cfgs[] = {{256 * 20, 64, 8192}, {256, 1024, 131072}, {256 * 4, 256, 32768}};
    for (auto c : cfgs)
        for (int active : {64, 16, 4}) {
            run<0>("add_f32", d_slots, c.blocks, c.threads, c.lds, active, sink, d_cyc);
            run<1>("plainRMW", d_slots, c.blocks, c.threads, c.lds, active, sink, d_cyc);
            run<2>("add_u32", d_slots, c.blocks, c.threads, c.lds, active, sink, d_cyc);
            run<3>("add_rtn", d_slots, c.blocks, c.threads, c.lds, active, sink, d_cyc);
        }
    return 0;
}
