// tools/lds_bench.hip -- cost of LDS operations at random addresses (the partition kernels' inner loops): cycles per wave
// instruction per CU for ds_add (no return), ds_add_rtn, ds_write_b32, ds_read_b32, ds_cmpst_rtn over K counters, W waves per CU.
// hipcc --offload-arch=gfx950 -O3 -o tools/lds_bench tools/lds_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

typedef volatile uint32_t __attribute__((address_space(3))) lds_vu32;
typedef volatile uint16_t __attribute__((address_space(3))) lds_vu16;
template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t K, uint32_t reps, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < K; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    uint32_t idx[16];
    uint32_t h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + seed;
#pragma unroll
    for (int j = 0; j < 16; ++j) { h = h * 1664525u + 1013904223u; idx[j] = (h >> 8) % K; }
    uint32_t acc = 0;
    for (uint32_t r = 0; r < reps; ++r) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (OP == 0) atomicAdd(&lds[idx[j]], 1u);                       // result unused: ds_add_u32
            if (OP == 1) acc += atomicAdd(&lds[idx[j]], 1u);                // ds_add_rtn_u32
            if (OP == 2) ((lds_vu32 *)lds)[idx[j]] = acc + j;      // ds_write_b32
            if (OP == 3) acc += ((lds_vu32 *)lds)[idx[j]];         // ds_read_b32
            if (OP == 4) acc += atomicCAS(&lds[idx[j]], acc, acc + 1u);     // ds_cmpst_rtn_b32
            if (OP == 5) ((lds_vu16 *)lds)[idx[j]] = (uint16_t)(acc + j);   // ds_write_b16
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) idx[j] = (idx[j] * 5u + 1u) & (K - 1u);       // (keeps addresses changing; K is a power of two)
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = lds[0];
}

template <int OP> float run(uint32_t K, int wg, uint32_t reps)
{
    uint32_t *d; hipMalloc(&d, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * (wg >= 1024 ? 1 : 1);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(wg), K * 4, 0, d, K, 4u, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(wg), K * 4, 0, d, K, reps, 1u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    // wave instructions per CU: (wg / 64) waves * reps * 16
    const double winst = (double)(wg / 64) * reps * 16.0;
    return (float)(ms * 1e-3 * 2.4e9 / winst);          // cycles (at 2.4 GHz) per wave instruction per CU
}

int main()
{
    const char *names[] = {"ds_add", "ds_add_rtn", "ds_write_b32", "ds_read_b32", "ds_cmpst_rtn", "ds_write_b16"};
    const uint32_t Ks[] = {32, 64, 128, 1024, 16384};
    const int wgs[] = {512, 1024};
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int wg : wgs)
        for (uint32_t K : Ks) {
            printf("wg %4d K %5u:", wg, K);
            printf(" %s %.1f", names[0], run<0>(K, wg, 2000));
            printf(" %s %.1f", names[1], run<1>(K, wg, 2000));
            printf(" %s %.1f", names[2], run<2>(K, wg, 2000));
            printf(" %s %.1f", names[3], run<3>(K, wg, 2000));
            printf(" %s %.1f", names[4], run<4>(K, wg, 2000));
            printf(" %s %.1f", names[5], run<5>(K, wg, 2000));
            printf("  (cycles per wave instruction per CU, one workgroup per CU)\n");
        }
    return 0;
}
