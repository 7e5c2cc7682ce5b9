// tools/atomic_bench.hip -- microbenchmark of scattered table updates on MI355X (not product code).
// Measures random-address update rates for the forms the score table could use.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z; }

template <int MODE>
__global__ __launch_bounds__(256) void k_upd(uint32_t *tab, uint64_t bytes, uint64_t n_upd, int per_thread)
{
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int k = 0; k < per_thread; ++k) {
        uint64_t i = tid * per_thread + k;
        if (i >= n_upd) break;
        uint64_t cell = mix64(i * 0x9E3779B97F4A7C15ull + 12345) % bytes;
        uint32_t *w = tab + (cell >> 2);
        uint32_t sh = (cell & 3) * 8;
        if (MODE == 0) atomicAdd(w, 1u << sh);                       // no-return 32-bit add
        else if (MODE == 1) acc += atomicAdd(w, 1u << sh);           // returning add
        else if (MODE == 2) {                                        // exact byte add by CAS, optimistic zero
            uint32_t expect = 0;
            for (;;) { uint32_t b = ((expect >> sh) + 1u) & 255u; uint32_t want = (expect & ~(255u << sh)) | (b << sh);
                       uint32_t old = atomicCAS(w, expect, want); if (old == expect) break; expect = old; }
        } else if (MODE == 3) { *w = 1u << sh; }                      // plain scattered store
        else if (MODE == 4) { acc += *w; }                            // plain scattered load
        else if (MODE == 5) { __hip_atomic_fetch_add(w, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
    if (acc == 0x12345678u) tab[0] = acc;
}

template <int MODE> static void run(const char *name, uint32_t *tab, uint64_t bytes, uint64_t n_upd)
{
    const int per = 8;
    uint64_t threads = (n_upd + per - 1) / per;
    dim3 grid((uint32_t)((threads + 255) / 256));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(tab, 0, bytes));
    hipLaunchKernelGGL(k_upd<MODE>, grid, dim3(256), 0, 0, tab, bytes, n_upd, per);   // warm
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipMemset(tab, 0, bytes));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_upd<MODE>, grid, dim3(256), 0, 0, tab, bytes, n_upd, per);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-28s table %8.1f MB  updates %6.1f M  %8.3f ms  %7.2f G upd/s\n", name, bytes / 1e6, n_upd / 1e6, best, n_upd / best / 1e6);
}

int main()
{
    const uint64_t sizes[] = {50ull << 20, 200ull << 20, 2048ull << 20, 8192ull << 20};
    uint32_t *tab; CK(hipMalloc(&tab, sizes[3] + 64));
    for (uint64_t bytes : sizes) {
        const uint64_t n = 64ull << 20;
        run<0>("atomicAdd u32 no-return", tab, bytes, n);
        run<1>("atomicAdd u32 returning", tab, bytes, n);
        run<2>("CAS byte add (exact)", tab, bytes, n);
        run<5>("atomicAdd wg-scope no-return", tab, bytes, n);
        run<3>("plain store", tab, bytes, n);
        run<4>("plain load", tab, bytes, n);
    }
    return 0;
}
