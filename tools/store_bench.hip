// tools/store_bench.hip -- what scattered global stores cost the CU's address path: cycles per wave instruction per CU for runs of
// R consecutive dwords per group of R lanes at random (A-byte aligned) places of a large buffer, and for 16-byte stores per lane.
// hipcc --offload-arch=gfx950 -O3 -o tools/store_bench tools/store_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// MODE 0: dword stores, runs of `run` lanes, run base aligned to `align` dwords; MODE 1: dwordx4 per lane, groups of `run` lanes
// contiguous (run * 16 bytes), base aligned to `align` dwords
template <int MODE, int NT>
__global__ __launch_bounds__(512) void k(uint32_t *buf, uint32_t mask, uint32_t run, uint32_t align, uint32_t reps)
{
    const uint32_t lane = threadIdx.x & 63u, g = lane / run, i = lane % run;
    uint32_t h = (blockIdx.x * 512u + (threadIdx.x & ~63u) + g) * 2654435761u + 12345u;
    for (uint32_t r = 0; r < reps; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            const uint32_t base = ((h >> 4) & mask) / align * align;          // in dwords
            if (MODE == 0) { if (NT) __builtin_nontemporal_store(r + j, buf + base + i); else buf[base + i] = r + j; }
            else {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4 v = {r, (uint32_t)j, lane, 0u};
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(buf + base + 4u * i)); else *reinterpret_cast<u32x4 *>(buf + base + 4u * i) = v;
            }
        }
    }
}

template <int MODE, int NT> void run_one_(uint32_t *buf, uint32_t mask, uint32_t run, uint32_t align, const char *what)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const uint32_t reps = 400;
    hipLaunchKernelGGL((k<MODE, NT>), dim3(512), dim3(512), 0, 0, buf, mask, run, align, 4u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(512), dim3(512), 0, 0, buf, mask, run, align, reps);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double winst = 2.0 * 8.0 * reps * 8.0;          // per CU: 2 workgroups x 8 waves x reps x 8
    const double bytes = 512.0 * 512.0 * reps * 8.0 * (MODE ? 16.0 : 4.0);
    printf("%s %-44s %7.1f cycles per wave instruction per CU   %6.0f GB/s   %.3f cycles per dword per CU\n", NT ? "nt   " : "plain", what, ms * 1e-3 * 2.4e9 / winst, bytes / ms / 1e6,
           ms * 1e-3 * 2.4e9 / winst / (64.0 * (MODE ? 4.0 : 1.0)));
}

template <int MODE> void run_one(uint32_t *buf, uint32_t mask, uint32_t run, uint32_t align, const char *what) { run_one_<MODE, 0>(buf, mask, run, align, what); run_one_<MODE, 1>(buf, mask, run, align, what); }

int main()
{
    uint32_t *buf; const size_t n = (size_t)1 << 30;      // 4 GB
    if (hipMalloc(&buf, n * 4 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const uint32_t mask = (uint32_t)(n - 1) & ~1023u;
    run_one<0>(buf, mask, 64, 64, "dword, 64 lanes contiguous, 256-B aligned");
    run_one<0>(buf, mask, 64, 1, "dword, 64 lanes contiguous, 4-B aligned");
    run_one<0>(buf, mask, 32, 1, "dword, runs of 32, 4-B aligned");
    run_one<0>(buf, mask, 16, 16, "dword, runs of 16, 64-B aligned");
    run_one<0>(buf, mask, 16, 1, "dword, runs of 16, 4-B aligned");
    run_one<0>(buf, mask, 8, 8, "dword, runs of 8, 32-B aligned");
    run_one<0>(buf, mask, 8, 1, "dword, runs of 8, 4-B aligned");
    run_one<0>(buf, mask, 4, 4, "dword, runs of 4, 16-B aligned");
    run_one<0>(buf, mask, 4, 1, "dword, runs of 4, 4-B aligned");
    run_one<0>(buf, mask, 1, 1, "dword, every lane its own place");
    run_one<1>(buf, mask, 1, 4, "dwordx4, every lane its own place, 16-B aligned");
    run_one<1>(buf, mask, 1, 1, "dwordx4, every lane its own place, 4-B aligned");
    run_one<1>(buf, mask, 4, 16, "dwordx4, 4 lanes = 64 B, 64-B aligned");
    run_one<1>(buf, mask, 4, 1, "dwordx4, 4 lanes = 64 B, 4-B aligned");
    run_one<1>(buf, mask, 16, 64, "dwordx4, 16 lanes = 256 B, 256-B aligned");
    run_one<1>(buf, mask, 64, 256, "dwordx4, 64 lanes = 1 KB contiguous");
    return 0;
}
