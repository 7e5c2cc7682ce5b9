// alloc_bench.hip -- what a hipMalloc / hipFree of pool-sized blocks costs on this box (round 6: the cold pass of bench.py
// reported between 1 ms and 5.2 s inside two hipMalloc calls of 10-30 GB).  Pure HIP, no library.
//   hipcc --offload-arch=gfx950 -O2 -o tools/alloc_bench tools/alloc_bench.hip && tools/alloc_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint4 *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4); }
static void info(const char *tag) { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); printf("  [%s] free %.2f of %.2f GB\n", tag, f / 1e9, t / 1e9); }
static void *timed_malloc(size_t bytes, const char *tag)
{
    void *p = nullptr;
    const double t0 = now();
    const hipError_t e = hipMalloc(&p, bytes);
    const double t1 = now();
    printf("hipMalloc %6.2f GB (%s): %9.3f ms%s\n", bytes / 1e9, tag, t1 - t0, e == hipSuccess ? "" : " FAILED");
    return e == hipSuccess ? p : nullptr;
}
static void timed_free(void *p, const char *tag)
{
    const double t0 = now();
    (void)hipFree(p);
    printf("hipFree            (%s): %9.3f ms\n", tag, now() - t0);
}
static void timed_touch(void *p, size_t bytes, const char *tag)
{
    (void)hipDeviceSynchronize();
    const double t0 = now();
    touch<<<2048, 256>>>((uint4 *)p, bytes / 16);
    (void)hipDeviceSynchronize();
    printf("first touch %6.2f GB (%s): %9.3f ms\n", bytes / 1e9, tag, now() - t0);
}
int main()
{
    (void)hipFree(nullptr);
    info("start");
    const size_t GB = 1ull << 30;
    for (size_t gb : {1, 4, 12, 24}) {
        void *p = timed_malloc(gb * GB, "fresh"); if (!p) return 1;
        timed_touch(p, gb * GB, "fresh"); timed_touch(p, gb * GB, "again");
        timed_free(p, "touched");
        p = timed_malloc(gb * GB, "same size again"); timed_touch(p, gb * GB, "same size again"); timed_free(p, "again");
    }
    // the bench's pattern: 80-150 GB of arrays resident (allocated and written by another allocator), then two pools
    printf("-- arrays first: 3 x 30 GB, touched; then 2 x 12 GB\n");
    std::vector<void *> arr;
    for (int i = 0; i < 3; ++i) { void *p = timed_malloc(30 * GB, "array"); if (!p) return 1; timed_touch(p, 30 * GB, "array"); arr.push_back(p); }
    info("arrays resident");
    void *a = timed_malloc(12 * GB, "pool a"), *b = timed_malloc(12 * GB, "pool b");
    timed_touch(a, 12 * GB, "pool a"); timed_free(a, "pool a"); timed_free(b, "pool b");
    printf("-- arrays freed, 2 x 12 GB at once after them\n");
    for (void *p : arr) timed_free(p, "array");
    info("after frees");
    a = timed_malloc(12 * GB, "pool a after frees"); b = timed_malloc(12 * GB, "pool b after frees");
    timed_touch(a, 12 * GB, "pool a after frees");
    timed_free(a, "a"); timed_free(b, "b");
    // a block 200 GB large: does time scale with size?
    a = timed_malloc(200 * GB, "200 GB"); if (a) { timed_touch(a, 200 * GB, "200 GB"); timed_free(a, "200 GB"); }
    a = timed_malloc(200 * GB, "200 GB again"); if (a) timed_free(a, "200 GB again, untouched");
    // stream-ordered pool
    hipStream_t st; (void)hipStreamCreate(&st);
    hipMemPool_t pool; (void)hipDeviceGetDefaultMemPool(&pool, 0);
    uint64_t thr = ~0ull; (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    for (int r = 0; r < 3; ++r) {
        void *p = nullptr; const double t0 = now();
        const hipError_t e = hipMallocAsync(&p, 12 * GB, st); (void)hipStreamSynchronize(st);
        printf("hipMallocAsync 12 GB round %d: %9.3f ms%s\n", r, now() - t0, e == hipSuccess ? "" : " FAILED");
        if (p) { const double t1 = now(); (void)hipFreeAsync(p, st); (void)hipStreamSynchronize(st); printf("hipFreeAsync: %9.3f ms\n", now() - t1); }
    }
    return 0;
}
