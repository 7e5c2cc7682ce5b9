// tools/load_bench.hip -- microbenchmark of the scan's window load patterns on MI355X (not product code).
// Persistent waves, one 1024-position window of lcp + da (u32) per wave and step, the next window's loads in
// flight while the current one is "consumed" (xor + ballot so nothing is optimised away).  Prints GB/s per pattern.
//   A  lane-strided dword loads (the scan's layout: register j of lane l = position 64 j + l), non-temporal
//   B  lcp: every lane its own 16 consecutive positions as 4 x dwordx4; da as in A
//   C  both arrays as per-lane 4 x dwordx4
//   D  coalesced dwordx4 (lane l = positions 4 l .. 4 l + 3 of a 256-position block)
//   E  A with two windows in flight
//   F  plain grid-stride dwordx4 kernel, no persistence (the chip's read rate)
// build: hipcc -O3 --offload-arch=gfx950 tools/load_bench.hip -o tools/load_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr uint32_t WIN = 1024;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PAT> struct Regs { uint32_t l[16], d[16]; };

template <int PAT>
__device__ __forceinline__ void load_win(Regs<PAT> &r, const uint32_t *lcp, const uint32_t *da, uint64_t lo, uint32_t lane)
{
    if (PAT == 0 || PAT == 4) {
        const uint32_t *lp = lcp + lo + lane, *dp = da + lo + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) { r.l[j] = __builtin_nontemporal_load(lp + 64 * j); r.d[j] = __builtin_nontemporal_load(dp + 64 * j); }
    } else if (PAT == 1) {
        const u32x4 *lp = reinterpret_cast<const u32x4 *>(lcp + lo + 16u * lane);
        const uint32_t *dp = da + lo + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const u32x4 v = __builtin_nontemporal_load(lp + j); r.l[4 * j] = v.x; r.l[4 * j + 1] = v.y; r.l[4 * j + 2] = v.z; r.l[4 * j + 3] = v.w; }
#pragma unroll
        for (int j = 0; j < 16; ++j) r.d[j] = __builtin_nontemporal_load(dp + 64 * j);
    } else if (PAT == 2) {
        const u32x4 *lp = reinterpret_cast<const u32x4 *>(lcp + lo + 16u * lane), *dp = reinterpret_cast<const u32x4 *>(da + lo + 16u * lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 v = __builtin_nontemporal_load(lp + j), w = __builtin_nontemporal_load(dp + j);
            r.l[4 * j] = v.x; r.l[4 * j + 1] = v.y; r.l[4 * j + 2] = v.z; r.l[4 * j + 3] = v.w;
            r.d[4 * j] = w.x; r.d[4 * j + 1] = w.y; r.d[4 * j + 2] = w.z; r.d[4 * j + 3] = w.w;
        }
    } else {
        const u32x4 *lp = reinterpret_cast<const u32x4 *>(lcp + lo) + lane, *dp = reinterpret_cast<const u32x4 *>(da + lo) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 v = __builtin_nontemporal_load(lp + 64 * j), w = __builtin_nontemporal_load(dp + 64 * j);
            r.l[4 * j] = v.x; r.l[4 * j + 1] = v.y; r.l[4 * j + 2] = v.z; r.l[4 * j + 3] = v.w;
            r.d[4 * j] = w.x; r.d[4 * j + 1] = w.y; r.d[4 * j + 2] = w.z; r.d[4 * j + 3] = w.w;
        }
    }
}

template <int PAT> __device__ __forceinline__ uint32_t consume(const Regs<PAT> &r, uint32_t alpha)
{
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc += (uint32_t)__popcll(__ballot(r.l[j] < alpha)); acc ^= r.d[j]; }
    return acc;
}

// LDSKB: static LDS per workgroup, to pin the number of resident workgroups per CU like the scan (48 KB -> 3)
template <int PAT, int LDSKB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_win(const uint32_t *lcp, const uint32_t *da, uint32_t n_win, uint32_t alpha, uint32_t *out)
{
    __shared__ uint32_t pad[LDSKB * 256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (alpha == 0xFFFFFFFFu) pad[threadIdx.x] = lane;            // never: keeps the array
    const uint32_t stride = gridDim.x * 4u;
    uint32_t win = blockIdx.x * 4u + wave, acc = 0;
    if (win >= n_win) return;
    Regs<PAT> a, b;
    load_win<PAT>(a, lcp, da, (uint64_t)win * WIN, lane);
    if (PAT == 4) {
        uint32_t w2 = win + stride;
        if (w2 < n_win) load_win<PAT>(b, lcp, da, (uint64_t)w2 * WIN, lane);
        for (;;) {
            acc += consume<PAT>(a, alpha);
            const uint32_t w3 = w2 + stride;
            if (w2 >= n_win) break;
            if (w3 < n_win) load_win<PAT>(a, lcp, da, (uint64_t)w3 * WIN, lane);
            acc += consume<PAT>(b, alpha);
            const uint32_t w4 = w3 + stride;
            if (w3 >= n_win) break;
            if (w4 < n_win) load_win<PAT>(b, lcp, da, (uint64_t)w4 * WIN, lane);
            w2 = w4;
        }
    } else {
        for (;;) {
            Regs<PAT> cur = a;
            const uint32_t next = win + stride;
            // the scan issues the next window's loads after the current registers have been staged: same here
            uint32_t part = consume<PAT>(cur, alpha);
            if (next < n_win) load_win<PAT>(a, lcp, da, (uint64_t)next * WIN, lane);
            acc += part;
            if (next >= n_win) break;
            win = next;
        }
    }
    if (acc == 0x12345678u || alpha == 0xFFFFFFFFu) out[0] = acc + pad[lane];
}

// pattern A loads + `recs` 8-byte records per window and wave appended to the wave's own region (what the binned scan does):
// SMODE 0 none, 1 nt 8 B per lane (64 records per instruction), 2 plain, 3 nt 16 B per lane (two records per lane), 4 stores
// issued BEFORE the next window's loads instead of after
template <int SMODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_mix(const uint32_t *lcp, const uint32_t *da, uint32_t n_win, uint32_t alpha,
                                                                                         uint64_t *pool, uint32_t cap_w, uint32_t recs, uint32_t *out)
{
    __shared__ uint32_t pad[48 * 256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (alpha == 0xFFFFFFFFu) pad[threadIdx.x] = lane;
    const uint32_t stride = gridDim.x * 4u;
    uint32_t win = blockIdx.x * 4u + wave, acc = 0, out_n = 0;
    if (win >= n_win) return;
    uint64_t *reg = pool + (size_t)(blockIdx.x * 4u + wave) * cap_w;
    Regs<0> a;
    load_win<0>(a, lcp, da, (uint64_t)win * WIN, lane);
    auto stores = [&](uint32_t v) {
        if (SMODE == 3) {
            for (uint32_t k0 = 0; k0 < recs; k0 += 128u) {
                const uint32_t k = k0 + 2u * lane;
                if (k + 1u < recs && out_n + k + 1u < cap_w) {
                    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                    u64x2 r; r.x = ((uint64_t)v << 8) | k; r.y = ((uint64_t)v << 9) | k;
                    __builtin_nontemporal_store(r, reinterpret_cast<u64x2 *>(reg + out_n + k));
                }
            }
        } else {
            for (uint32_t k0 = 0; k0 < recs; k0 += 64u) {
                const uint32_t k = k0 + lane;
                if (k < recs && out_n + k < cap_w) {
                    const uint64_t r = ((uint64_t)v << 8) | k;
                    if (SMODE == 2) reg[out_n + k] = r; else __builtin_nontemporal_store(r, reg + out_n + k);
                }
            }
        }
        out_n += recs;
    };
    uint32_t iter = 0;
    uint64_t *wg_reg = pool + (size_t)blockIdx.x * 4u * cap_w;      // SMODE 6: the workgroup's four regions as one stream
    unsigned long long *cursor = reinterpret_cast<unsigned long long *>(out + 4);
    for (;;) {
        Regs<0> cur = a;
        const uint32_t next = win + stride;
        uint32_t part = consume<0>(cur, alpha);
        if (SMODE == 4) { stores(part); asm volatile("" ::: "memory"); }
        if (next < n_win) load_win<0>(a, lcp, da, (uint64_t)next * WIN, lane);
        if (SMODE == 5) { if ((iter & 3u) == 3u) { stores(part); stores(part); stores(part); stores(part); } }
        else if (SMODE == 6) {
            for (uint32_t k0 = 0; k0 < recs; k0 += 64u) {
                const size_t at = ((size_t)iter * 4u + wave) * recs + k0 + lane;
                if (at < (size_t)4u * cap_w) __builtin_nontemporal_store(((uint64_t)part << 8) | lane, wg_reg + at);
            }
        } else if (SMODE == 7) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(cursor, (unsigned long long)recs);
            base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            for (uint32_t k0 = 0; k0 < recs; k0 += 64u) __builtin_nontemporal_store(((uint64_t)part << 8) | lane, pool + (base + k0 + lane) % ((size_t)gridDim.x * 4u * cap_w));
        } else if (SMODE == 9 || SMODE == 10) {
            const uint32_t ring = cap_w;                       // records in the wave's ring (the launch passes a small cap_w)
            for (uint32_t k0 = 0; k0 < recs; k0 += 64u) {
                const uint32_t at = (out_n + k0 + lane) % ring;
                if (SMODE == 9) __builtin_nontemporal_store(((uint64_t)part << 8) | lane, reg + at); else reg[at] = ((uint64_t)part << 8) | lane;
            }
            out_n += recs;
        } else if (SMODE == 12 || SMODE == 13) {
            // window-indexed slots: window w writes at pool + w * slot -- the band of windows in flight is contiguous, so the
            // stores sweep the pool sequentially like the loads sweep the arrays (12: 8-byte records, 13: 4-byte records)
            if (SMODE == 12) {
                uint64_t *dst = pool + (size_t)win * cap_w;
                for (uint32_t k0 = 0; k0 < recs; k0 += 64u) if (k0 + lane < recs) __builtin_nontemporal_store(((uint64_t)part << 8) | lane, dst + k0 + lane);
            } else {
                uint32_t *dst = reinterpret_cast<uint32_t *>(pool) + (size_t)win * cap_w;
                for (uint32_t k0 = 0; k0 < recs; k0 += 64u) if (k0 + lane < recs) __builtin_nontemporal_store(part + lane, dst + k0 + lane);
            }
        } else if (SMODE == 14) {
            if ((iter & 7u) == 7u) for (int rep = 0; rep < 8; ++rep) stores(part);
        } else if (SMODE == 16) {
            for (uint32_t k0 = 0; k0 < recs; k0 += 64u)
                if (k0 + lane < recs && out_n + k0 + lane < cap_w) __hip_atomic_exchange(reg + out_n + k0 + lane, ((uint64_t)part << 8) | lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out_n += recs;
        } else if (SMODE == 17) {
            // the same stores written in inline assembly: the compiler's s_waitcnt pass does not know they are outstanding and
            // keeps its tight counted waits on the loads (safe: vmcnt(N) can only wait LONGER with more operations in flight)
            for (uint32_t k0 = 0; k0 < recs; k0 += 64u) {
                const uint32_t k = k0 + lane;
                if (k < recs && out_n + k < cap_w) {
                    const uint64_t r = ((uint64_t)part << 8) | k;
                    uint64_t *addr = reg + out_n + k;
                    asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(addr), "v"(r) : "memory");
                }
            }
            out_n += recs;
        } else if (SMODE == 8) {
            uint32_t *r4 = reinterpret_cast<uint32_t *>(reg);
            for (uint32_t k0 = 0; k0 < recs; k0 += 64u) if (out_n + k0 + lane < 2u * cap_w) __builtin_nontemporal_store(part + lane, r4 + out_n + k0 + lane);
            out_n += recs;
        } else
        if (SMODE != 0 && SMODE != 4) stores(part);
        ++iter;
        acc += part;
        if (next >= n_win) break;
        win = next;
    }
    if (acc == 0x12345678u || alpha == 0xFFFFFFFFu) out[0] = acc + pad[lane];
}

// decoupled: waves 0..2 of a workgroup only load (pattern A); wave 3 only stores, at the pace of its siblings (3 x recs records per step)
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_split(const uint32_t *lcp, const uint32_t *da, uint32_t n_win, uint32_t alpha,
                                                                                           uint64_t *pool, uint32_t cap_w, uint32_t recs, uint32_t *out)
{
    __shared__ uint32_t pad[48 * 256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (alpha == 0xFFFFFFFFu) pad[threadIdx.x] = lane;
    const uint32_t stride = gridDim.x * 3u;
    if (wave == 3u) {
        uint64_t *reg = pool + (size_t)blockIdx.x * 4u * cap_w;
        uint32_t out_n = 0;
        for (uint32_t w = blockIdx.x * 3u; w < n_win; w += stride) {
            for (uint32_t k0 = 0; k0 < 3u * recs; k0 += 64u) {
                const uint32_t k = out_n + k0 + lane;
                if (k < 4u * cap_w) { if (NT) __builtin_nontemporal_store(((uint64_t)w << 8) | lane, reg + k); else reg[k] = ((uint64_t)w << 8) | lane; }
            }
            out_n += 3u * recs;
            __builtin_amdgcn_s_sleep(64);
        }
        return;
    }
    uint32_t win = blockIdx.x * 3u + wave, acc = 0;
    if (win >= n_win) return;
    Regs<0> a;
    load_win<0>(a, lcp, da, (uint64_t)win * WIN, lane);
    for (;;) {
        Regs<0> cur = a;
        const uint32_t next = win + stride;
        uint32_t part = consume<0>(cur, alpha);
        if (next < n_win) load_win<0>(a, lcp, da, (uint64_t)next * WIN, lane);
        acc += part;
        if (next >= n_win) break;
        win = next;
    }
    if (acc == 0x12345678u || alpha == 0xFFFFFFFFu) out[0] = acc + pad[lane];
}

__global__ __launch_bounds__(256) void k_plain(const u32x4 *lcp, const u32x4 *da, uint64_t n4, uint32_t *out)
{
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += stride) {
        const u32x4 v = __builtin_nontemporal_load(lcp + i), w = __builtin_nontemporal_load(da + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void k_fill(uint32_t *p, uint64_t n, uint32_t seed)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (uint32_t)(i * 2654435761u + seed) >> 8;
}

template <typename F> static void timeit(const char *name, uint64_t bytes, F launch)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    const int R = 5;
    for (int r = 0; r < R; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; sum += ms;
    }
    printf("%-44s best %7.3f ms  avg %7.3f ms  %7.1f GB/s (best)\n", name, best, sum / R, bytes / best / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000000ull;
    const uint32_t n_win = (uint32_t)(n / WIN);
    const uint64_t nn = (uint64_t)n_win * WIN;
    uint32_t *lcp, *da, *out;
    CK(hipMalloc(&lcp, nn * 4 + 64)); CK(hipMalloc(&da, nn * 4 + 64)); CK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, lcp, nn, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, da, nn, 7u);
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    printf("device %s, %u CUs, N = %llu symbols (%.1f GB read per launch)\n", prop.name, cus, (unsigned long long)nn, nn * 8 / 1e9);
    const uint64_t bytes = nn * 8;
    const dim3 g3(cus * 3), g2(cus * 2), g4(cus * 4), blk(256);
    timeit("A lane-strided dword, 3 WG/CU", bytes, [&] { hipLaunchKernelGGL((k_win<0, 48>), g3, blk, 0, 0, lcp, da, n_win, 16u, out); });
    timeit("B lcp per-lane 4x dwordx4, da strided, 3 WG/CU", bytes, [&] { hipLaunchKernelGGL((k_win<1, 48>), g3, blk, 0, 0, lcp, da, n_win, 16u, out); });
    timeit("C both per-lane 4x dwordx4, 3 WG/CU", bytes, [&] { hipLaunchKernelGGL((k_win<2, 48>), g3, blk, 0, 0, lcp, da, n_win, 16u, out); });
    timeit("D coalesced dwordx4, 3 WG/CU", bytes, [&] { hipLaunchKernelGGL((k_win<3, 48>), g3, blk, 0, 0, lcp, da, n_win, 16u, out); });
    timeit("E lane-strided dword, 2 windows in flight, 3 WG/CU", bytes, [&] { hipLaunchKernelGGL((k_win<4, 48>), g3, blk, 0, 0, lcp, da, n_win, 16u, out); });
    timeit("A lane-strided dword, 4 WG/CU", bytes, [&] { hipLaunchKernelGGL((k_win<0, 36>), g4, blk, 0, 0, lcp, da, n_win, 16u, out); });
    timeit("A lane-strided dword, 2 WG/CU", bytes, [&] { hipLaunchKernelGGL((k_win<0, 72>), g2, blk, 0, 0, lcp, da, n_win, 16u, out); });
    timeit("C both per-lane 4x dwordx4, 4 WG/CU", bytes, [&] { hipLaunchKernelGGL((k_win<2, 36>), g4, blk, 0, 0, lcp, da, n_win, 16u, out); });
    timeit("E 2 windows in flight, 2 WG/CU", bytes, [&] { hipLaunchKernelGGL((k_win<4, 72>), g2, blk, 0, 0, lcp, da, n_win, 16u, out); });
    {
        const uint32_t recs = 128u, n_waves = cus * 3 * 4, cap_w = ((n_win / n_waves + 2) * recs + 15u) & ~15u;
        uint64_t *pool; CK(hipMalloc(&pool, (size_t)n_waves * cap_w * 8 + 64));
        const uint64_t b2 = bytes + (uint64_t)n_win * recs * 8;
        printf("record stores: %u per window and wave (%.2f GB per launch), region %u records\n", recs, n_win * (double)recs * 8 / 1e9, cap_w);
        timeit("A + no stores", bytes, [&] { hipLaunchKernelGGL((k_mix<0>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + nt 8-B record stores (bytes incl. stores)", b2, [&] { hipLaunchKernelGGL((k_mix<1>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + plain 8-B record stores", b2, [&] { hipLaunchKernelGGL((k_mix<2>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + nt 16-B-per-lane record stores", b2, [&] { hipLaunchKernelGGL((k_mix<3>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + nt 8-B stores issued before the loads", b2, [&] { hipLaunchKernelGGL((k_mix<4>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + 4 KB bursts every 4th window", b2, [&] { hipLaunchKernelGGL((k_mix<5>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + one stream per workgroup (4 x 1 KB adjacent)", b2, [&] { hipLaunchKernelGGL((k_mix<6>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + ONE global stream (atomic cursor)", b2, [&] { CK(hipMemsetAsync(out, 0, 64)); hipLaunchKernelGGL((k_mix<7>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        for (uint32_t ring : {512u, 2048u, 8192u}) {
            char nm[128];
            snprintf(nm, sizeof nm, "A + nt stores into a %u-record ring per wave (%.0f MB pool)", ring, n_waves * (double)ring * 8 / 1e6);
            timeit(nm, b2, [&] { hipLaunchKernelGGL((k_mix<9>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, ring, recs, out); });
            snprintf(nm, sizeof nm, "A + plain stores into a %u-record ring per wave (%.0f MB pool)", ring, n_waves * (double)ring * 8 / 1e6);
            timeit(nm, b2, [&] { hipLaunchKernelGGL((k_mix<10>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, ring, recs, out); });
        }
        timeit("split: 3 loading waves + 1 storing wave per workgroup, no stores", bytes, [&] { hipLaunchKernelGGL((k_split<1>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 0u, out); });
        timeit("split: 3 loading waves + 1 storing wave (nt, 1 GB)", b2, [&] { hipLaunchKernelGGL((k_split<1>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("split: 3 loading waves + 1 storing wave (plain, 1 GB)", b2, [&] { hipLaunchKernelGGL((k_split<0>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        {
            uint64_t *wpool; CK(hipMalloc(&wpool, (size_t)n_win * 256 * 8 + 64));
            timeit("A + window-indexed slots of 128 x 8 B (dense)", b2, [&] { hipLaunchKernelGGL((k_mix<12>), g3, blk, 0, 0, lcp, da, n_win, 16u, wpool, 128u, recs, out); });
            timeit("A + window-indexed slots of 256 x 8 B (half used)", b2, [&] { hipLaunchKernelGGL((k_mix<12>), g3, blk, 0, 0, lcp, da, n_win, 16u, wpool, 256u, recs, out); });
            timeit("A + window-indexed slots of 128 x 4 B (dense, 0.5 GB)", bytes + (uint64_t)n_win * recs * 4, [&] { hipLaunchKernelGGL((k_mix<13>), g3, blk, 0, 0, lcp, da, n_win, 16u, wpool, 128u, recs, out); });
            timeit("A + window-indexed slots of 256 x 4 B (half used, 0.5 GB)", bytes + (uint64_t)n_win * recs * 4, [&] { hipLaunchKernelGGL((k_mix<13>), g3, blk, 0, 0, lcp, da, n_win, 16u, wpool, 256u, recs, out); });
            CK(hipFree(wpool));
        }
        timeit("A + 4-byte records (0.5 GB)", bytes + (uint64_t)n_win * recs * 4, [&] { hipLaunchKernelGGL((k_mix<8>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + 64 records per window (0.5 GB)", bytes + (uint64_t)n_win * 64 * 8, [&] { hipLaunchKernelGGL((k_mix<1>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 64u, out); });
        timeit("A + 8 records per window (62 MB)", bytes + (uint64_t)n_win * 8 * 8, [&] { hipLaunchKernelGGL((k_mix<1>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 8u, out); });
        timeit("A + 1 record per window (8 MB)", bytes + (uint64_t)n_win * 8, [&] { hipLaunchKernelGGL((k_mix<1>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 1u, out); });
        timeit("A + 8 records per window, plain stores", bytes + (uint64_t)n_win * 8 * 8, [&] { hipLaunchKernelGGL((k_mix<2>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 8u, out); });
        timeit("A + 64 records every 8th window (same 62 MB)", bytes + (uint64_t)n_win * 8 * 8, [&] { hipLaunchKernelGGL((k_mix<14>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 8u, out); });
        timeit("A + 8 records per window as atomic exchanges", bytes + (uint64_t)n_win * 8 * 8, [&] { hipLaunchKernelGGL((k_mix<16>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 8u, out); });
        timeit("A + 128 records per window, stores every 8th window (1 GB)", b2, [&] { hipLaunchKernelGGL((k_mix<14>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + 8 records per window, asm stores", bytes + (uint64_t)n_win * 8 * 8, [&] { hipLaunchKernelGGL((k_mix<17>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 8u, out); });
        timeit("A + 64 records per window, asm stores (0.5 GB)", bytes + (uint64_t)n_win * 64 * 8, [&] { hipLaunchKernelGGL((k_mix<17>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 64u, out); });
        timeit("A + 128 records per window, asm stores (1 GB)", b2, [&] { hipLaunchKernelGGL((k_mix<17>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, recs, out); });
        timeit("A + 32 records per window (0.25 GB)", bytes + (uint64_t)n_win * 32 * 8, [&] { hipLaunchKernelGGL((k_mix<1>), g3, blk, 0, 0, lcp, da, n_win, 16u, pool, cap_w, 32u, out); });
        CK(hipFree(pool));
    }
    timeit("F plain grid-stride dwordx4 (16384 WGs)", bytes, [&] { hipLaunchKernelGGL(k_plain, dim3(16384), blk, 0, 0, (const u32x4 *)lcp, (const u32x4 *)da, nn / 4, out); });
    timeit("F plain grid-stride dwordx4 (2048 WGs)", bytes, [&] { hipLaunchKernelGGL(k_plain, dim3(2048), blk, 0, 0, (const u32x4 *)lcp, (const u32x4 *)da, nn / 4, out); });
    return 0;
}
