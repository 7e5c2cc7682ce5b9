// lime_api.cpp -- implementation of the C ABI in include/lime_hip.h on top of the HIP kernels
// in lime_kernels.hip.  Host side only: argument checks, scratch management in HBM, kernel
// sequencing on the caller's stream, staging for the host-pointer entry points.
// There is no CPU code path for the computation: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdarg.h>
#include <stdio.h>
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sched.h>
#include <sys/stat.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "lime_device.h"
#include "lime_hip.h"
#include "lime_kernels.h"

using namespace lime;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? LIME_ERR_NOMEM : LIME_ERR_HIP,            \
                        "%s: %s", #expr, hipGetErrorString(e_));                              \
    } while (0)

struct lime_ctx {
    int device = 0;
    DevStats *d_stats = nullptr;                // followed by the sticky word: passes with a pool overflow not settled by lime_get_stats
    uint32_t *d_sticky = nullptr;
    // clusterChoose's scratch, kept between calls (round 5: per call four hipMalloc / hipFree pairs, two 4 MB copies into pageable vectors and 10^6
    // float divisions were 1.3 ms of configs[2]'s 4.3 ms lime_fused_choose_dev): device words for the rows' max / non-zero counts (+ the
    // table-free finish's region words), pinned host words where they land
    uint8_t *d_choose = nullptr; size_t choose_cap = 0;
    void *h_choose = nullptr; size_t h_choose_cap = 0;
    void *h_stats = nullptr;                    // pinned: where read_stats lands the counters (a copy into pageable memory is staged by the runtime: +30 us per call)
    unsigned long long *d_total = nullptr;
    // per-tile scratch (capacity in tiles)
    size_t tile_cap = 0;
    TileSummary *d_summ = nullptr;
    uint32_t *d_tile_cnt = nullptr;
    uint64_t *d_tile_off = nullptr;
    CrossRec *d_cross = nullptr;
    WinMasks *d_wmask = nullptr; size_t wmask_cap = 0;
    // cluster lists
    lime_cluster_t *d_small = nullptr; uint32_t small_cap = 0;
    lime_cluster_t *d_big = nullptr; uint32_t big_cap = 0;
    lime_cluster_t *d_out = nullptr; size_t out_cap = 0;
    uint32_t *d_big_scratch = nullptr;
    uint32_t max_blocks = 0;                // persistent grid of the scan kernel; 0 = as many workgroups as fit the device (LIME_MAX_BLOCKS)
    uint32_t list_blocks = 8192;
    int ablate = 0;                         // LIME_ABLATE (only in a -DLIME_ABLATE_BUILD library): kernel timing experiments, results invalid when != 0
    // binned table updates (bin-then-apply): record pool, per-bin counters, binned records
    uint32_t *d_pool = nullptr; size_t pool_cap = 0;          // 32-bit records (n_waves x n_sub x cap_w)
    uint32_t *d_recs = nullptr; size_t recs_cap = 0;
    uint32_t *d_wave_cnt = nullptr; size_t wave_cap = 0;
    uint32_t *d_counts = nullptr; size_t counts_cap = 0;
    uint32_t *d_totals = nullptr; uint64_t *d_binbase = nullptr;
    uint64_t *d_regbase = nullptr; size_t regbase_cap = 0;
    uint32_t *d_tbase = nullptr;            // second level by tiles: tiles before each bin, and the tiles' region index
    uint16_t *d_tidx = nullptr; size_t tidx_cap = 0;
    bool by_tiles = true;                   // LIME_SECOND_LEVEL=sweeps: k_part2 + k_apply instead (comparison runs)
    // owner-partitioned exchange: the long clusters' update records of this rank; the owner's regrouped records
    uint64_t *d_bigrec = nullptr; uint32_t *d_bigrec_n = nullptr; uint32_t bigrec_cap = 0;
    uint32_t *d_xrecs = nullptr, *d_xrecs2 = nullptr; size_t xrecs_cap = 0;
    uint64_t *d_xoff = nullptr; size_t xoff_cap = 0; uint64_t *d_xreg = nullptr; size_t xreg_cap = 0;
    uint64_t *h_xoff = nullptr; size_t h_xoff_cap = 0; hipEvent_t ev_xoff = nullptr; bool ev_xoff_pending = false;   // pinned staging of the offsets lime_apply_records_dev uploads (no stream synchronisation in an exchange step)
    uint32_t rec_n_bins = 0, rec_bin_shift = 0;                // layout of the records the last lime_fused_records_dev left
    int upd_pref = -1;                      // LIME_UPDATE_PATH: -1 auto, 0 compare-and-swap on the table, 1 binned
    bool density_known = false; double density = 0.0;          // table updates per owned symbol of the last pass read back
    bool bin_levels_forced = false;
    uint32_t bin_one_level = BIN_ONE_LEVEL, bin_two_level = BIN_TWO_LEVEL;   // LIME_BIN_LEVELS="a,b" (tests: force the second level on small tables)
    double pool_density = 0.45;             // records per owned symbol the pool is sized for before anything has been measured (first passes below 2^28 symbols, which run without the density probe: text has 0.24 .. 0.39; grows on LIME_FLAG_POOL_FULL)
    bool pool_density_fixed = false;        // set by LIME_POOL_DENSITY or by a repeated pass: sizing_density() then leaves it alone
    int scan_static_pct = -1;               // share (%) of the scan's rounds of window chunks that go round-robin, the rest is handed out as workgroups get there; -1: by the input's length (base_args); LIME_SCAN_STATIC_PCT: tests, comparison runs
    uint32_t part_split = 2;                // producers (of k_part) per scan workgroup at most (LIME_PART_SPLIT: comparison runs): two = one partition workgroup per resident slot of the device; four -- round 4's first choice -- cut the streams into more, less filled tiles: k_part_lines +4 % at N = 1e10 and on the text workload
    uint32_t pool_slack = 512;              // + this many records per wave and sub-region (LIME_POOL_SLACK: tests make pools overflow)
    uint64_t probe_min = 1ull << 28;        // first passes of fewer symbols run without the density probe (binned, pool for 0.45 records per symbol); LIME_PROBE_MIN: tests
    bool probe = true;                      // LIME_NO_PROBE: no sampled density probe in front of a ctx's first pass (tests, comparison runs)
    bool force_p64 = false;                 // LIME_FORCE_P64: the partition kernels' 64-bit-position variants on any pass (tests)
    uint64_t p64_test_base = 0;             // LIME_P64_TEST_BASE (tests): the binned records' positions start at this number instead of 0 -- the bin bases are
                                            // shifted by it and the kernels get the records' array address minus it --, so that a small pass crosses a multiple of 2^32
    double alloc_ms = 0.0, probe_ms = 0.0; uint32_t n_probes = 0, n_repeats = 0, n_fallbacks = 0, n_table_free = 0;   // host-side costs a cold pass pays (lime_get_host_times)
    struct Last {                           // the last lime_fused_dev call, so that lime_get_stats can repeat it with a larger pool
        bool valid = false, binned = false;
        const uint32_t *lcp = nullptr, *da = nullptr; const uint8_t *ebwt = nullptr;
        uint64_t n_own = 0, n_avail = 0; int eof = 0; uint32_t n_reads = 0, n_refs = 0, alpha = 0;
        uint8_t *sim = nullptr; int zero_sim = 0; hipStream_t st = nullptr; uint32_t n_waves = 0; bool records_only = false;
        uint64_t own_total = 0;             // owned symbols the counters in d_stats stand for (chunks of a stream accumulate)
        double share = 1.0;                 // binned: the part of a wave's records one sub-region was sized for (sub_share)
        bool fell_back = false;             // the binned path was wanted and could not be had (memory): LIME_FLAG_CAS_FALLBACK
    } last;
    // knobs of lime_set_option that have no other home (all -1 / 0 / false = the library's own choice)
    int apply_wide = -1, sort_nt = -1, part_lines = -1;     // which variant of k_apply_tiles / of k_sort_tiles' row stores / whether k_part_lines may run
    int choose_free = -1;                   // lime_fused_choose_dev: 1 = without the table wherever the layout has a second level, 0 = never
    bool no_staging = false, force_staging = false, force_rccl = false, debug_stats = false;
    uint64_t detect_chunk = 0, score_chunk = 0;             // symbols per chunk of lime_detect / lime_score* walks (0: by the sources)
    bool no_direct = false;                 // binned updates through the update queue (k_scan<., 0, 1>) even for tables of one or two sub-regions (option no_direct: comparison runs, tests)
    uint32_t dense_min = 64;                // k_scan: windows with more accepted clusters list their 2-symbol clusters apart (option dense_min; tests: 0 = every window)
    int io_threads = 0;                     // host threads that stage pageable sources into the pinned ring (0: 8, at most the CPUs this process may use)
    // timing with HIP events on the launch stream: per pass {pass start, scan start, scan end, pass end}
    bool timing = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
};

extern "C" const char *lime_last_error(void) { return g_err.c_str(); }
extern "C" const char *lime_version(void) { return "lime_amd 0.1 (gfx950)"; }
extern "C" void lime_free(void *p) { free(p); }
extern "C" int lime_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
// the device with the most free memory; devices within 1 GiB of the best count as equally free and `salt`
// (a pid) picks among them, so that processes started together do not all land on device 0
extern "C" int lime_pick_device(unsigned salt)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 1) return 0;
    std::vector<size_t> fr(n, 0);
    size_t best = 0;
    for (int d = 0; d < n; ++d) {
        size_t f = 0, t = 0;
        if (hipSetDevice(d) == hipSuccess && hipMemGetInfo(&f, &t) == hipSuccess) fr[d] = f;
        if (fr[d] > best) best = fr[d];
    }
    std::vector<int> cand;
    for (int d = 0; d < n; ++d) if (fr[d] + ((size_t)1 << 30) >= best) cand.push_back(d);
    return cand.empty() ? 0 : cand[salt % cand.size()];
}

extern "C" size_t lime_sim_bytes(uint32_t n_reads, uint32_t n_refs)
{
    size_t b = (size_t)n_reads * n_refs;
    return (b + 15u) & ~(size_t)15u;
}

static std::atomic<bool> g_debug_alloc{false}, g_poison_cache{false};   // "poison_cache" (on under LIME_TEST_HOOKS): a block taken from the cache is filled with 0xA5 -- nothing may rely on what a fresh allocation holds          // lime_set_option "debug_alloc": every device allocation of the library on stderr

// ---- device blocks released by the contexts of this process, kept for the next one -------------------------------------------------
// Why (measured, tools/alloc_bench.hip, round 6): a hipMalloc on this platform costs 0.04-0.3 ms whatever its size AS LONG AS the driver hands
// out pages that were never used since they were last cleared; pages that this or another process has written and freed are cleared by the
// driver INSIDE the hipMalloc that gets them, at 25-35 GB/s -- 2.0 s for 25.8 GB allocated right after the same 25.8 GB were freed, 3.8 s
// for 12.9 GB after 3 x 32 GB were freed, 6.0 s for 215 GB -- and the allocator does not prefer clean pages.  That is the "1 ms to 5.2 s"
// of round 5's cold passes (bench.py frees 80-150 GB of arrays between workloads), and a pool that was re-grown by 0.02 % -- free 15.6 GB,
// allocate 15.6 GB -- paid 4.9 s for it.  So the library never gives a large block back while the process lives: lime_shutdown and regrow put
// blocks of 64 MB and more here, the next request takes the smallest cached block that is large enough (and at most twice as large), and
// lime_trim_cache() -- or a failed hipMalloc -- returns them to the driver.  At most a quarter of the device's memory is held.
namespace {
struct BlockCache {
    struct B { void *p; size_t bytes; int dev; };
    static constexpr size_t MIN = (size_t)64 << 20;
    std::mutex mu;
    std::vector<B> idle, out;                               // cached blocks; blocks handed out (their true sizes)
    // lime_reserve: blocks taken from the driver ONCE (at process start, where the time shows as what it is), from which the large buffers of every
    // context are carved afterwards: first fit over the free extents (offset-sorted, merged on return), 2 MB granules
    struct Ext { size_t off, bytes; };
    struct Arena { char *base; size_t bytes; int dev; std::vector<Ext> free_; size_t live; };
    struct Carve { void *p; size_t arena, off, bytes; };
    static constexpr size_t GRAN = (size_t)2 << 20;
    std::vector<Arena> arenas;
    std::vector<Carve> carved;
    bool add_arena(int dev, void *base, size_t bytes)
    {
        std::lock_guard<std::mutex> g(mu);
        arenas.push_back(Arena{static_cast<char *>(base), bytes, dev, {Ext{0, bytes}}, 0});
        return true;
    }
    void *carve(int dev, size_t bytes, size_t *got)
    {
        const size_t want = (bytes + GRAN - 1) / GRAN * GRAN;
        std::lock_guard<std::mutex> g(mu);
        for (size_t ai = 0; ai < arenas.size(); ++ai) {
            Arena &A = arenas[ai];
            if (A.dev != dev) continue;
            for (size_t i = 0; i < A.free_.size(); ++i)
                if (A.free_[i].bytes >= want) {
                    const size_t off = A.free_[i].off;
                    if (A.free_[i].bytes == want) A.free_.erase(A.free_.begin() + (long)i);
                    else { A.free_[i].off += want; A.free_[i].bytes -= want; }
                    A.live += want;
                    carved.push_back(Carve{A.base + off, ai, off, want});
                    *got = want;
                    return A.base + off;
                }
        }
        return nullptr;
    }
    bool uncarve(void *p)                                   // true: p was a piece of an arena and is free again
    {
        std::lock_guard<std::mutex> g(mu);
        for (size_t ci = 0; ci < carved.size(); ++ci)
            if (carved[ci].p == p) {
                const Carve c = carved[ci];
                carved.erase(carved.begin() + (long)ci);
                Arena &A = arenas[c.arena];
                A.live -= c.bytes;
                size_t i = 0;
                while (i < A.free_.size() && A.free_[i].off < c.off) ++i;
                A.free_.insert(A.free_.begin() + (long)i, Ext{c.off, c.bytes});
                if (i + 1 < A.free_.size() && A.free_[i].off + A.free_[i].bytes == A.free_[i + 1].off) { A.free_[i].bytes += A.free_[i + 1].bytes; A.free_.erase(A.free_.begin() + (long)i + 1); }
                if (i > 0 && A.free_[i - 1].off + A.free_[i - 1].bytes == A.free_[i].off) { A.free_[i - 1].bytes += A.free_[i].bytes; A.free_.erase(A.free_.begin() + (long)i); }
                return true;
            }
        return false;
    }
    void *take(int dev, size_t bytes, size_t *got)
    {
        std::lock_guard<std::mutex> g(mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < idle.size(); ++i)
            if (idle[i].dev == dev && idle[i].bytes >= bytes && idle[i].bytes / 2 <= bytes && (best == (size_t)-1 || idle[i].bytes < idle[best].bytes)) best = i;
        if (best == (size_t)-1) return nullptr;
        const B b = idle[best];
        idle.erase(idle.begin() + (long)best);
        out.push_back(b);
        *got = b.bytes;
        return b.p;
    }
    bool tracked(const void *p)
    {
        std::lock_guard<std::mutex> g(mu);
        for (const B &b : out) if (b.p == p) return true;
        for (const Carve &c : carved) if (c.p == p) return true;
        return false;
    }
    void note(int dev, void *p, size_t bytes) { if (bytes >= MIN) { std::lock_guard<std::mutex> g(mu); out.push_back(B{p, bytes, dev}); } }
    // true: the cache keeps p; false: the caller frees it
    bool give(void *p)
    {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < out.size(); ++i)
            if (out[i].p == p) {
                const B b = out[i];
                out.erase(out.begin() + (long)i);
                size_t held = b.bytes, total = 0, fr = 0;
                for (const B &x : idle) if (x.dev == b.dev) held += x.bytes;
                if (hipMemGetInfo(&fr, &total) != hipSuccess || held > total / 4) return false;
                idle.push_back(b);
                return true;
            }
        return false;
    }
    size_t trim(int dev)                                    // dev < 0: every device
    {
        std::vector<B> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < idle.size();) if (dev < 0 || idle[i].dev == dev) { drop.push_back(idle[i]); idle.erase(idle.begin() + (long)i); } else ++i;
        }
        {
            // (arenas nothing is carved from any more go too; the indices of the others stay what the carve records hold: emptied in place)
            std::lock_guard<std::mutex> g(mu);
            for (Arena &A : arenas)
                if (A.base && A.live == 0 && (dev < 0 || A.dev == dev)) { drop.push_back(B{A.base, A.bytes, A.dev}); A.base = nullptr; A.bytes = 0; A.free_.clear(); }
        }
        int cur = 0; (void)hipGetDevice(&cur);
        size_t bytes = 0;
        for (const B &b : drop) { (void)hipSetDevice(b.dev); (void)hipFree(b.p); bytes += b.bytes; }
        if (!drop.empty()) (void)hipSetDevice(cur);
        return bytes;
    }
};
BlockCache g_blocks;
}
extern "C" size_t lime_trim_cache(void) { return g_blocks.trim(-1); }
extern "C" int lime_reserve(size_t bytes)
{
    if (!bytes) return LIME_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return LIME_ERR_HIP;
    void *p = nullptr;
    const size_t want = (bytes + BlockCache::GRAN - 1) / BlockCache::GRAN * BlockCache::GRAN;
    if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return LIME_ERR_NOMEM; }
    g_blocks.add_arena(dev, p, want);
    return LIME_OK;
}

static void dev_release(void *p)
{
    if (!p) return;
    // (hipFree waits for the device before a block goes back; a block that goes to the cache instead can be handed out again at once, so it waits the
    // same way: on the success paths everything that used the block has been waited for anyway, on an error path work on it may still be queued)
    if (g_blocks.tracked(p)) (void)hipDeviceSynchronize();
    if (g_blocks.uncarve(p)) return;
    if (!g_blocks.give(p)) (void)hipFree(p);
}
static hipError_t dev_acquire(void **p, size_t bytes)
{
    int dev = 0; (void)hipGetDevice(&dev);
    size_t got = 0;
    if (bytes >= BlockCache::MIN && ((*p = g_blocks.take(dev, bytes, &got)) || (*p = g_blocks.carve(dev, bytes, &got)))) {
        if (g_poison_cache.load(std::memory_order_relaxed)) { (void)hipMemset(*p, 0xA5, got); (void)hipDeviceSynchronize(); }
        return hipSuccess;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && g_blocks.trim(dev)) { (void)hipGetLastError(); e = hipMalloc(p, bytes); }
    if (e == hipSuccess) g_blocks.note(dev, *p, bytes);
    return e;
}
// ---- options: every tuning / test knob of the library, by name.  Nothing here changes a result; most change which kernels run.
static int set_option(lime_ctx *c, const char *key, const char *s)
{
    if (!key || !s) return fail(LIME_ERR_ARG, "lime_set_option: NULL argument");
    auto is = [&](const char *k) { return !strcmp(key, k); };
    const long v = atol(s);
    if (is("update_path")) c->upd_pref = !strcmp(s, "cas") ? 0 : !strcmp(s, "bin") ? 1 : -1;
    else if (is("bin_levels")) {
        unsigned a1 = 0, a2 = 0;
        if (!*s) { c->bin_one_level = BIN_ONE_LEVEL; c->bin_two_level = BIN_TWO_LEVEL; c->bin_levels_forced = false; }
        else if (sscanf(s, "%u,%u", &a1, &a2) == 2 && a1 >= 1 && a2 >= 1 && a1 <= BIN_MAX && a2 <= BIN_MAX) { c->bin_one_level = a1; c->bin_two_level = a2; c->bin_levels_forced = true; }
        else return fail(LIME_ERR_ARG, "bin_levels: want \"one,two\" with 1 <= one, two <= %u", BIN_MAX);
    }
    else if (is("pool_density")) { const double d = atof(s); if (d > 0) { c->pool_density = d; c->pool_density_fixed = true; } }      // tests: force a small pool
    else if (is("scan_static_pct")) c->scan_static_pct = v >= 0 && v <= 100 ? (int)v : -1;
    else if (is("second_level")) c->by_tiles = strcmp(s, "sweeps") != 0;
    else if (is("part_split")) { if (v >= 1 && v <= 16) c->part_split = (uint32_t)v; }
    else if (is("pool_slack")) { if (v >= 0) c->pool_slack = (uint32_t)v; }
    else if (is("no_probe")) c->probe = v == 0;
    else if (is("probe_min")) { const unsigned long long u = strtoull(s, nullptr, 0); c->probe_min = u < (1ull << 24) ? (1ull << 24) : u; }
    else if (is("force_p64")) c->force_p64 = v != 0;
    else if (is("p64_test_base")) { c->p64_test_base = strtoull(s, nullptr, 0) & ~15ull; if (c->p64_test_base) c->force_p64 = true; }
    else if (is("max_blocks")) c->max_blocks = v > 0 ? (uint32_t)v : 0u;
    else if (is("choose_free")) c->choose_free = *s ? (v != 0) : -1;
    else if (is("apply_wide")) c->apply_wide = *s ? (v != 0) : -1;
    else if (is("sort_nt")) c->sort_nt = *s ? (v != 0) : -1;
    else if (is("part_lines")) c->part_lines = *s ? (v != 0) : -1;
    else if (is("no_staging")) c->no_staging = v != 0 || !*s;
    else if (is("force_staging")) c->force_staging = v != 0 || !*s;
    else if (is("detect_chunk")) c->detect_chunk = strtoull(s, nullptr, 10);
    else if (is("score_chunk")) c->score_chunk = strtoull(s, nullptr, 10);
    else if (is("force_rccl")) c->force_rccl = v != 0 || !*s;
    else if (is("debug_stats")) c->debug_stats = v != 0 || !*s;
    else if (is("apply_group")) set_apply_group((uint32_t)v);
    else if (is("debug_alloc")) g_debug_alloc.store(v != 0 || !*s, std::memory_order_relaxed);
    else if (is("poison_cache")) g_poison_cache.store(v != 0 || !*s, std::memory_order_relaxed);
    else if (is("no_direct")) c->no_direct = v != 0 || !*s;
    else if (is("dense_min")) c->dense_min = *s ? (uint32_t)strtoul(s, nullptr, 0) : 64u;
    else if (is("io_threads")) c->io_threads = v >= 1 ? (v > 64 ? 64 : (int)v) : 0;
    else return fail(LIME_ERR_ARG, "lime_set_option: unknown option \"%s\"", key);
    return LIME_OK;
}
extern "C" int lime_set_option(lime_ctx *c, const char *key, const char *value)
{
    if (!c) return fail(LIME_ERR_ARG, "lime_set_option: ctx is NULL");
    return set_option(c, key, value);
}

extern "C" int lime_init(int device, lime_ctx **out)
{
    if (!out) return fail(LIME_ERR_ARG, "lime_init: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(LIME_ERR_HIP, "lime_init: no HIP device (%s); this library has no CPU path",
                    e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    lime_ctx *c = new (std::nothrow) lime_ctx();
    if (!c) return fail(LIME_ERR_NOMEM, "lime_init: out of host memory");
    HIP_TRY(hipGetDevice(&c->device));
    HIP_TRY(hipMalloc(&c->d_stats, sizeof(DevStats) + 16));
    c->d_sticky = reinterpret_cast<uint32_t *>(c->d_stats + 1);
    HIP_TRY(hipMalloc(&c->d_total, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->d_stats, 0, sizeof(DevStats) + 16));
    HIP_TRY(hipHostMalloc(&c->h_stats, sizeof(lime_stats_t) + 16));
    launch_preload();
#ifdef LIME_ABLATE_BUILD
    if (const char *s = getenv("LIME_ABLATE")) c->ablate = atoi(s);
#endif
    // The environment is read HERE and nowhere else, and only two kinds of variable: LIME_IO_THREADS (a resource limit the user sets, like the
    // reference's `threads` argument) always; the tuning / test knobs of lime_set_option only when LIME_TEST_HOOKS=1 says that this process is a
    // test or an experiment (tests/conftest.py sets it; bench.py refuses to run with it).  A release run takes no hidden switch.
    if (const char *s = getenv("LIME_IO_THREADS")) (void)set_option(c, "io_threads", s);
    if (const char *h = getenv("LIME_TEST_HOOKS")) if (atoi(h) != 0) {
        g_poison_cache.store(true, std::memory_order_relaxed);
        static const char *const hooks[][2] = {
            {"LIME_UPDATE_PATH", "update_path"}, {"LIME_BIN_LEVELS", "bin_levels"}, {"LIME_POOL_DENSITY", "pool_density"}, {"LIME_SCAN_STATIC_PCT", "scan_static_pct"},
            {"LIME_SECOND_LEVEL", "second_level"}, {"LIME_PART_SPLIT", "part_split"}, {"LIME_POOL_SLACK", "pool_slack"}, {"LIME_NO_PROBE", "no_probe"},
            {"LIME_PROBE_MIN", "probe_min"}, {"LIME_FORCE_P64", "force_p64"}, {"LIME_P64_TEST_BASE", "p64_test_base"}, {"LIME_MAX_BLOCKS", "max_blocks"},
            {"LIME_CHOOSE_FREE", "choose_free"}, {"LIME_APPLY_WIDE", "apply_wide"}, {"LIME_SORT_NT", "sort_nt"}, {"LIME_PART_LINES", "part_lines"},
            {"LIME_NO_STAGING", "no_staging"}, {"LIME_FORCE_STAGING", "force_staging"}, {"LIME_DETECT_CHUNK", "detect_chunk"}, {"LIME_SCORE_CHUNK", "score_chunk"},
            {"LIME_FORCE_RCCL", "force_rccl"}, {"LIME_DENSE_MIN", "dense_min"}, {"LIME_NO_DIRECT", "no_direct"}, {"LIME_DEBUG_STATS", "debug_stats"}, {"LIME_DEBUG_ALLOC", "debug_alloc"}, {"LIME_APPLY_GROUP", "apply_group"}};
        for (const auto &hk : hooks)
            if (const char *s = getenv(hk[0])) {
                const int rc = set_option(c, hk[1], s);
                if (rc) { lime_shutdown(c); return fail(rc, "lime_init: %s=%s: %s", hk[0], s, g_err.c_str()); }
            }
    }
    *out = c;
    return LIME_OK;
}

extern "C" void lime_shutdown(lime_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->h_choose) (void)hipHostFree(c->h_choose);
    for (void *p : {(void *)c->d_choose, (void *)c->d_stats, (void *)c->d_total, (void *)c->d_summ, (void *)c->d_tile_cnt, (void *)c->d_tile_off, (void *)c->d_cross,
                    (void *)c->d_wmask, (void *)c->d_small, (void *)c->d_big, (void *)c->d_out, (void *)c->d_big_scratch, (void *)c->d_pool, (void *)c->d_recs,
                    (void *)c->d_wave_cnt, (void *)c->d_counts, (void *)c->d_totals, (void *)c->d_binbase, (void *)c->d_regbase, (void *)c->d_tbase, (void *)c->d_tidx})
        dev_release(p);                                  // (large blocks stay in the process's cache: BlockCache)
    if (c->h_xoff) (void)hipHostFree(c->h_xoff);
    if (c->ev_xoff) (void)hipEventDestroy(c->ev_xoff);
    for (void *p : {(void *)c->d_bigrec, (void *)c->d_bigrec_n, (void *)c->d_xrecs, (void *)c->d_xrecs2, (void *)c->d_xoff, (void *)c->d_xreg}) dev_release(p);
    delete c;
}

static int d2h_pageable(const lime_ctx *c, void *dst, const void *d_src, size_t bytes, hipStream_t st);    // large results into the caller's pageable memory: staged by this library's threads
static thread_local double g_alloc_ms = 0.0;             // (host time spent in hipFree / hipMalloc by regrow: moved into the ctx's account by its callers)
template <typename T> static int regrow(T *&p, size_t count)
{
    const auto t0 = std::chrono::steady_clock::now();
    size_t f0 = 0, f1 = 0, tot = 0;
    const bool dbg = g_debug_alloc.load(std::memory_order_relaxed);
    if (dbg) (void)hipMemGetInfo(&f0, &tot);
    const bool had = p != nullptr;
    if (p) { dev_release(p); p = nullptr; }
    const auto t1 = std::chrono::steady_clock::now();
    void *q = nullptr;
    const hipError_t e = dev_acquire(&q, count * sizeof(T));
    p = static_cast<T *>(q);
    const auto t2 = std::chrono::steady_clock::now();
    g_alloc_ms += std::chrono::duration<double, std::milli>(t2 - t0).count();
    if (dbg) {
        (void)hipMemGetInfo(&f1, &tot);
        fprintf(stderr, "regrow: %.3f GB%s release %.3f ms acquire %.3f ms; free before %.2f after %.2f of %.2f GB\n", (double)(count * sizeof(T)) / 1e9, had ? " (replaces a block)" : "",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), (double)f0 / 1e9, (double)f1 / 1e9, (double)tot / 1e9);
    }
    if (e != hipSuccess) { p = nullptr; HIP_TRY(e); }
    return LIME_OK;
}

// scratch sized for an array of n_avail positions; grow-only, so steady-state calls allocate nothing
static int ensure_scratch(lime_ctx *c, uint64_t n_avail, bool detect, bool score, hipStream_t st)
{
    const size_t n_tiles = (size_t)((n_avail + WIN - 1) / WIN);
    int rc;
    if (n_tiles > c->tile_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        size_t cap = n_tiles + 16;
        if ((rc = regrow(c->d_summ, cap))) return rc;
        if ((rc = regrow(c->d_tile_cnt, cap))) return rc;
        if ((rc = regrow(c->d_tile_off, cap))) return rc;
        if ((rc = regrow(c->d_cross, cap))) return rc;
        c->tile_cap = cap;
    }
    if (score) {
        // clusters longer than SMALL_MAX; at most n/(SMALL_MAX+1) exist, sized for 1 in 4 of that
        const uint64_t want_small = 16, want_big = n_avail / (4u * SMALL_MAX) + 65536u;
        if (want_big > 0xFFFFFFF0ull) return fail(LIME_ERR_ARG, "array too long for one shard: %llu", (unsigned long long)n_avail);
        if (want_small > c->small_cap) { HIP_TRY(hipStreamSynchronize(st)); if ((rc = regrow(c->d_small, want_small))) return rc; c->small_cap = (uint32_t)want_small; }
        if (want_big > c->big_cap) { HIP_TRY(hipStreamSynchronize(st)); if ((rc = regrow(c->d_big, want_big))) return rc; c->big_cap = (uint32_t)want_big; }
        if (!c->d_big_scratch) {
            const size_t words = (size_t)BIG_GRID * BIG_SCRATCH_WORDS;
            if ((rc = regrow(c->d_big_scratch, words))) return rc;      // (through regrow: its time is in the ctx's allocation account)
            HIP_TRY(hipMemsetAsync(c->d_big_scratch, 0, words * sizeof(uint32_t), st));
            launch_fill_u32(c->d_big_scratch, HT_SIZE, HT_EMPTY, st, BIG_GRID, BIG_SCRATCH_WORDS);      // one launch (32 of them were 0.16 ms of a cold 0.25 ms pass)
            HIP_TRY(hipGetLastError());
        }
    }
    if (detect && c->tile_cap > c->wmask_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if ((rc = regrow(c->d_wmask, c->tile_cap))) return rc;
        c->wmask_cap = c->tile_cap;
    }
    return LIME_OK;
}

static int check_ctx(lime_ctx *c, const char *who)
{
    if (!c) return fail(LIME_ERR_ARG, "%s: ctx is NULL", who);
    HIP_TRY(hipSetDevice(c->device));
    return LIME_OK;
}

static bool misaligned(const void *p, size_t a) { return ((uintptr_t)p & (a - 1)) != 0; }

static ScanArgs base_args(lime_ctx *c, const uint32_t *lcp, const uint32_t *da, const uint8_t *ebwt,
                          uint64_t n_own, uint64_t n_avail, int eof, uint32_t n_reads, uint32_t n_refs,
                          uint32_t alpha, uint8_t *sim)
{
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.lcp = lcp; a.da = da; a.ebwt = ebwt;
    a.n_own = n_own; a.n_avail = n_avail; a.pos_base = 0; a.eof = eof;
    a.n_reads = n_reads; a.n_refs = n_refs; a.alpha = alpha;
    a.n_tiles = (uint32_t)((n_avail + WIN - 1) / WIN);
    a.sim = sim; a.summ = c->d_summ; a.open = reinterpret_cast<OpenRec *>(c->d_summ); a.stats = c->d_stats;
    a.small = c->d_small; a.cross_cap = c->small_cap; a.big = c->d_big; a.big_cap = c->big_cap;
    a.tile_cnt = c->d_tile_cnt; a.tile_off = c->d_tile_off; a.cross = c->d_cross; a.out = c->d_out;
    a.wmask = c->d_wmask;
    a.edge = &c->d_stats->edge;
    a.sticky = c->d_sticky; a.dyn = c->d_sticky + 1;
    // Measured with the final round-4 kernels (LIME_SCAN_STATIC_PCT = 0 / 25 / 50 / 75, ABAB): long inputs run faster with every chunk but a
    // workgroup's first handed out as the workgroups get there (configs[2] 1.66 -> 1.60 ms, N = 1e10 14.6 -> 14.4, configs[4]'s shape 17.1 -> 15.9),
    // 1e8 symbols 1 .. 2 % faster with three quarters of the rounds round-robin (fewer trips to the device-wide counter in a 0.2 ms kernel)
    a.static_pct = c->scan_static_pct >= 0 ? (uint32_t)c->scan_static_pct : (n_avail >= 500000000ull ? 0u : 75u);
    a.ablate = c->ablate;
    a.dense_min = c->dense_min;
    a.no_direct = c->no_direct ? 1u : 0u;
    return a;
}

static int flags_to_rc(uint32_t flags)
{
    if (flags & LIME_FLAG_INTERNAL) return fail(LIME_ERR_HIP, "the scan's window hand-out failed (a wave lost its chunk): the pass is invalid");
    if (flags & LIME_FLAG_BADCLUSTER) return fail(LIME_ERR_ARG, "a cluster record lies outside the arrays");
    if (flags & LIME_FLAG_OVERFLOW) return fail(LIME_ERR_NOMEM, "internal cluster list overflow");
    if (flags & LIME_FLAG_POOL_FULL) return fail(LIME_ERR_NOMEM, "update record pool too small (the pass could not be repeated)");
    if (flags & LIME_FLAG_MAXLEN) return fail(LIME_ERR_MAXLEN, "maximum cluster size is greater than %u (sizeMaxBuf)", LIME_MAX_CLUSTER);
    if (flags & LIME_FLAG_HALO) return fail(LIME_ERR_HALO, "a run owned by this shard does not close inside its halo");
    if (flags & LIME_FLAG_DOCID) return fail(LIME_ERR_DOCID, "a da value >= n_reads + n_refs was met while scoring");
    return LIME_OK;
}

extern "C" int lime_set_timing(lime_ctx *c, int on)
{
    int rc = check_ctx(c, "lime_set_timing"); if (rc) return rc;
    c->timing = on != 0; c->ev_used = 0;
    return LIME_OK;
}

static int timing_mark(lime_ctx *c, hipStream_t st)
{
    if (!c->timing) return LIME_OK;
    if (c->ev_used == c->ev.size()) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); c->ev.push_back(e); }
    HIP_TRY(hipEventRecord(c->ev[c->ev_used++], st));
    return LIME_OK;
}

extern "C" int lime_get_timing_ex(lime_ctx *c, double ms_avg[4], uint64_t *launches)
{
    int rc = check_ctx(c, "lime_get_timing_ex"); if (rc) return rc;
    double sum[4] = {0, 0, 0, 0}; uint64_t n = 0;
    for (size_t i = 0; i + 3 < c->ev_used; i += 4) {
        HIP_TRY(hipEventSynchronize(c->ev[i + 3]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i + 1], c->ev[i + 2])); sum[0] += ms;     // the scan kernel
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 3])); sum[1] += ms;         // the whole pass
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i + 2], c->ev[i + 3])); sum[2] += ms;     // after the scan
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1])); sum[3] += ms;         // before the scan (table clear)
        ++n;
    }
    c->ev_used = 0;
    if (ms_avg) for (int k = 0; k < 4; ++k) ms_avg[k] = n ? sum[k] / (double)n : 0.0;
    if (launches) *launches = n;
    return LIME_OK;
}

extern "C" int lime_get_host_times(lime_ctx *c, double out[8])
{
    int rc = check_ctx(c, "lime_get_host_times"); if (rc) return rc;
    if (!out) return fail(LIME_ERR_ARG, "lime_get_host_times: out is NULL");
    out[0] = c->alloc_ms; out[1] = c->probe_ms; out[2] = (double)c->n_probes; out[3] = (double)c->n_repeats; out[4] = (double)c->n_fallbacks;
    out[5] = c->density_known ? c->density : -1.0;
    out[6] = (double)c->n_table_free; out[7] = 0.0;
    return LIME_OK;
}

extern "C" int lime_get_timing(lime_ctx *c, double *scan_ms_avg, uint64_t *launches)
{
    double ms[4];
    int rc = lime_get_timing_ex(c, ms, launches);
    if (!rc && scan_ms_avg) *scan_ms_avg = ms[0];
    return rc;
}

// ---- device-pointer API -----------------------------------------------------------------
// Which way the scan's table updates go.  Binned (records -> bins -> table regions built in LDS, the table
// written once and never cleared) pays for update-dense passes over tables beyond the caches; compare-and-swap
// on the table for sparse ones and wherever the table must be added to (zero_sim == 0, streaming chunks).
static bool want_binned(const lime_ctx *c, uint64_t n_own, size_t sim_bytes, int zero_sim, bool keep_stats, int ebwt)
{
    if (!zero_sim || keep_stats || !n_own) return false;
    if (sim_bytes > ((size_t)BIN_MAX << BIN_SHIFT_MAX) || sim_bytes >= (1ull << CELL_BITS) || sim_bytes > ((uint64_t)MAX_SUB << 32)) return false;
    if (c->upd_pref >= 0) return c->upd_pref == 1;
    if (n_own < (1u << 24)) return false;                 // short passes: the extra launches cost more than they save
    if (sim_bytes < (1u << 20)) return false;             // tiny tables: all updates would land in one or two bins
    // Measured (tools/r03_big.sh, 0.03 updates per symbol, EBWT=1): a table the Infinity Cache holds takes the compare-and-swaps
    // under the scan (configs[1], 50 MB: 0.24 ms against 0.3-0.4 binned); beyond it every update is a 64-byte request to
    // HBM at ~20 G/s while a record costs the later kernels ~7 ps (10^10 symbols: 1 GB table 26.7 ms against 18.4 binned,
    // configs[4]'s 10.3 GB table 31.5 against 20.8, configs[3]'s shape 8.3 against 7.6) -- worth ~0.3 ms of extra launches
    // from about 5 million records on.
    // (round 5, EBWT=1 and a cached table: the compare-and-swap scan runs 12 waves per CU, the record-emitting one 16 -- 0.223 against 0.190 ms
    // per 10^8 symbols -- and the binned pass's fixed launches are 45 us since two of them were merged: level at 10^8 symbols (0.242 : 0.244 ms),
    // binned ahead from there on whatever the density -- 2*10^8: 0.414 against 0.44-0.456, tools/r05_c2_paths.sh)
    if (c->density_known) return sim_bytes > (256u << 20) ? c->density * (double)n_own >= 5e6 : (c->density >= 0.06 || (ebwt && n_own >= 150000000ull));
    // Nothing known yet (a first pass too short for the density probe to pay -- below 2^28 symbols its fixed 0.13 ms is a third to a half of the
    // pass --, or LIME_NO_PROBE): binned.  It is the path that loses little where it loses (configs[1], 0.03 records per symbol: 0.29 against
    // 0.24 ms) and wins much where it wins (the same shape at 0.17: 0.40 against 0.85 ms; text statistics: 0.56 against 4.3 ms); rounds 2-4 took
    // compare-and-swap for tables the Infinity Cache holds.
    return true;
}

// records per owned symbol the pool of the next pass is sized for: what the last pass measured, with a margin (the waves'
// shares differ: ensure_binned adds its own), once one has been read back; the default before that; and never below what
// a repeated pass (pool too small) settled on
// waves of a scan workgroup that count their records together = one producer of k_part: as few as the LDS histogram allows
// (its BIN_MAX counters are shared by the workgroup's producers), so that the partition runs several workgroups per CU
static uint32_t part_prod_waves(const lime_ctx *c, int ebwt, uint32_t n_bins)
{
    const uint32_t wpw = scan_waves_per_wg(ebwt, 0);
    uint32_t best = wpw;
    for (uint32_t pw = wpw; pw >= 1u; --pw) {
        if (wpw % pw) continue;
        const uint32_t h = wpw / pw;
        if (h <= c->part_split && (uint64_t)h * n_bins <= BIN_MAX) best = pw;
    }
    return best;
}

// which record-emitting scan serves a pass: 2 = the scorers write finished records (tables of one or two sub-regions), 1 = through the update queue
static int bin_mode(const lime_ctx *c, uint32_t n_sub) { return (n_sub <= 2u && !c->no_direct) ? 2 : 1; }

static double sizing_density(const lime_ctx *c)
{
    if (c->pool_density_fixed || !c->density_known) return c->pool_density;
    return c->density * 1.25 + 0.002;             // (rounds 3-4 capped this at the default: a collection denser than 0.2 overflowed its first pool)
}

// the largest part of a wave's records that one of its sub-regions (4 GB of table each, the last one what is left) has to take when the
// cells spread evenly over the table: the sub-regions are sized for THAT share of the wave's records (round 5; rounds 3-4 gave every one of
// the n_sub sub-regions room for the wave's whole share, n_sub times the memory and -- with 32-bit positions -- a third of the reach)
static double sub_share(size_t sim_bytes)
{
    return sim_bytes > (1ull << 32) ? (double)(1ull << 32) / (double)sim_bytes : 1.0;
}

// the scan's per-(wave, sub-region) record counts, its per-(bin, producer) counts and the bins' totals / bases (also what the density probe needs)
static int ensure_bin_counters(lime_ctx *c, size_t segs, size_t want_counts, hipStream_t st)
{
    int rc;
    if (segs > c->wave_cap) { HIP_TRY(hipStreamSynchronize(st)); c->wave_cap = 0; if ((rc = regrow(c->d_wave_cnt, segs))) return rc; c->wave_cap = segs; }
    if (want_counts > c->counts_cap) { HIP_TRY(hipStreamSynchronize(st)); c->counts_cap = 0; if ((rc = regrow(c->d_counts, want_counts))) return rc; c->counts_cap = want_counts; }
    if (!c->d_totals) {
        HIP_TRY(hipMalloc(&c->d_totals, (BIN_MAX + 1) * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_binbase, (BIN_MAX + 2) * sizeof(uint64_t)));
    }
    return LIME_OK;
}

// *p64: the pool holds 2^32 records or more -- the partition kernels then run with 64-bit positions (launch_part)
static int ensure_binned(lime_ctx *c, uint64_t n_own, uint32_t n_waves, uint32_t n_prod, uint32_t n_bins, uint32_t bin_shift,
                         uint32_t n_sub, double share, uint32_t *cap_w, bool *p64, hipStream_t st)
{
    int rc;
    const double per_wave = (double)n_own * sizing_density(c) / (double)n_waves;
    // (the waves draw their windows from counters: their record counts differ by a few % -- x 1.35; a sub-region's part of them by a little more)
    const double per_sub = per_wave * share * (n_sub > 1u ? 1.5 : 1.35);
    if (per_sub > 4.0e9) return fail(LIME_ERR_ARG, "update record pool: more than 2^32 records per scan wave and sub-region");
    uint64_t cw = (((uint64_t)per_sub + c->pool_slack) & ~15ull) + 16u;   // a multiple of 16 records: sub-regions start on a 64-byte line
    const size_t segs = (size_t)n_waves * n_sub;
    if (c->pool_cap / segs > cw && c->pool_cap / segs < 0xFFFFFFF0ull) cw = (c->pool_cap / segs) & ~15ull;      // grow-only: use all of what is there
    size_t want = (size_t)cw * segs;
    // 64-bit positions: their high part rides in a record's bits above t through k_part's stage (31 - bin_shift of them)
    if ((uint64_t)want >> (32u + 31u - bin_shift)) return fail(LIME_ERR_ARG, "update record pool too large for one shard");
    *p64 = c->force_p64 || want >= 0xF0000000ull;
    {   // the second level's 16-bit rows (one per tile, a bin's last one partly used) fit the pool
        const size_t rows_words = ((size_t)tiles_bound(want, n_bins) * row_stride() + 1) / 2;
        if (want < rows_words) want = rows_words;
    }
    if (want > c->pool_cap && c->pool_cap >= want - want / 5) {
        // A pool within 20 % of what this pass would ask for is kept: the sizes carry margins of 1.25 x 1.35, a pass that overflows is repeated with
        // a larger one, and replacing a block costs more than it looks -- the driver clears recycled pages inside hipMalloc at about 30 GB/s (the
        // 15.6 GB pool of N = 10^10 clustered was re-grown by 0.02 % after the probe's density had been replaced by the measured one: 4.9 s)
        const uint64_t cw2 = (c->pool_cap / segs) & ~15ull;
        const size_t want2 = (size_t)cw2 * segs, rows2 = ((size_t)tiles_bound(want2, n_bins) * row_stride() + 1) / 2;
        if (cw2 >= 16 && want2 <= c->pool_cap && rows2 <= c->pool_cap) { cw = cw2; want = want2; *p64 = c->force_p64 || want >= 0xF0000000ull; }
    }
    if (want > c->pool_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        c->pool_cap = 0; c->recs_cap = 0;                        // (nothing is there if one of the two cannot be had)
        if ((rc = regrow(c->d_pool, want + 16)) || (rc = regrow(c->d_recs, want + 16))) {      // slack: k_part2 / k_apply read aligned groups of four 4-byte records
            if (c->d_pool) { dev_release(c->d_pool); c->d_pool = nullptr; }
            return rc;
        }
        c->pool_cap = want; c->recs_cap = want;
    }
    if ((rc = ensure_bin_counters(c, segs, (size_t)n_bins * n_prod, st))) return rc;
    if (bin_shift > REGION_SHIFT) {
        if (!c->d_tbase) HIP_TRY(hipMalloc(&c->d_tbase, (BIN_MAX + 2) * sizeof(uint32_t)));
        const size_t want_idx = (size_t)tiles_bound(c->pool_cap, n_bins) * (((size_t)1 << (bin_shift - REGION_SHIFT)) + 1);
        if (want_idx > c->tidx_cap) { HIP_TRY(hipStreamSynchronize(st)); c->tidx_cap = 0; if ((rc = regrow(c->d_tidx, want_idx))) return rc; c->tidx_cap = want_idx; }
    }
    const size_t want_reg = ((size_t)n_bins << (bin_shift - REGION_SHIFT)) + 2;
    if (bin_shift > REGION_SHIFT && want_reg > c->regbase_cap) {
        HIP_TRY(hipStreamSynchronize(st)); c->regbase_cap = 0; if ((rc = regrow(c->d_regbase, want_reg))) return rc; c->regbase_cap = want_reg;
    }
    *cap_w = (uint32_t)cw;
    return LIME_OK;
}

// The table's bins for the binned update path: one bin per 64 KB region for small tables; else as few levels of fan-out
// as fit: at most 2048 bins of 2^k regions (the bins' open output lines then merge in the L2), more bins only when k would
// pass its limit.  A pure function of the table's size (and LIME_BIN_LEVELS): every rank of an exchange gets the same.
static void bin_layout(const lime_ctx *c, size_t sim_bytes, uint32_t *n_bins, uint32_t *bin_shift_out)
{
    uint32_t bin_shift = REGION_SHIFT;
    auto bins_at = [&](uint32_t sh) { return (sim_bytes + ((size_t)1 << sh) - 1) >> sh; };
    const uint32_t bmax = BIN_MAX;                        // what the scan's LDS histogram holds
    const uint32_t one = c->bin_one_level < bmax ? c->bin_one_level : bmax, two = c->bin_two_level < bmax ? c->bin_two_level : bmax;
    if (bins_at(bin_shift) > one) {
        while ((bins_at(bin_shift) > two && bin_shift < BIN_SHIFT_MAX) || bins_at(bin_shift) > bmax) ++bin_shift;
        // fewer, wider bins while that leaves at least 256 of them and at most 64 regions per bin: measured on a 1 GB
        // table (N = 10^10) 477 bins of 32 regions beat 1908 of 8 by 1 ms in 11; a 5 GB table keeps its 1193 bins of 64
        if (!c->bin_levels_forced)
            while (bin_shift < REGION_SHIFT + 6 && bin_shift < BIN_SHIFT_MAX && bins_at(bin_shift + 1) >= 256) ++bin_shift;
        // Round 6: wider bins still where that brings the table under LINES_BINS bins -- k_part_lines (whole 64-byte lines, two workgroups per CU) then
        // does the first level instead of k_part (pieces of lines: 2 against 3.4 TB/s), and since k_apply_tiles shares a wave among short runs the
        // second level no longer pays for the regions per bin: configs[2] (5 GB: 1193 bins of 64 regions -> 299 of 256) 3.10 -> 2.95 ms per pass,
        // configs[4]'s shape (10.3 GB: 1229 of 128 -> 308 of 512) clustered 31.8 -> 31.0 ms.  Tables beyond 477 x 32 MB = 16 GB keep what they had.
        constexpr uint32_t LINES_BINS = 477;                 // (what fits a CU twice, 32- and 64-bit positions: tools/kres.py gates it)
        if (!c->bin_levels_forced && bins_at(bin_shift) > LINES_BINS) {
            uint32_t sh = bin_shift;
            while (sh < BIN_SHIFT_MAX && bins_at(sh) > LINES_BINS) ++sh;
            if (bins_at(sh) <= LINES_BINS) bin_shift = sh;
        }
    }
    *n_bins = (uint32_t)bins_at(bin_shift);               // <= BIN_MAX: want_binned checked the table size
    *bin_shift_out = bin_shift;
}

static int read_stats(lime_ctx *c, lime_stats_t *s, hipStream_t st, uint32_t *sticky = nullptr);
// which variant of k_apply_tiles / of k_sort_tiles' row stores a pass of `records` update records takes (lime_set_option "apply_wide" / "sort_nt" force one)
static bool many_records_of(const lime_ctx *c, double records) { return c->apply_wide >= 0 ? c->apply_wide != 0 : records >= 2e8; }
static bool big_rows_of(const lime_ctx *c, double records) { return c->sort_nt >= 0 ? c->sort_nt != 0 : records >= 1e8; }

// Update records per owned symbol, estimated from a sample before the first pass on a ctx: the record-emitting scan kernel runs over every
// 2^ps-th chunk of 16 windows -- spread over the whole collection, on all CUs -- with sub-regions of capacity 0: every update record is counted
// (lime_stats_t.n_updates) and none is stored, no histogram entry made, no table touched.  About 1/64 of a pass + one synchronisation.
// Reference: what is sampled is the number of `SimArray_[r][g] += t` executions per symbol, ClusterBWT_DA.cpp:178-184, 243-248.
static int density_probe(lime_ctx *c, const uint32_t *d_lcp, const uint32_t *d_da, const uint8_t *d_ebwt, uint64_t n_own, uint64_t n_avail, int eof,
                         uint32_t n_reads, uint32_t n_refs, uint32_t alpha, size_t sim_bytes, hipStream_t st)
{
    int rc;
    const auto t0 = std::chrono::steady_clock::now();
    const int ebwt = d_ebwt != nullptr;
    const uint32_t n_tiles = (uint32_t)((n_avail + WIN - 1) / WIN);
    const uint32_t ps = n_own < (1ull << 30) ? 6u : n_own < (1ull << 32) ? 7u : 8u;
    uint32_t n_bins = 0, bin_shift = REGION_SHIFT;
    bin_layout(c, sim_bytes, &n_bins, &bin_shift);
    const uint32_t n_sub = (uint32_t)((sim_bytes + 0xFFFFFFFFull) >> 32), wpw = scan_waves_per_wg(ebwt, 0);
    const uint32_t grid = scan_grid(ebwt, 0, bin_mode(c, n_sub), n_tiles, c->max_blocks, ps);
    const uint32_t prod_waves = part_prod_waves(c, ebwt, n_bins), n_prod = grid * (wpw / prod_waves);
    g_alloc_ms = 0.0;
    rc = ensure_bin_counters(c, (size_t)grid * wpw * n_sub, (size_t)n_bins * n_prod, st);
    c->alloc_ms += g_alloc_ms;
    if (rc) return rc;
    launch_zero2(c->d_stats, sizeof(DevStats), nullptr, 0, st);
    ScanArgs a = base_args(c, d_lcp, d_da, d_ebwt, n_own, n_avail, eof, n_reads, n_refs, alpha, nullptr);
    a.upd_mode = 1; a.pool = reinterpret_cast<uint32_t *>(c->d_stats);       // (never written: no slot is below a capacity of 0)
    a.cap_w = 0; a.n_sub = n_sub; a.wave_cnt = c->d_wave_cnt; a.counts = c->d_counts;
    a.n_bins = n_bins; a.bin_shift = bin_shift; a.prod_waves = prod_waves;
    a.sub_rb = 0xFFFFFFFFu; a.sub_gb = 0u;
    if (n_sub == 2) { a.sub_rb = (uint32_t)((1ull << 32) / n_refs); a.sub_gb = (uint32_t)((1ull << 32) - (uint64_t)a.sub_rb * n_refs); }
    a.probe_shift = ps; a.static_pct = 100u;
    launch_tile(ebwt, 0, a, c->max_blocks, st);
    HIP_TRY(hipGetLastError());
    lime_stats_t s;
    if ((rc = read_stats(c, &s, st))) return rc;                              // waits for the sample
    HIP_TRY(hipMemsetAsync(c->d_sticky, 0, 4, st));                           // (every sub-region "overflowed": not a pass to settle)
    const uint64_t chunk = (uint64_t)wpw * WIN, phys = (n_avail + chunk - 1) / chunk, logical = (phys + (1ull << ps) - 1) >> ps;
    uint64_t sampled = logical * chunk;
    if (sampled > n_own) sampled = n_own;
    c->density = (double)s.n_updates / (double)(sampled ? sampled : 1); c->density_known = true;
    c->probe_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ++c->n_probes;
    if (c->debug_stats) fprintf(stderr, "density_probe: every %u-th chunk, %llu symbols, %llu updates -> %.4f records per symbol\n", 1u << ps,
                                            (unsigned long long)sampled, (unsigned long long)s.n_updates, c->density);
    return LIME_OK;
}

// keep_stats: this call continues a position-range sequence on the same table (lime_fused_stream):
// cluster / update counters and flags accumulate, only the per-call list counters restart.
static int fused_dev_impl(lime_ctx *c, const uint32_t *d_lcp, const uint32_t *d_da, const uint8_t *d_ebwt,
                          uint64_t n_own, uint64_t n_avail, int eof, uint32_t n_reads, uint32_t n_refs,
                          uint32_t alpha, uint8_t *d_sim, int zero_sim, bool keep_stats, hipStream_t st,
                          uint32_t *d_edge = nullptr, bool no_bin = false, bool records_only = false)
{
    int rc;
    if (n_own > n_avail) return fail(LIME_ERR_ARG, "lime_fused_dev: n_own > n_avail");
    if (n_avail && (!d_lcp || !d_da || (!d_sim && !records_only))) return fail(LIME_ERR_ARG, "lime_fused_dev: NULL array");
    if (misaligned(d_lcp, 16) || misaligned(d_da, 16) || misaligned(d_ebwt, 8) || misaligned(d_sim, 16))
        return fail(LIME_ERR_ARG, "lime_fused_dev: device arrays must be 16-byte aligned (ebwt: 8)");
    if (!n_reads || !n_refs) return fail(LIME_ERR_ARG, "lime_fused_dev: n_reads and n_refs must be > 0");
    if (n_refs >= MAX_REFS || (uint64_t)n_reads + n_refs > 0xFFFFFFF0ull)
        return fail(LIME_ERR_ARG, "lime_fused_dev: n_refs must be < 2^%u and n_reads + n_refs <= 2^32 - 16", T_SHIFT);
    if ((n_avail + WIN - 1) / WIN > 0xFFFFFFF0ull) return fail(LIME_ERR_ARG, "array too long for one shard: %llu", (unsigned long long)n_avail);
    g_alloc_ms = 0.0;
    rc = ensure_scratch(c, n_avail, false, true, st);
    c->alloc_ms += g_alloc_ms;
    if (rc) return rc;
    const size_t sim_bytes = lime_sim_bytes(n_reads, n_refs);
    const int ebwt = d_ebwt != nullptr;
    const uint32_t n_tiles = (uint32_t)((n_avail + WIN - 1) / WIN);
    // no_bin: a chunk of a multi-chunk stream -- its device buffers are reused by later chunks, so the pass could not be
    // repeated after a pool overflow, and the later chunks add to the table by compare-and-swap
    const bool bin_fits = !(sim_bytes > ((size_t)BIN_MAX << BIN_SHIFT_MAX) || sim_bytes >= (1ull << CELL_BITS) || sim_bytes > ((uint64_t)MAX_SUB << 32));
    if (records_only && !bin_fits) return fail(LIME_ERR_ARG, "lime_fused_records_dev: table too large for update records");
    // The first pass on a ctx knows nothing of the collection's update density, which decides the update path and sizes the record pool
    // (rounds 2-4 guessed 0.2 records per symbol: text has 0.24 .. 0.39, the iid generators 0.03 .. 0.12, and a pass that guessed wrong
    // was repeated or ran on the other path).  LiME_paired.sh:62-68 runs every collection ONCE, so the density is sampled first: the
    // scan kernel itself over every 2^k-th chunk of 16 windows, counting its update records without storing one (density_probe).
    if (n_avail && !no_bin && !keep_stats && zero_sim && bin_fits && c->probe && !c->density_known && !c->pool_density_fixed && !c->ablate &&
        c->upd_pref != 0 && n_own >= c->probe_min && sim_bytes >= (1u << 20) && n_tiles < 0x7FF00000u)
        if ((rc = density_probe(c, d_lcp, d_da, d_ebwt, n_own, n_avail, eof, n_reads, n_refs, alpha, sim_bytes, st))) return rc;
    bool binned = n_avail && !no_bin && want_binned(c, n_own, sim_bytes, zero_sim, keep_stats, ebwt);
    if (records_only) binned = true;                      // the records ARE the result: the binned path or nothing
    uint32_t grid = 0, cap_w = 0, n_bins = 0, bin_shift = REGION_SHIFT, n_sub = 1, prod_waves = 0, n_prod = 0;
    bool p64 = false, fell_back = false;
    const double share = sub_share(sim_bytes);
    if (binned) {
        bin_layout(c, sim_bytes, &n_bins, &bin_shift);
        n_sub = (uint32_t)((sim_bytes + 0xFFFFFFFFull) >> 32);
        grid = scan_grid(ebwt, 0, bin_mode(c, n_sub), n_tiles, c->max_blocks);
        prod_waves = part_prod_waves(c, ebwt, n_bins);
        n_prod = grid * (scan_waves_per_wg(ebwt, 0) / prod_waves);
        g_alloc_ms = 0.0;
        rc = ensure_binned(c, n_own, grid * scan_waves_per_wg(ebwt, 0), n_prod, n_bins, bin_shift, n_sub, share, &cap_w, &p64, st);
        c->alloc_ms += g_alloc_ms;
        if (rc == LIME_ERR_NOMEM && !records_only) {
            // no room for the records: the pass runs on the compare-and-swap path -- several times slower where the binned path was wanted --
            // and says so: LIME_FLAG_CAS_FALLBACK in the pass's statistics, the reason in lime_last_error()
            (void)hipGetLastError();
            const std::string why = g_err;
            (void)fail(LIME_ERR_NOMEM, "lime_fused_dev: no device memory for the update records of the binned path (%s): this pass falls back to "
                                       "compare-and-swap on the table (LIME_FLAG_CAS_FALLBACK)", why.c_str());
            binned = false; fell_back = true; ++c->n_fallbacks;
        } else if (rc) return rc;
        if (p64 && !c->by_tiles && bin_shift > REGION_SHIFT) return fail(LIME_ERR_ARG, "LIME_SECOND_LEVEL=sweeps: a record pool of 2^32 records or more needs the tile kernels");
    }
    if ((rc = timing_mark(c, st))) return rc;
    if (keep_stats) {
        HIP_TRY(hipMemsetAsync(&c->d_stats->n_cross, 0, 2 * sizeof(uint32_t), st));      // n_cross, n_big
        HIP_TRY(hipMemsetAsync(&c->d_stats->n_open, 0, sizeof(uint32_t), st));
        if (zero_sim && !binned) HIP_TRY(hipMemsetAsync(d_sim, 0, sim_bytes, st));
    } else {
        // the counters and (compare-and-swap path: the binned path writes every byte of the table itself) the table, one launch
        static_assert(sizeof(DevStats) % 4 == 0, "whole words");
        const bool zt = zero_sim && !binned && d_sim;
        launch_zero2(c->d_stats, sizeof(DevStats), zt ? d_sim : nullptr, zt ? (sim_bytes & ~(size_t)15) : 0, st);
        if (zt && (sim_bytes & 15)) HIP_TRY(hipMemsetAsync(d_sim + (sim_bytes & ~(size_t)15), 0, sim_bytes & 15, st));
    }
    if (records_only) {                                   // the long clusters' updates leave as records too
        if (!c->d_bigrec) {
            c->bigrec_cap = 16u << 20;
            HIP_TRY(dev_acquire((void **)&c->d_bigrec, (size_t)c->bigrec_cap * sizeof(uint64_t)));
            HIP_TRY(hipMalloc(&c->d_bigrec_n, sizeof(uint32_t)));
        }
        HIP_TRY(hipMemsetAsync(c->d_bigrec_n, 0, sizeof(uint32_t), st));
        c->rec_n_bins = n_bins; c->rec_bin_shift = bin_shift;
    }
    if (!n_avail && !records_only) { if ((rc = timing_mark(c, st)) || (rc = timing_mark(c, st)) || (rc = timing_mark(c, st))) return rc; return LIME_OK; }
    ScanArgs a = base_args(c, d_lcp, d_da, d_ebwt, n_own, n_avail, eof, n_reads, n_refs, alpha, d_sim);
    if (d_edge) a.edge = d_edge;                          // a chunk of a stream: its own (cleared) word
    if (binned) {
        a.upd_mode = 1; a.pool = c->d_pool; a.cap_w = cap_w; a.n_sub = n_sub; a.wave_cnt = c->d_wave_cnt; a.counts = c->d_counts;
        a.n_bins = n_bins; a.bin_shift = bin_shift; a.prod_waves = prod_waves;
        a.sub_rb = 0xFFFFFFFFu; a.sub_gb = 0u;
        if (n_sub == 2) { a.sub_rb = (uint32_t)((1ull << 32) / n_refs); a.sub_gb = (uint32_t)((1ull << 32) - (uint64_t)a.sub_rb * n_refs); }
    }
    if (records_only) { a.sim = nullptr; a.bigrec = c->d_bigrec; a.bigrec_n = c->d_bigrec_n; a.bigrec_cap = c->bigrec_cap; }
    if ((rc = timing_mark(c, st))) return rc;
    launch_tile(ebwt, 0, a, c->max_blocks, st);
    if ((rc = timing_mark(c, st))) return rc;
    if (!(binned && !c->ablate)) launch_resolve(0, a, st);
    if (binned && !c->ablate) {                          // (timing experiments cut the scan short: nothing to partition)
        const bool tiles = bin_shift > REGION_SHIFT && c->by_tiles && !records_only;
        launch_rowscan_resolve(a, c->d_counts, c->d_totals, n_bins, n_prod, st);      // (the open segments are closed in the same launch)
        launch_bin_bases(c->d_totals, c->d_binbase, tiles ? c->d_tbase : nullptr, n_bins, st);
        const uint32_t *recs = c->d_recs;
        if (c->p64_test_base && !records_only && (c->by_tiles || bin_shift == REGION_SHIFT)) {      // tests: positions from a base near a multiple of 2^32 on
            launch_add_u64(c->d_binbase, (size_t)n_bins + 1, c->p64_test_base, st);
            recs = c->d_recs - c->p64_test_base;                   // (an address only: the kernels add positions >= the base to it)
        }
        launch_part(a, n_prod, c->d_binbase, const_cast<uint32_t *>(recs), st, p64, c->part_lines != 0);
        if (records_only) {
            // the records grouped by bin are the result: the owners of the bins build the table (lime_apply_records_dev)
        } else if (bin_shift > REGION_SHIFT && c->by_tiles) {      // second level tile by tile into the (by now free) pool, regions from the tiles' runs
            // (how many records: what the last pass counted per symbol, once one has been read back)
            const double expect = c->density_known ? c->density * (double)n_own : 0.0;
            launch_apply_by_tiles(d_sim, sim_bytes, recs, c->d_binbase, n_bins, bin_shift, c->d_tbase, c->d_tidx,
                                  reinterpret_cast<uint16_t *>(c->d_pool), many_records_of(c, expect), st, big_rows_of(c, expect), !c->p64_test_base);
        } else if (bin_shift > REGION_SHIFT) {            // second level into the (by now free) pool, then regions from there
            uint32_t *recs2 = c->d_pool;
            launch_part2(c->d_recs, c->d_binbase, n_bins, bin_shift, c->d_regbase, recs2, st);
            // the base after the last region = the total (regions past the table's end hold no records)
            HIP_TRY(hipMemcpyAsync(c->d_regbase + ((size_t)n_bins << (bin_shift - REGION_SHIFT)), c->d_binbase + n_bins, sizeof(uint64_t),
                                   hipMemcpyDeviceToDevice, st));
            launch_apply(d_sim, sim_bytes, recs2, c->d_regbase, bin_shift, st);
        } else {
            launch_apply(d_sim, sim_bytes, recs, c->d_binbase, bin_shift, st);
        }
    }
    launch_score_big(ebwt, a, c->d_big_scratch, st);      // after k_apply: its compare-and-swaps add to the finished table
    if ((rc = timing_mark(c, st))) return rc;
    HIP_TRY(hipGetLastError());
    if (!keep_stats) {
        lime_ctx::Last &l = c->last;
        l.valid = true; l.binned = binned; l.lcp = d_lcp; l.da = d_da; l.ebwt = d_ebwt; l.n_own = n_own; l.n_avail = n_avail;
        l.eof = eof; l.n_reads = n_reads; l.n_refs = n_refs; l.alpha = alpha; l.sim = d_sim; l.zero_sim = zero_sim; l.st = st;
        l.n_waves = grid * scan_waves_per_wg(ebwt, 0);
        l.own_total = n_own; l.records_only = records_only; l.share = share; l.fell_back = fell_back;
    } else {
        c->last.own_total += n_own;         // a later chunk of a stream: the update counter keeps accumulating
        c->last.binned = false;             // and the pass stored in `last` can no longer be repeated on its own
    }
    return LIME_OK;
}

extern "C" int lime_fused_dev(lime_ctx *c, const uint32_t *d_lcp, const uint32_t *d_da, const uint8_t *d_ebwt,
                              uint64_t n_own, uint64_t n_avail, int eof, uint32_t n_reads, uint32_t n_refs,
                              uint32_t alpha, uint8_t *d_sim, int zero_sim, void *stream)
{
    int rc = check_ctx(c, "lime_fused_dev"); if (rc) return rc;
    return fused_dev_impl(c, d_lcp, d_da, d_ebwt, n_own, n_avail, eof, n_reads, n_refs, alpha, d_sim, zero_sim, false,
                          (hipStream_t)stream);
}


// ---- owner-partitioned exchange of table updates (several GPUs, large tables) -------------------------------------
// Instead of a private table per rank and a dense reduce-scatter of whole tables (every rank allocates and writes T bytes
// and moves T (G-1)/G over xGMI), a rank leaves its updates as records grouped by table bin; the owner of a range of bins
// receives the slices of its bins from every rank and builds its block of the table alone: T/G bytes per rank, about
// 4 bytes per update over the links.  The reference's counterpart is the cluster-range split of ClusterBWT_DA.cpp:630-670
// with all threads adding into one table.
extern "C" int lime_records_layout(lime_ctx *c, uint32_t n_reads, uint32_t n_refs, uint32_t *n_bins, uint32_t *bin_shift)
{
    int rc = check_ctx(c, "lime_records_layout"); if (rc) return rc;
    if (!n_reads || !n_refs || !n_bins || !bin_shift) return fail(LIME_ERR_ARG, "lime_records_layout: bad argument");
    bin_layout(c, lime_sim_bytes(n_reads, n_refs), n_bins, bin_shift);
    return LIME_OK;
}

extern "C" int lime_fused_records_dev(lime_ctx *c, const uint32_t *d_lcp, const uint32_t *d_da, const uint8_t *d_ebwt,
                                      uint64_t n_own, uint64_t n_avail, int eof, uint32_t n_reads, uint32_t n_refs,
                                      uint32_t alpha, void *stream)
{
    int rc = check_ctx(c, "lime_fused_records_dev"); if (rc) return rc;
    return fused_dev_impl(c, d_lcp, d_da, d_ebwt, n_own, n_avail, eof, n_reads, n_refs, alpha, nullptr, 1, false,
                          (hipStream_t)stream, nullptr, false, true);
}

extern "C" int lime_records_get(lime_ctx *c, lime_records_t *out, uint64_t *h_binbase, void *stream)
{
    int rc = check_ctx(c, "lime_records_get"); if (rc) return rc;
    if (!out) return fail(LIME_ERR_ARG, "lime_records_get: out is NULL");
    if (!c->last.valid || !c->last.records_only) return fail(LIME_ERR_ARG, "lime_records_get: the last pass on this ctx was not lime_fused_records_dev");
    hipStream_t st = (hipStream_t)stream;
    uint32_t nb = 0;
    HIP_TRY(hipMemcpyAsync(&nb, c->d_bigrec_n, sizeof nb, hipMemcpyDeviceToHost, st));
    if (h_binbase) HIP_TRY(hipMemcpyAsync(h_binbase, c->d_binbase, ((size_t)c->rec_n_bins + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (nb > c->bigrec_cap) return fail(LIME_ERR_NOMEM, "more update records of long clusters (%u) than their list holds (%u)", nb, c->bigrec_cap);
    out->n_bins = c->rec_n_bins; out->bin_shift = c->rec_bin_shift;
    out->d_recs = c->d_recs; out->d_binbase = c->d_binbase; out->d_bigrecs = c->d_bigrec; out->n_bigrecs = nb;
    return LIME_OK;
}

// the same without a word read back (lime_comm_exchange_records takes bases and counts from the device): the arrays, where the
// long clusters' record count lives, and what their list holds
int lime_internal_records_peek(lime_ctx *c, lime_records_t *out, const uint32_t **d_bigrec_n, uint32_t *bigrec_cap)
{
    int rc = check_ctx(c, "lime_comm_exchange_records"); if (rc) return rc;
    if (!c->last.valid || !c->last.records_only) return fail(LIME_ERR_ARG, "lime_comm_exchange_records: the last pass on this ctx was not lime_fused_records_dev");
    out->n_bins = c->rec_n_bins; out->bin_shift = c->rec_bin_shift;
    out->d_recs = c->d_recs; out->d_binbase = c->d_binbase; out->d_bigrecs = c->d_bigrec; out->n_bigrecs = 0;
    *d_bigrec_n = c->d_bigrec_n; *bigrec_cap = c->bigrec_cap;
    return LIME_OK;
}

// d_rx: the record slices received for this rank's bins, source after source; h_srcoff[s * (nb + 1) + b]: where source s's
// records of local bin b start in d_rx (h_srcoff[s * (nb + 1) + nb]: where they end).  Builds bytes [cell_lo, cell_lo +
// block_bytes) of the table -- cell_lo = first own bin << bin_shift -- in d_block: every byte is written.
extern "C" int lime_apply_records_dev(lime_ctx *c, uint32_t n_src, const uint32_t *d_rx, const uint64_t *h_srcoff, uint32_t nb,
                                      uint32_t bin_shift, const uint64_t *d_bigrecs, uint64_t n_bigrecs, uint64_t cell_lo,
                                      uint64_t block_bytes, uint8_t *d_block, void *stream)
{
    int rc = check_ctx(c, "lime_apply_records_dev"); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!n_src || !h_srcoff || !d_block || (n_bigrecs && !d_bigrecs)) return fail(LIME_ERR_ARG, "lime_apply_records_dev: NULL argument");
    if (bin_shift < REGION_SHIFT || bin_shift > BIN_SHIFT_MAX || (block_bytes & 15u) || misaligned(d_block, 16) ||
        block_bytes > ((uint64_t)nb << bin_shift) || (cell_lo & (((uint64_t)1 << bin_shift) - 1u)))
        return fail(LIME_ERR_ARG, "lime_apply_records_dev: bad block geometry");
    if (!nb || !block_bytes) return LIME_OK;
    // where every bin starts in the regrouped array: the sources' counts added up
    std::vector<uint64_t> dstbase((size_t)nb + 1, 0);
    for (uint32_t b = 0; b < nb; ++b) {
        uint64_t cnt = 0;
        for (uint32_t s = 0; s < n_src; ++s) {
            const uint64_t lo = h_srcoff[(size_t)s * (nb + 1) + b], hi = h_srcoff[(size_t)s * (nb + 1) + b + 1];
            if (hi < lo) return fail(LIME_ERR_ARG, "lime_apply_records_dev: source offsets not ascending");
            cnt += hi - lo;
        }
        dstbase[b + 1] = dstbase[b] + cnt;
    }
    const uint64_t total = dstbase[nb];
    if (total > 0xF0000000ull) return fail(LIME_ERR_ARG, "lime_apply_records_dev: too many records for one block");
    if (total && !d_rx) return fail(LIME_ERR_ARG, "lime_apply_records_dev: d_rx is NULL");
    const size_t f2 = (size_t)1 << (bin_shift - REGION_SHIFT), n_reg = (size_t)nb * f2;
    size_t xwant = (size_t)total + 16;
    {   // (the second level's 16-bit tile rows)
        const size_t rows_words = ((size_t)tiles_bound(total, nb) * row_stride() + 1) / 2 + 16;
        if (xwant < rows_words) xwant = rows_words;
    }
    if (xwant > c->xrecs_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if ((rc = regrow(c->d_xrecs, xwant + total / 8))) return rc;
        if ((rc = regrow(c->d_xrecs2, xwant + total / 8))) return rc;
        c->xrecs_cap = xwant + total / 8;
    }
    if (bin_shift > REGION_SHIFT) {
        if (!c->d_tbase) HIP_TRY(hipMalloc(&c->d_tbase, (BIN_MAX + 2) * sizeof(uint32_t)));
        const size_t want_idx = (size_t)tiles_bound(total, nb) * (f2 + 1);
        if (want_idx > c->tidx_cap) { HIP_TRY(hipStreamSynchronize(st)); if ((rc = regrow(c->d_tidx, want_idx))) return rc; c->tidx_cap = want_idx; }
    }
    const size_t off_words = (size_t)n_src * (nb + 1) + (nb + 1);
    if (off_words > c->xoff_cap) { HIP_TRY(hipStreamSynchronize(st)); if ((rc = regrow(c->d_xoff, off_words))) return rc; c->xoff_cap = off_words; }
    if (n_reg + 2 > c->xreg_cap) { HIP_TRY(hipStreamSynchronize(st)); if ((rc = regrow(c->d_xreg, n_reg + 2))) return rc; c->xreg_cap = n_reg + 2; }
    uint64_t *d_srcoff = c->d_xoff, *d_dstbase = c->d_xoff + (size_t)n_src * (nb + 1);
    // the offsets go up from a pinned buffer of the ctx (the caller's and this function's vectors go out of scope while the copy
    // may still be queued): the only wait is for the PREVIOUS call's copy out of that buffer, long done by now
    if (c->ev_xoff_pending) { HIP_TRY(hipEventSynchronize(c->ev_xoff)); c->ev_xoff_pending = false; }
    if (off_words > c->h_xoff_cap) {
        if (c->h_xoff) (void)hipHostFree(c->h_xoff);
        c->h_xoff = nullptr; c->h_xoff_cap = 0;
        HIP_TRY(hipHostMalloc(&c->h_xoff, (off_words + off_words / 4) * sizeof(uint64_t)));
        c->h_xoff_cap = off_words + off_words / 4;
    }
    if (!c->ev_xoff) HIP_TRY(hipEventCreateWithFlags(&c->ev_xoff, hipEventDisableTiming));
    memcpy(c->h_xoff, h_srcoff, (size_t)n_src * (nb + 1) * sizeof(uint64_t));
    memcpy(c->h_xoff + (size_t)n_src * (nb + 1), dstbase.data(), ((size_t)nb + 1) * sizeof(uint64_t));
    HIP_TRY(hipMemcpyAsync(d_srcoff, c->h_xoff, off_words * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(c->ev_xoff, st)); c->ev_xoff_pending = true;
    launch_regroup(d_rx, d_srcoff, n_src, nb, d_dstbase, c->d_xrecs, st);
    if (bin_shift > REGION_SHIFT && c->by_tiles) {
        launch_apply_by_tiles(d_block, (size_t)block_bytes, c->d_xrecs, d_dstbase, nb, bin_shift, c->d_tbase, c->d_tidx,
                              reinterpret_cast<uint16_t *>(c->d_xrecs2), many_records_of(c, (double)total), st, big_rows_of(c, (double)total));
    } else if (bin_shift > REGION_SHIFT) {
        launch_part2(c->d_xrecs, d_dstbase, nb, bin_shift, c->d_xreg, c->d_xrecs2, st);
        HIP_TRY(hipMemcpyAsync(c->d_xreg + n_reg, d_dstbase + nb, sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        launch_apply(d_block, (size_t)block_bytes, c->d_xrecs2, c->d_xreg, bin_shift, st);
    } else {
        launch_apply(d_block, (size_t)block_bytes, c->d_xrecs, d_dstbase, bin_shift, st);
    }
    launch_apply_bigrecs(d_bigrecs, n_bigrecs, cell_lo, cell_lo + block_bytes, d_block, st);
    HIP_TRY(hipGetLastError());
    return LIME_OK;
}

static int read_stats(lime_ctx *c, lime_stats_t *s, hipStream_t st, uint32_t *sticky)
{
    struct H { lime_stats_t s; uint32_t sticky[4]; };
    H *h = static_cast<H *>(c->h_stats);
    HIP_TRY(hipMemcpyAsync(h, c->d_stats, sizeof(lime_stats_t) + 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *s = h->s;
    if (sticky) *sticky = h->sticky[0];
    return LIME_OK;
}

extern "C" int lime_get_stats(lime_ctx *c, lime_stats_t *out, void *stream)
{
    int rc = check_ctx(c, "lime_get_stats"); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    lime_stats_t s;
    uint32_t unsettled = 0;
    if ((rc = read_stats(c, &s, st, &unsettled))) return rc;
    if (c->debug_stats) fprintf(stderr, "lime_get_stats: flags %u unsettled %u wave_records_max %u n_updates %llu\n", s.flags, unsettled, s.wave_records_max, (unsigned long long)s.n_updates);
    // A pass EARLIER than the last one overflowed its record pool and no lime_get_stats came before the next pass
    // (every pass clears the flags; the sticky word survives): that pass's table is short and cannot be repaired now.
    if (unsettled > ((s.flags & LIME_FLAG_POOL_FULL) ? 1u : 0u)) {
        HIP_TRY(hipMemsetAsync(c->d_sticky, 0, 4, st));
        if (out) *out = s;
        return fail(LIME_ERR_NOMEM, "an earlier lime_fused_dev pass on this ctx overflowed its update record pool and was followed by another "
                                    "pass before lime_get_stats: that pass's table is incomplete (call lime_get_stats after every pass)");
    }
    if (unsettled) HIP_TRY(hipMemsetAsync(c->d_sticky, 0, 4, st));
    // binned table updates: the record pool was too small for this pass -> the table is incomplete.  The pass
    // told how many records its busiest wave produced: repeat it with a pool sized for that (the caller's arrays
    // are still in place: nothing was reported to it yet), at most twice; then on the compare-and-swap path.
    lime_ctx::Last &l = c->last;
    for (int attempt = 0; (s.flags & LIME_FLAG_POOL_FULL) && l.valid && l.binned && attempt < 3; ++attempt) {
        const int pref = c->upd_pref;
        if (attempt == 2 && !l.records_only) c->upd_pref = 0;
        // (wave_records_max is the fullest SUB-REGION's count, sized for `share` of its wave's records)
        const double need = (double)s.wave_records_max * (double)l.n_waves / (double)(l.n_own ? l.n_own : 1) / (l.share > 0.0 ? l.share : 1.0);
        ++c->n_repeats;
        const double was = sizing_density(c);
        c->pool_density = need * 1.08 > was * 1.5 ? need * 1.08 : was * 1.5; c->pool_density_fixed = true;
        const bool timing = c->timing; c->timing = false;
        rc = fused_dev_impl(c, l.lcp, l.da, l.ebwt, l.n_own, l.n_avail, l.eof, l.n_reads, l.n_refs, l.alpha, l.sim, l.zero_sim,
                            false, l.st, nullptr, false, l.records_only);
        c->timing = timing; c->upd_pref = pref;
        if (rc) return rc;
        if ((rc = read_stats(c, &s, l.st, &unsettled))) return rc;
        if (unsettled) HIP_TRY(hipMemsetAsync(c->d_sticky, 0, 4, l.st));        // settled here (or reported through the flags below)
    }
    // records for an owner-partitioned exchange: the long clusters' updates (one 8-byte record per read x genome pair of every
    // cluster beyond the in-scan limit: a single cluster of 8 k symbols can make 16 million) did not fit their list -- treated
    // like the pool: the count went on past the capacity, so the list is regrown to it and the pass repeated
    if (l.valid && l.records_only && (s.flags & LIME_FLAG_OVERFLOW) && c->d_bigrec_n) {
        uint32_t nb = 0;
        HIP_TRY(hipMemcpyAsync(&nb, c->d_bigrec_n, sizeof nb, hipMemcpyDeviceToHost, l.st));
        HIP_TRY(hipStreamSynchronize(l.st));
        if (nb > c->bigrec_cap) {
            const uint64_t want = (uint64_t)nb + nb / 8u + 4096u;
            if (want > 0xFFFFFFF0ull) return fail(LIME_ERR_NOMEM, "%u update records of long clusters: too many for one shard's list", nb);
            uint64_t *bigger = nullptr;
            HIP_TRY(dev_acquire((void **)&bigger, (size_t)want * sizeof(uint64_t)));
            dev_release(c->d_bigrec);
            c->d_bigrec = bigger; c->bigrec_cap = (uint32_t)want;
            const bool timing = c->timing; c->timing = false;
            rc = fused_dev_impl(c, l.lcp, l.da, l.ebwt, l.n_own, l.n_avail, l.eof, l.n_reads, l.n_refs, l.alpha, l.sim, l.zero_sim,
                                false, l.st, nullptr, false, true);
            c->timing = timing;
            if (rc) return rc;
            if ((rc = read_stats(c, &s, l.st, &unsettled))) return rc;
            if (unsettled) HIP_TRY(hipMemsetAsync(c->d_sticky, 0, 4, l.st));
        }
    }
    if (l.valid && l.own_total) { c->density = (double)s.n_updates / (double)l.own_total; c->density_known = true; }
    if (l.valid && l.fell_back) s.flags |= LIME_FLAG_CAS_FALLBACK;
    if (out) *out = s;
    if (c->big_cap && s.n_big > c->big_cap)
        return fail(LIME_ERR_NOMEM, "more clusters longer than %u symbols (%u) than the list holds (%u)", SMALL_MAX, s.n_big, c->big_cap);
    if ((rc = flags_to_rc(s.flags))) return rc;
    if (s.edge & LIME_EDGE_OPEN)
        return fail(LIME_ERR_HALO, "a run owned by this shard is still open where its arrays end: whether it is a cluster is decided by "
                                   "the shards' edge words together (lime_combine_edges)");
    return LIME_OK;
}

extern "C" int lime_combine_edges(const uint32_t *edge, uint32_t n_shards)
{
    if (n_shards && !edge) return fail(LIME_ERR_ARG, "lime_combine_edges: edge is NULL");
    bool open = false, r = false, g = false;
    for (uint32_t k = 0; k < n_shards; ++k) {
        const uint32_t e = edge[k];
        if (open) {                                        // the run goes on through this shard's leading positions
            r = r || (e & LIME_EDGE_LEAD_R); g = g || (e & LIME_EDGE_LEAD_G);
            if ((e & LIME_EDGE_LEAD_HEAD) || k + 1 == n_shards) {          // closed by a head, or by the end of the collection
                if (r && g) return fail(LIME_ERR_MAXLEN, "maximum cluster size is greater than %u (sizeMaxBuf): a cluster crosses shard borders", LIME_MAX_CLUSTER);
                open = false;
            }
        }
        if (e & LIME_EDGE_OPEN) { open = true; r = (e & LIME_EDGE_OPEN_R) != 0; g = (e & LIME_EDGE_OPEN_G) != 0; }
    }
    return LIME_OK;
}

extern "C" int lime_detect_dev(lime_ctx *c, const uint32_t *d_lcp, const uint32_t *d_da, uint64_t n_own,
                               uint64_t n_avail, int eof, uint64_t pos_base, uint32_t n_reads, uint32_t alpha,
                               const lime_cluster_t **d_clusters, uint64_t *n_clusters, uint64_t *max_len,
                               void *stream)
{
    int rc = check_ctx(c, "lime_detect_dev"); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!d_clusters || !n_clusters || !max_len) return fail(LIME_ERR_ARG, "lime_detect_dev: NULL output");
    *d_clusters = nullptr; *n_clusters = 0; *max_len = 0;
    if (n_own > n_avail) return fail(LIME_ERR_ARG, "lime_detect_dev: n_own > n_avail");
    if (n_avail && (!d_lcp || !d_da)) return fail(LIME_ERR_ARG, "lime_detect_dev: NULL array");
    if (misaligned(d_lcp, 16) || misaligned(d_da, 16))
        return fail(LIME_ERR_ARG, "lime_detect_dev: device arrays must be 16-byte aligned");
    if (!n_avail) return LIME_OK;
    if ((rc = ensure_scratch(c, n_avail, true, false, st))) return rc;
    c->last.valid = false;
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(DevStats), st));
    ScanArgs a = base_args(c, d_lcp, d_da, nullptr, n_own, n_avail, eof, n_reads, 1, alpha, nullptr);
    a.pos_base = pos_base;
    launch_tile(0, 1, a, c->max_blocks, st);
    launch_resolve(1, a, st);
    launch_scan_tiles(c->d_tile_cnt, c->d_tile_off, a.n_tiles, c->d_total, st);
    HIP_TRY(hipGetLastError());
    unsigned long long total = 0;
    lime_stats_t s;
    HIP_TRY(hipMemcpyAsync(&total, c->d_total, sizeof total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&s, c->d_stats, sizeof s, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = flags_to_rc(s.flags & LIME_FLAG_HALO))) return rc;
    if (total != s.n_clusters) return fail(LIME_ERR_HIP, "internal: record count %llu != counter %llu", total, (unsigned long long)s.n_clusters);
    if (total > c->out_cap) {
        size_t cap = (size_t)total + (size_t)total / 8 + 1024;
        if ((rc = regrow(c->d_out, cap))) return rc;
        c->out_cap = cap;
    }
    if (total) {
        a.out = c->d_out;
        launch_emit(a, st);
        HIP_TRY(hipGetLastError());
    }
    *d_clusters = c->d_out; *n_clusters = total; *max_len = s.max_len;
    return LIME_OK;
}

// pos_base: the collection position of d_da[0] (the records' pStart are collection positions)
static int score_dev_impl(lime_ctx *c, const uint32_t *d_da, const uint8_t *d_ebwt, uint64_t n,
                          const lime_cluster_t *d_clusters, uint64_t n_clusters, uint32_t n_reads,
                          uint32_t n_refs, uint8_t *d_sim, int zero_sim, uint64_t pos_base, hipStream_t st)
{
    int rc;
    if (!d_sim || (n && !d_da) || (n_clusters && !d_clusters)) return fail(LIME_ERR_ARG, "lime_score_dev: NULL array");
    if (misaligned(d_sim, 4)) return fail(LIME_ERR_ARG, "lime_score_dev: d_sim must be 4-byte aligned");
    if (!n_reads || !n_refs) return fail(LIME_ERR_ARG, "lime_score_dev: n_reads and n_refs must be > 0");
    if (n_refs >= MAX_REFS || (uint64_t)n_reads + n_refs > 0xFFFFFFF0ull)
        return fail(LIME_ERR_ARG, "lime_score_dev: n_refs must be < 2^%u and n_reads + n_refs <= 2^32 - 16", T_SHIFT);
    if ((rc = ensure_scratch(c, n, false, true, st))) return rc;
    // every listed cluster longer than the in-tile limit lands in the big list
    if (n_clusters + 16 > c->big_cap) {
        if (n_clusters + 16 > 0xFFFFFFF0ull) return fail(LIME_ERR_ARG, "too many clusters for one call");
        HIP_TRY(hipStreamSynchronize(st));
        if ((rc = regrow(c->d_big, (size_t)n_clusters + 16))) return rc;
        c->big_cap = (uint32_t)(n_clusters + 16);
    }
    c->last.valid = false;
    const int ebwt = d_ebwt != nullptr;
    const size_t sim_bytes = lime_sim_bytes(n_reads, n_refs);
    uint64_t batches = (n_clusters + 255) / 256;          // 64 clusters per wave, 4 waves per workgroup
    // Binned updates for the list flow too (round 4; ClusterBWT_DA.cpp:301-340 with the arrays resident): the table is built from
    // scratch (zero_sim) and 16-byte aligned.  A pool that proves too small (the list says nothing about its update density) is found out right here --
    // the call waits for the pass -- and the list is scored again by compare-and-swap.
    bool binned = zero_sim && n_clusters && !pos_base && !misaligned(d_sim, 16) && sim_bytes >= (1u << 20) &&
                  sim_bytes <= ((size_t)BIN_MAX << BIN_SHIFT_MAX) && sim_bytes < (1ull << CELL_BITS) && sim_bytes <= ((uint64_t)MAX_SUB << 32) &&
                  c->upd_pref == 1;
    // (only when asked for, LIME_UPDATE_PATH=bin: measured with the arrays resident -- tools/bench_list.py, clustered generator, 306 MB table -- the
    // list flow is bound by its per-cluster gather of da / ebwt, not by its updates: 1e8 symbols, 1.7e7 updates 1.09 ms by compare-and-swap
    // against 1.33 binned; 4e8 symbols, 6.6e7 updates 4.25 against 4.56)
    for (int attempt = 0; attempt < 2; ++attempt) {
        {
            const bool zt = zero_sim && !binned && d_sim;
            launch_zero2(c->d_stats, sizeof(DevStats), zt ? d_sim : nullptr, zt ? (sim_bytes & ~(size_t)15) : 0, st);
            if (zt && (sim_bytes & 15)) HIP_TRY(hipMemsetAsync(d_sim + (sim_bytes & ~(size_t)15), 0, sim_bytes & 15, st));
        }
        if (!n_clusters) return LIME_OK;
        ScanArgs a = base_args(c, nullptr, d_da, d_ebwt, n, n, 1, n_reads, n_refs, 0, d_sim);
        a.pos_base = pos_base;
        uint32_t blocks = (uint32_t)(batches < c->list_blocks ? batches : c->list_blocks);
        uint32_t n_bins = 0, bin_shift = REGION_SHIFT, n_sub = 1, cap_w = 0;
        bool p64 = false;
        if (binned) {
            if (blocks > 1024u) blocks = 1024u;           // fewer, longer producers: a partition workgroup per scoring workgroup
            bin_layout(c, sim_bytes, &n_bins, &bin_shift);
            n_sub = (uint32_t)((sim_bytes + 0xFFFFFFFFull) >> 32);
            if ((rc = ensure_binned(c, n, blocks * (SCAN_WG / 64), blocks, n_bins, bin_shift, n_sub, sub_share(sim_bytes), &cap_w, &p64, st))) return rc;
            a.upd_mode = 1; a.pool = c->d_pool; a.cap_w = cap_w; a.n_sub = n_sub; a.wave_cnt = c->d_wave_cnt; a.counts = c->d_counts;
            a.n_bins = n_bins; a.bin_shift = bin_shift; a.prod_waves = SCAN_WG / 64;
            a.sub_rb = 0xFFFFFFFFu; a.sub_gb = 0u;
            if (n_sub == 2) { a.sub_rb = (uint32_t)((1ull << 32) / n_refs); a.sub_gb = (uint32_t)((1ull << 32) - (uint64_t)a.sub_rb * n_refs); }
        }
        launch_score_list(ebwt, a, d_clusters, n_clusters, blocks, st);
        if (binned) {
            launch_bin_rowscan(c->d_counts, c->d_totals, n_bins, blocks, st);
            launch_scan_tiles(c->d_totals, c->d_binbase, n_bins, reinterpret_cast<unsigned long long *>(c->d_binbase + n_bins), st);
            launch_part(a, blocks, c->d_binbase, c->d_recs, st, p64, c->part_lines != 0);
            if (bin_shift > REGION_SHIFT) launch_apply_by_tiles(d_sim, sim_bytes, c->d_recs, c->d_binbase, n_bins, bin_shift, c->d_tbase, c->d_tidx,
                                                                reinterpret_cast<uint16_t *>(c->d_pool), many_records_of(c, 0.0), st, big_rows_of(c, 0.0));
            else launch_apply(d_sim, sim_bytes, c->d_recs, c->d_binbase, bin_shift, st);
        }
        launch_score_big(ebwt, a, c->d_big_scratch, st);  // (after the table is built: its compare-and-swaps add to it)
        HIP_TRY(hipGetLastError());
        if (!binned) break;
        lime_stats_t s;
        if ((rc = read_stats(c, &s, st))) return rc;      // waits for the pass
        if (!(s.flags & LIME_FLAG_POOL_FULL)) break;
        binned = false;                                   // the table is incomplete: again, by compare-and-swap
    }
    return LIME_OK;
}

extern "C" int lime_score_dev(lime_ctx *c, const uint32_t *d_da, const uint8_t *d_ebwt, uint64_t n,
                              const lime_cluster_t *d_clusters, uint64_t n_clusters, uint32_t n_reads,
                              uint32_t n_refs, uint8_t *d_sim, int zero_sim, void *stream)
{
    int rc = check_ctx(c, "lime_score_dev"); if (rc) return rc;
    return score_dev_impl(c, d_da, d_ebwt, n, d_clusters, n_clusters, n_reads, n_refs, d_sim, zero_sim, 0, (hipStream_t)stream);
}

extern "C" int lime_choose_dev(lime_ctx *c, const uint8_t *d_sim, uint32_t n_reads, uint32_t n_refs,
                               uint8_t *d_row_max, uint32_t *d_row_nnz, void *stream)
{
    int rc = check_ctx(c, "lime_choose_dev"); if (rc) return rc;
    if (!d_sim || !d_row_max || !d_row_nnz) return fail(LIME_ERR_ARG, "lime_choose_dev: NULL array");
    if (misaligned(d_sim, 16)) return fail(LIME_ERR_ARG, "lime_choose_dev: d_sim must be 16-byte aligned (and lime_sim_bytes() long)");
    if (!n_reads) return LIME_OK;
    launch_choose(d_sim, n_reads, n_refs, d_row_max, d_row_nnz, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return LIME_OK;
}

extern "C" int lime_synth_dev(lime_ctx *c, uint64_t seed, uint64_t i0, uint64_t count, uint32_t n_reads,
                              uint32_t n_refs, uint32_t alpha, uint32_t mode, uint32_t *d_lcp, uint32_t *d_da,
                              uint8_t *d_ebwt, void *stream)
{
    int rc = check_ctx(c, "lime_synth_dev"); if (rc) return rc;
    if (!n_reads || !n_refs) return fail(LIME_ERR_ARG, "lime_synth_dev: n_reads and n_refs must be > 0");
    if (!count) return LIME_OK;
    launch_synth(seed, i0, count, n_reads, n_refs, alpha, mode, d_lcp, d_da, d_ebwt, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return LIME_OK;
}

// ---- host-pointer API: stage through HBM, run the device path, copy back ------------------
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) dev_release(p); }                  // (large ones stay in the process's block cache)
    int alloc(size_t bytes) { HIP_TRY(dev_acquire(&p, bytes ? bytes : 16)); return LIME_OK; }
    int upload(const void *src, size_t bytes) {
        int rc = alloc(bytes + 16); if (rc) return rc;
        if (bytes) HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return LIME_OK;
    }
};
}

// ---- streaming from host memory: the collection goes through HBM in position-range chunks ------
// Chunk k owns positions [k*chunk, (k+1)*chunk) and carries a read-ahead halo, exactly like a shard of
// the multi-GPU partition (ClusterLCP.cpp:150-161,246-264); two device buffers alternate so that the
// copy of chunk k+1 runs while chunk k is scanned; all chunks add into one table in HBM.
namespace {
struct Pipe {
    hipStream_t copy = nullptr, comp = nullptr;
    hipEvent_t copied[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    ~Pipe() {
        for (int b = 0; b < 2; ++b) { if (copied[b]) (void)hipEventDestroy(copied[b]); if (consumed[b]) (void)hipEventDestroy(consumed[b]); }
        if (copy) (void)hipStreamDestroy(copy);
        if (comp) (void)hipStreamDestroy(comp);
    }
    int init() {
        HIP_TRY(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(hipEventCreateWithFlags(&copied[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&consumed[b], hipEventDisableTiming));
        }
        return LIME_OK;
    }
};
const uint64_t STREAM_HALO = (uint64_t)LIME_MAX_CLUSTER + LIME_TILE;   // a run the reference accepts closes inside it
const uint64_t STREAM_CHUNK = 64ull << 20;                              // default symbols per chunk, sources pinned (or small)
const uint64_t STAGED_CHUNK = 4ull << 20;                               // through the pinned ring: short chunks fill the pipeline sooner and the
                                                                        // three slots stay small (measured best of 4 / 16 / 64 Mi on 10^9 symbols)
}

// ---- host arrays -> HBM, chunk by chunk ---------------------------------------------------------
// The callers' arrays are usually pageable (the drop-in programs hand over mmap-ed files): a copy straight from
// them is synchronous and staged by the runtime in small pieces.  The Feeder moves chunk k through a ring of three
// pinned slots instead: a producer thread (with LIME_IO_THREADS helpers) copies the chunk's pieces into slot k % 3
// while the copy engine empties slot (k-1) % 3 into one device buffer set and the kernels work on the other.
// Sources that are already pinned (hipHostMalloc / hipHostRegister) skip the ring.
// ---- mapped files the callers hand over as arrays (the drop-in programs): lime_register_file tells the library which file a mapping shows, and
// the staging threads then fill the pinned slots with pread() from the file -- the page cache copied by the kernel at several GB/s per thread --
// instead of memcpy from the mapping, which takes a page fault per 4 KB (16-page fault-around) of every first touch: 10.6 GB/s with 8 threads in
// round 4, a fifth of what the link takes from pinned memory.  Reference: the per-thread FILE* reads of ClusterLCP.cpp:100-123, 206-212.
namespace {
struct FileMap { const char *base; size_t bytes; int fd; };
std::mutex g_files_mu;
std::vector<FileMap> g_files;
// bytes [src, src + len) from the registered file that holds them (false: not in one -- the caller copies from memory)
bool read_from_file(void *dst, const void *src, size_t len)
{
    FileMap m{nullptr, 0, -1};
    {
        std::lock_guard<std::mutex> g(g_files_mu);
        for (const FileMap &f : g_files)
            if ((const char *)src >= f.base && (const char *)src + len <= f.base + f.bytes) { m = f; break; }
    }
    if (m.fd < 0) return false;
    size_t done = 0;
    const off_t at = (off_t)((const char *)src - m.base);
    while (done < len) {
        const ssize_t k = pread(m.fd, (char *)dst + done, len - done, at + (off_t)done);
        if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
        done += (size_t)k;
    }
    return true;
}
}
extern "C" int lime_register_file(const void *base, size_t bytes, int fd)
{
    if (!base || fd < 0) return fail(LIME_ERR_ARG, "lime_register_file: bad argument");
    // (the contract is in include/lime_hip.h: the mapping shows the file from offset 0, read-only; what can be checked is)
    struct stat sb;
    if (fstat(fd, &sb) != 0) return fail(LIME_ERR_IO, "lime_register_file: fstat failed");
    if ((uint64_t)bytes > (uint64_t)sb.st_size) return fail(LIME_ERR_ARG, "lime_register_file: %zu bytes registered, the file has %lld", bytes, (long long)sb.st_size);
    std::lock_guard<std::mutex> g(g_files_mu);
    for (const FileMap &f : g_files)
        if ((const char *)base < f.base + f.bytes && f.base < (const char *)base + bytes)
            return fail(LIME_ERR_ARG, "lime_register_file: the range overlaps a registered one (unregister before unmapping)");
    const int own = dup(fd);                               // the caller may close its descriptor
    if (own < 0) return fail(LIME_ERR_IO, "lime_register_file: dup failed");
    g_files.push_back(FileMap{(const char *)base, bytes, own});
    return LIME_OK;
}
extern "C" void lime_unregister_file(const void *base)
{
    std::lock_guard<std::mutex> g(g_files_mu);
    for (size_t i = 0; i < g_files.size(); ++i)
        if (g_files[i].base == (const char *)base) { close(g_files[i].fd); g_files.erase(g_files.begin() + (long)i); return; }
}

namespace {
struct Piece { const void *src; size_t bytes; void *dst; };

// a few host threads that copy (or pread) pieces of 1 MB: created once per walk, not per piece
struct IoPool {
    struct Task { void *dst; const void *src; size_t len; };
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv_go, cv_done;
    std::vector<Task> tasks; size_t next = 0, left = 0; bool stop = false;
    void start(int n)
    {
        for (int t = 0; t < n; ++t)
            th.emplace_back([this]() {
                for (;;) {
                    Task k;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_go.wait(lk, [&] { return stop || next < tasks.size(); });
                        if (stop) return;
                        k = tasks[next++];
                    }
                    if (!read_from_file(k.dst, k.src, k.len)) memcpy(k.dst, k.src, k.len);
                    { std::lock_guard<std::mutex> g(mu); if (--left == 0) cv_done.notify_all(); }
                }
            });
    }
    void run(std::vector<Task> &&t)                        // returns when every task is done
    {
        if (t.empty()) return;
        if (th.empty()) { for (const Task &k : t) if (!read_from_file(k.dst, k.src, k.len)) memcpy(k.dst, k.src, k.len); return; }
        std::unique_lock<std::mutex> lk(mu);
        tasks = std::move(t); next = 0; left = tasks.size();
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return left == 0; });
        tasks.clear(); next = 0;
    }
    ~IoPool()
    {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
};

// host threads that move pageable sources into the pinned ring: what the ctx says (lime_set_option "io_threads": the drop-in programs pass the
// reference's `threads` argument, LIME_IO_THREADS overrides at lime_init), else 8; never more than the CPUs this process may run on (its
// affinity mask: a cgroup- or taskset-limited job must not be oversubscribed -- ADVICE r5)
static int staging_threads(const lime_ctx *c)
{
    int t = c && c->io_threads > 0 ? c->io_threads : 8;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int n = CPU_COUNT(&set); if (n >= 1 && t > n) t = n; }
    else { const unsigned hw = std::thread::hardware_concurrency(); if (hw && (unsigned)t > hw) t = (int)hw; }
    return t < 1 ? 1 : t;
}

struct Feeder {
    static constexpr int NS = 3, MAXP = 3;
    void *slot[NS] = {nullptr, nullptr, nullptr};
    hipEvent_t h2d_done[NS] = {nullptr, nullptr, nullptr};
    size_t slot_bytes = 0;
    bool staged = false;
    int io_threads = 1, device = 0;
    uint64_t n_chunks = 0;
    std::function<int(uint64_t, Piece *)> describe;       // pieces of chunk k (at most MAXP); returns their number
    std::thread producer;
    std::mutex mu; std::condition_variable cv;
    uint64_t filled = 0, issued = 0; bool stop = false, failed = false;

    // would a walk over `total_bytes` of these sources go through the ring?  (the caller picks its chunk size by it)
    static bool will_stage(const lime_ctx *c, bool all_sources_pinned, size_t total_bytes)
    {
        return !all_sources_pinned && !(c && c->no_staging) && (total_bytes >= ((size_t)8 << 20) || (c && c->force_staging));
    }
    static bool pinned(const void *p)
    {
        if (!p) return true;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    }
    IoPool pool;
    int init(const lime_ctx *c, size_t bytes_per_chunk, uint64_t chunks, bool all_sources_pinned, std::function<int(uint64_t, Piece *)> d)
    {
        describe = std::move(d); n_chunks = chunks; slot_bytes = bytes_per_chunk;
        // small collections: the ring's set-up (pinned allocations, a thread) costs more than it hides
        staged = will_stage(c, all_sources_pinned, bytes_per_chunk * chunks);
        if (!staged) return LIME_OK;
        HIP_TRY(hipGetDevice(&device));
        // staging threads: the ctx's io_threads (the drop-in programs pass their `threads` argument; LIME_IO_THREADS at lime_init), else 8 -- never more
        // than the CPUs this process may run on
        io_threads = staging_threads(c);
        if (io_threads > 1) pool.start(io_threads);
        for (int i = 0; i < NS; ++i) {
            HIP_TRY(hipHostMalloc(&slot[i], slot_bytes + 64, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&h2d_done[i], hipEventDisableTiming));
        }
        producer = std::thread([this]() {
            (void)hipSetDevice(device);
            Piece pc[MAXP];
            for (uint64_t k = 0; k < n_chunks; ++k) {
                if (k >= NS) {                               // slot k % NS: its previous content must have left for the device
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return issued > k - NS || stop; });
                    if (stop) return;
                    lk.unlock();
                    if (hipEventSynchronize(h2d_done[k % NS]) != hipSuccess) { std::lock_guard<std::mutex> g(mu); failed = true; cv.notify_all(); return; }
                }
                { std::lock_guard<std::mutex> g(mu); if (stop) return; }
                const int np = describe(k, pc);
                size_t off = 0;
                std::vector<IoPool::Task> tk;
                for (int i = 0; i < np; ++i) {
                    for (size_t o = 0; o < pc[i].bytes; o += (size_t)1 << 20)
                        tk.push_back(IoPool::Task{(char *)slot[k % NS] + off + o, (const char *)pc[i].src + o, pc[i].bytes - o < ((size_t)1 << 20) ? pc[i].bytes - o : (size_t)1 << 20});
                    off += (pc[i].bytes + 15) & ~(size_t)15;
                }
                pool.run(std::move(tk));
                { std::lock_guard<std::mutex> g(mu); filled = k + 1; }
                cv.notify_all();
            }
        });
        return LIME_OK;
    }
    // the copies of chunk k, asynchronous on `copy` (chunks must be fed in order)
    int feed(uint64_t k, hipStream_t copy)
    {
        Piece pc[MAXP];
        const int np = describe(k, pc);
        if (!staged) {
            for (int i = 0; i < np; ++i) HIP_TRY(hipMemcpyAsync(pc[i].dst, pc[i].src, pc[i].bytes, hipMemcpyHostToDevice, copy));
            return LIME_OK;
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return filled > k || failed; });
            if (failed) return fail(LIME_ERR_HIP, "staging thread failed");
        }
        size_t off = 0;
        for (int i = 0; i < np; ++i) {
            HIP_TRY(hipMemcpyAsync(pc[i].dst, (char *)slot[k % NS] + off, pc[i].bytes, hipMemcpyHostToDevice, copy));
            off += (pc[i].bytes + 15) & ~(size_t)15;
        }
        HIP_TRY(hipEventRecord(h2d_done[k % NS], copy));
        { std::lock_guard<std::mutex> g(mu); issued = k + 1; }
        cv.notify_all();
        return LIME_OK;
    }
    ~Feeder()
    {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv.notify_all();
        if (producer.joinable()) producer.join();
        for (int i = 0; i < NS; ++i) { if (h2d_done[i]) { (void)hipEventSynchronize(h2d_done[i]); (void)hipEventDestroy(h2d_done[i]); } if (slot[i]) (void)hipHostFree(slot[i]); }
    }
};
}

// Host arrays -> device buffers through the pinned staging ring, asynchronous pieces on `st`, complete on return
// (lime_fused_multi: one host thread per device calls this, so k devices upload at k times the rate of one).
int lime_internal_upload(int n_arr, const void *const *src, void *const *dst, const size_t *bytes, hipStream_t st)
{
    const size_t CH = (size_t)32 << 20;                   // bytes per array and chunk
    size_t total = 0, longest = 0;
    bool pinned = true;
    for (int i = 0; i < n_arr; ++i) { total += bytes[i]; if (bytes[i] > longest) longest = bytes[i]; pinned = pinned && Feeder::pinned(src[i]); }
    if (!total) return LIME_OK;
    if (n_arr > Feeder::MAXP) return fail(LIME_ERR_ARG, "lime_internal_upload: too many arrays");
    const uint64_t n_chunks = (longest + CH - 1) / CH;
    Feeder feeder;
    int rc = feeder.init(nullptr, (size_t)n_arr * (CH + 16), n_chunks, pinned, [&](uint64_t k, Piece *pc) {
        int np = 0;
        for (int i = 0; i < n_arr; ++i) {
            const size_t off = (size_t)k * CH;
            if (off >= bytes[i]) continue;
            pc[np++] = Piece{(const char *)src[i] + off, bytes[i] - off < CH ? bytes[i] - off : CH, (char *)dst[i] + off};
        }
        return np;
    });
    if (rc) return rc;
    for (uint64_t k = 0; k < n_chunks; ++k) if ((rc = feeder.feed(k, st))) { (void)hipStreamSynchronize(st); return rc; }
    HIP_TRY(hipStreamSynchronize(st));
    return LIME_OK;
}

extern "C" int lime_fused_stream(lime_ctx *c, const uint32_t *lcp, const uint32_t *da, const uint8_t *ebwt, uint64_t n,
                                 uint32_t n_reads, uint32_t n_refs, uint32_t alpha, uint64_t chunk, uint8_t *sim,
                                 uint64_t *n_clusters, uint64_t *max_len)
{
    int rc = check_ctx(c, "lime_fused_stream"); if (rc) return rc;
    if (!sim || (n && (!lcp || !da))) return fail(LIME_ERR_ARG, "lime_fused_stream: NULL array");
    if (!n_reads || !n_refs) return fail(LIME_ERR_ARG, "lime_fused_stream: n_reads and n_refs must be > 0");
    const bool src_pinned = Feeder::pinned(lcp) && Feeder::pinned(da) && Feeder::pinned(ebwt);
    if (!chunk) chunk = Feeder::will_stage(c, src_pinned, (size_t)n * 9) ? STAGED_CHUNK : STREAM_CHUNK;
    chunk = (chunk + LIME_TILE - 1) / LIME_TILE * LIME_TILE;
    const uint64_t cap = (chunk < n ? chunk : n) + STREAM_HALO;         // elements per device buffer
    Pipe pp;
    if ((rc = pp.init())) return rc;
    DevBuf dl[2], dd[2], de[2], ds;
    const int nbuf = n > chunk ? 2 : 1;
    for (int b = 0; b < nbuf; ++b) {
        if ((rc = dl[b].alloc(cap * 4 + 16))) return rc;
        if ((rc = dd[b].alloc(cap * 4 + 16))) return rc;
        if (ebwt && (rc = de[b].alloc(cap + 16))) return rc;
    }
    if ((rc = ds.alloc(lime_sim_bytes(n_reads, n_refs)))) return rc;
    if (!n) HIP_TRY(hipMemsetAsync(ds.p, 0, lime_sim_bytes(n_reads, n_refs), pp.comp));
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    DevBuf dedge;                                          // one edge word per chunk (runs longer than the halo across chunk borders)
    if ((rc = dedge.alloc((size_t)(n_chunks + 1) * 4))) return rc;
    HIP_TRY(hipMemsetAsync(dedge.p, 0, (size_t)(n_chunks + 1) * 4, pp.comp));
    Feeder feeder;
    if ((rc = feeder.init(c, cap * 9 + 64, n_chunks, src_pinned,
                          [&](uint64_t kk, Piece *pc) {
                              const uint64_t lo = kk * chunk, own = n - lo < chunk ? n - lo : chunk;
                              const uint64_t avail = n - lo < own + STREAM_HALO ? n - lo : own + STREAM_HALO;
                              const int b = (int)(kk & 1);
                              pc[0] = Piece{lcp + lo, (size_t)avail * 4, dl[b].p}; pc[1] = Piece{da + lo, (size_t)avail * 4, dd[b].p};
                              if (ebwt) pc[2] = Piece{ebwt + lo, (size_t)avail, de[b].p};
                              return ebwt ? 3 : 2;
                          }))) return rc;
    uint64_t k = 0;
    for (uint64_t lo = 0; lo < n; lo += chunk, ++k) {
        const int b = (int)(k & 1);
        const uint64_t own = n - lo < chunk ? n - lo : chunk;
        const uint64_t avail = n - lo < own + STREAM_HALO ? n - lo : own + STREAM_HALO;
        const int eof = lo + avail == n;
        if (k >= 2) HIP_TRY(hipStreamWaitEvent(pp.copy, pp.consumed[b], 0));        // the buffer is free again
        if ((rc = feeder.feed(k, pp.copy))) { (void)hipDeviceSynchronize(); return rc; }
        HIP_TRY(hipEventRecord(pp.copied[b], pp.copy));
        HIP_TRY(hipStreamWaitEvent(pp.comp, pp.copied[b], 0));
        rc = fused_dev_impl(c, (const uint32_t *)dl[b].p, (const uint32_t *)dd[b].p, ebwt ? (const uint8_t *)de[b].p : nullptr,
                            own, avail, eof, n_reads, n_refs, alpha, (uint8_t *)ds.p, k == 0, k != 0, pp.comp, (uint32_t *)dedge.p + k,
                            n_chunks > 1);
        if (rc) { (void)hipDeviceSynchronize(); return rc; }
        HIP_TRY(hipEventRecord(pp.consumed[b], pp.comp));
    }
    lime_stats_t s;
    rc = lime_get_stats(c, &s, pp.comp);
    if (n_clusters) *n_clusters = n ? s.n_clusters : 0;
    if (max_len) *max_len = n ? s.max_len : 0;
    if (rc) { (void)hipDeviceSynchronize(); return rc; }
    if (n_chunks > 1) {
        std::vector<uint32_t> edges(n_chunks);
        HIP_TRY(hipMemcpyAsync(edges.data(), dedge.p, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, pp.comp));
        HIP_TRY(hipStreamSynchronize(pp.comp));
        if ((rc = lime_combine_edges(edges.data(), (uint32_t)n_chunks))) { (void)hipDeviceSynchronize(); return rc; }
    }
    if ((rc = d2h_pageable(c, sim, ds.p, (size_t)n_reads * n_refs, pp.comp))) return rc;
    HIP_TRY(hipStreamSynchronize(pp.comp));
    HIP_TRY(hipStreamSynchronize(pp.copy));
    return LIME_OK;
}

// The walk of lime_detect / lime_detect_to_file: position-range chunks with a read-ahead halo (see lime_fused_stream); records of chunk k follow
// those of chunk k-1, so the list stays in ascending pStart = the reference's 1-thread order.  The copy of chunk k+1 (through the pinned ring
// when the arrays are pageable) is under way while chunk k is scanned.  sink(device records, count, stream): takes a chunk's records (the list
// of the ctx is reused by the next chunk: the walk waits for the stream after every sink call).
static int detect_walk(lime_ctx *c, const uint32_t *lcp, const uint32_t *da, uint64_t n, uint32_t n_reads, uint32_t alpha, uint64_t chunk,
                       uint64_t *max_len, const std::function<int(const lime_cluster_t *, uint64_t, hipStream_t)> &sink)
{
    chunk = (chunk + LIME_TILE - 1) / LIME_TILE * LIME_TILE;
    const uint64_t cap = (chunk < n ? chunk : n) + STREAM_HALO, n_chunks = (n + chunk - 1) / chunk;
    int rc;
    uint64_t ml = 0;
    Pipe pp;
    if ((rc = pp.init())) return rc;
    DevBuf dl[2], dd[2];
    for (int b = 0; b < (n_chunks > 1 ? 2 : 1); ++b) { if ((rc = dl[b].alloc(cap * 4 + 16))) return rc; if ((rc = dd[b].alloc(cap * 4 + 16))) return rc; }
    Feeder feeder;
    if ((rc = feeder.init(c, cap * 8 + 64, n_chunks, Feeder::pinned(lcp) && Feeder::pinned(da), [&](uint64_t kk, Piece *pc) {
            const uint64_t lo = kk * chunk, own = n - lo < chunk ? n - lo : chunk;
            const uint64_t avail = n - lo < own + STREAM_HALO ? n - lo : own + STREAM_HALO;
            pc[0] = Piece{lcp + lo, (size_t)avail * 4, dl[kk & 1].p}; pc[1] = Piece{da + lo, (size_t)avail * 4, dd[kk & 1].p};
            return 2;
        }))) return rc;
    if ((rc = feeder.feed(0, pp.copy))) return rc;
    HIP_TRY(hipEventRecord(pp.copied[0], pp.copy));
    for (uint64_t k = 0; k < n_chunks; ++k) {
        const int b = (int)(k & 1);
        const uint64_t lo = k * chunk, own = n - lo < chunk ? n - lo : chunk;
        const uint64_t avail = n - lo < own + STREAM_HALO ? n - lo : own + STREAM_HALO;
        if (k + 1 < n_chunks) {                       // the next chunk's copy goes out before this chunk's scan is waited for
            const int nb = (int)((k + 1) & 1);
            if (k + 1 >= 2) HIP_TRY(hipStreamWaitEvent(pp.copy, pp.consumed[nb], 0));
            if ((rc = feeder.feed(k + 1, pp.copy))) { (void)hipDeviceSynchronize(); return rc; }
            HIP_TRY(hipEventRecord(pp.copied[nb], pp.copy));
        }
        HIP_TRY(hipStreamWaitEvent(pp.comp, pp.copied[b], 0));
        const lime_cluster_t *dc = nullptr;
        uint64_t cnt = 0, m = 0;
        rc = lime_detect_dev(c, (const uint32_t *)dl[b].p, (const uint32_t *)dd[b].p, own, avail, lo + avail == n, lo, n_reads,
                             alpha, &dc, &cnt, &m, pp.comp);
        if (rc) { (void)hipDeviceSynchronize(); return rc; }
        if (m > ml) ml = m;
        HIP_TRY(hipEventRecord(pp.consumed[b], pp.comp));   // (the arrays of the chunk have been read: lime_detect_dev waited for its count)
        if (cnt && (rc = sink(dc, cnt, pp.comp))) { (void)hipDeviceSynchronize(); return rc; }
        HIP_TRY(hipStreamSynchronize(pp.comp));           // the record list of the ctx is reused by the next chunk
    }
    HIP_TRY(hipStreamSynchronize(pp.copy));
    *max_len = ml;
    return LIME_OK;
}

static uint64_t detect_chunk(const lime_ctx *c, const uint32_t *lcp, const uint32_t *da, uint64_t n)
{
    if (c->detect_chunk) return c->detect_chunk;
    return Feeder::will_stage(c, Feeder::pinned(lcp) && Feeder::pinned(da), (size_t)n * 8) ? STAGED_CHUNK : STREAM_CHUNK;
}

// one walk with chunks of `chunk` symbols, the records gathered in host memory (*clusters: malloc'ed, the caller's)
static int detect_to_host(lime_ctx *c, const uint32_t *lcp, const uint32_t *da, uint64_t n, uint32_t n_reads, uint32_t alpha, uint64_t chunk,
                          lime_cluster_t **clusters, uint64_t *n_clusters, uint64_t *max_len)
{
    lime_cluster_t *h = nullptr;
    uint64_t have = 0, room = 0, ml = 0;
    auto sink = [&](const lime_cluster_t *dc, uint64_t cnt, hipStream_t st) -> int {
        if (have + cnt > room) {
            room = (have + cnt) + (have + cnt) / 2 + 1024;
            lime_cluster_t *g = (lime_cluster_t *)realloc(h, (size_t)room * sizeof(lime_cluster_t));
            if (!g) return fail(LIME_ERR_NOMEM, "lime_detect: out of host memory");
            h = g;
        }
        HIP_TRY(hipMemcpyAsync(h + have, dc, (size_t)cnt * sizeof(lime_cluster_t), hipMemcpyDeviceToHost, st));
        have += cnt;
        return LIME_OK;
    };
    const int rc = detect_walk(c, lcp, da, n, n_reads, alpha, chunk, &ml, sink);
    if (rc) { free(h); return rc; }
    *clusters = h; *n_clusters = have; *max_len = ml;
    return LIME_OK;
}

extern "C" int lime_detect(lime_ctx *c, const uint32_t *lcp, const uint32_t *da, uint64_t n, uint32_t n_reads,
                           uint32_t alpha, lime_cluster_t **clusters, uint64_t *n_clusters, uint64_t *max_len)
{
    int rc = check_ctx(c, "lime_detect"); if (rc) return rc;
    if (!clusters || !n_clusters || !max_len) return fail(LIME_ERR_ARG, "lime_detect: NULL output");
    *clusters = nullptr; *n_clusters = 0; *max_len = 0;
    if (n && (!lcp || !da)) return fail(LIME_ERR_ARG, "lime_detect: NULL array");
    if (!n) return LIME_OK;
    const uint64_t chunk0 = detect_chunk(c, lcp, da, n);
    rc = detect_to_host(c, lcp, da, n, n_reads, alpha, chunk0, clusters, n_clusters, max_len);
    if (rc == LIME_ERR_HALO && chunk0 < n)
        // a run longer than the halo crosses a chunk border.  ClusterLCP itself has no length limit (only
        // ClusterBWT_DA refuses such a cluster later): redo the whole collection as one chunk
        rc = detect_to_host(c, lcp, da, n, n_reads, alpha, n, clusters, n_clusters, max_len);
    return rc;
}

// lime_detect with the records written to `path` as they come (the .clrs file of ClusterLCP.cpp:229-235, ascending pStart): every chunk's
// records land in one of two pinned buffers and a writer thread appends them to the file while the next chunk is scanned -- no list of the
// whole collection on the host (10^9 symbols: 0.9 GB that round 4 grew by realloc, copied through a pageable buffer and wrote at the end).
extern "C" int lime_detect_to_file(lime_ctx *c, const uint32_t *lcp, const uint32_t *da, uint64_t n, uint32_t n_reads,
                                   uint32_t alpha, const char *path, uint64_t *n_clusters, uint64_t *max_len)
{
    int rc = check_ctx(c, "lime_detect_to_file"); if (rc) return rc;
    if (!path || !n_clusters || !max_len) return fail(LIME_ERR_ARG, "lime_detect_to_file: NULL argument");
    *n_clusters = 0; *max_len = 0;
    if (n && (!lcp || !da)) return fail(LIME_ERR_ARG, "lime_detect_to_file: NULL array");
    // The records go to a temporary file next to `path` that takes its name only when the scan has succeeded: a failure on the way (pinned
    // memory, a HIP error, a full disk) leaves no truncated .clrs behind for ClusterBWT_DA to read (the reference's fopen "w" +
    // exit(1) does leave one, ClusterLCP.cpp:95-98 -- not a behaviour to keep)
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(LIME_ERR_IO, "lime_detect_to_file: cannot create %s", path);
    if (!n) { close(fd); if (rename(tmp.c_str(), path)) { (void)unlink(tmp.c_str()); return fail(LIME_ERR_IO, "lime_detect_to_file: cannot create %s", path); } return LIME_OK; }
    const uint64_t chunk0 = detect_chunk(c, lcp, da, n);
    const size_t buf_records = (size_t)((chunk0 < n ? chunk0 : n) / 2 + 4096);      // a chunk has at most half as many clusters as positions
    void *pin[2] = {nullptr, nullptr};
    struct Job { int b; uint64_t cnt, at; };
    std::mutex mu; std::condition_variable cv;
    std::vector<Job> jobs; size_t taken = 0, written = 0; bool stop = false, io_failed = false;
    uint64_t have = 0, ml = 0;
    std::thread writer;
    auto finish = [&]() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv.notify_all();
        if (writer.joinable()) writer.join();
        for (int b = 0; b < 2; ++b) if (pin[b]) (void)hipHostFree(pin[b]);
        close(fd);
    };
    if (hipHostMalloc(&pin[0], buf_records * sizeof(lime_cluster_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&pin[1], buf_records * sizeof(lime_cluster_t), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); finish(); (void)unlink(tmp.c_str());
        return fail(LIME_ERR_NOMEM, "lime_detect_to_file: no pinned memory for the record buffers");
    }
    writer = std::thread([&]() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || taken < jobs.size(); });
                if (taken >= jobs.size()) return;
                j = jobs[taken++];
            }
            size_t done = 0; const size_t len = (size_t)j.cnt * sizeof(lime_cluster_t);
            while (done < len) {
                const ssize_t k = pwrite(fd, (const char *)pin[j.b] + done, len - done, (off_t)(j.at * sizeof(lime_cluster_t) + done));
                if (k <= 0) { if (k < 0 && errno == EINTR) continue; std::lock_guard<std::mutex> g(mu); io_failed = true; break; }
                done += (size_t)k;
            }
            { std::lock_guard<std::mutex> g(mu); ++written; }
            cv.notify_all();
        }
    });
    uint64_t n_jobs = 0;
    auto sink = [&](const lime_cluster_t *dc, uint64_t cnt, hipStream_t st) -> int {
        if (cnt > buf_records) return fail(LIME_ERR_HIP, "internal: %llu records in one chunk", (unsigned long long)cnt);
        const int b = (int)(n_jobs & 1);
        {   // the buffer's previous content (two jobs ago) must be in the file
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return n_jobs < 2 || written + 2 > n_jobs || io_failed; });
            if (io_failed) return fail(LIME_ERR_IO, "lime_detect_to_file: write to %s failed", path);
        }
        HIP_TRY(hipMemcpyAsync(pin[b], dc, (size_t)cnt * sizeof(lime_cluster_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        { std::lock_guard<std::mutex> g(mu); jobs.push_back(Job{b, cnt, have}); }
        cv.notify_all();
        have += cnt; ++n_jobs;
        return LIME_OK;
    };
    rc = detect_walk(c, lcp, da, n, n_reads, alpha, chunk0, &ml, sink);
    {   // everything handed over is in the file
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return written == jobs.size() || io_failed; });
    }
    const bool bad_io = io_failed;
    finish();
    if (!rc && bad_io) rc = fail(LIME_ERR_IO, "lime_detect_to_file: write to %s failed", path);
    if (rc == LIME_ERR_HALO && chunk0 < n) {
        // (a run longer than the halo across a chunk border: the whole collection as ONE chunk, through memory -- directly, not through
        // lime_detect, whose first walk would repeat the chunked scan that has just failed)
        lime_cluster_t *h = nullptr;
        if ((rc = detect_to_host(c, lcp, da, n, n_reads, alpha, n, &h, &have, &ml))) { (void)unlink(tmp.c_str()); return rc; }
        rc = lime_write_clrs(tmp.c_str(), h, have);
        free(h);
        if (rc) { (void)unlink(tmp.c_str()); return fail(LIME_ERR_IO, "lime_detect_to_file: write to %s failed", path); }
    }
    if (rc) { (void)unlink(tmp.c_str()); return rc; }
    if (rename(tmp.c_str(), path)) { (void)unlink(tmp.c_str()); return fail(LIME_ERR_IO, "lime_detect_to_file: cannot create %s", path); }
    *n_clusters = have; *max_len = ml;
    return LIME_OK;
}

// clusterAnalyze of a host cluster list into the device table d_sim, the arrays going through HBM
// in position-range chunks: a chunk takes the clusters that START in it (any input order: an index
// sorted by pStart is built when needed) and the arrays up to the end of the last of them.
static int score_in_chunks(lime_ctx *c, const uint32_t *da, const uint8_t *ebwt, uint64_t n,
                           const lime_cluster_t *clusters, uint64_t n_clusters, uint32_t n_reads, uint32_t n_refs,
                           uint8_t *d_sim)
{
    int rc;
    HIP_TRY(hipMemset(d_sim, 0, lime_sim_bytes(n_reads, n_refs)));
    HIP_TRY(hipDeviceSynchronize());                      // the chunks below run on streams of their own
    if (!n_clusters) return LIME_OK;
    uint64_t chunk = c->score_chunk ? c->score_chunk : Feeder::will_stage(c, Feeder::pinned(da) && Feeder::pinned(ebwt), (size_t)n * 5) ? STAGED_CHUNK : STREAM_CHUNK;
    bool sorted = true;
    for (uint64_t i = 0; i < n_clusters; ++i) {
        const lime_cluster_t &q = clusters[i];
        if (q.len > LIME_MAX_CLUSTER) return fail(LIME_ERR_MAXLEN, "a cluster is longer than %u symbols", LIME_MAX_CLUSTER);
        if (q.pStart > n || q.len > n - q.pStart) return fail(LIME_ERR_ARG, "a cluster record lies outside the arrays");
        if (i && q.pStart < clusters[i - 1].pStart) sorted = false;
    }
    std::vector<lime_cluster_t> order;
    const lime_cluster_t *cl = clusters;
    if (!sorted) {
        order.assign(clusters, clusters + n_clusters);
        std::sort(order.begin(), order.end(), [](const lime_cluster_t &x, const lime_cluster_t &y) { return x.pStart < y.pStart; });
        cl = order.data();
    }
    // the plan: chunk k = clusters [i, j) that start in [lo, lo + chunk), arrays [lo, end) up to the end of the last of them
    struct Plan { uint64_t i, j, lo, end; };
    std::vector<Plan> plan;
    size_t max_el = 0, max_cl = 0;
    for (uint64_t i = 0; i < n_clusters;) {
        const uint64_t lo = cl[i].pStart, hi = lo + chunk;
        uint64_t j = i, end = lo;
        while (j < n_clusters && cl[j].pStart < hi) { if (cl[j].pStart + cl[j].len > end) end = cl[j].pStart + cl[j].len; ++j; }
        plan.push_back(Plan{i, j, lo, end});
        if (end - lo > max_el) max_el = (size_t)(end - lo);
        if (j - i > max_cl) max_cl = (size_t)(j - i);
        i = j;
    }
    Pipe pp;
    if ((rc = pp.init())) return rc;
    DevBuf dd[2], de[2], dc[2];
    for (int b = 0; b < (plan.size() > 1 ? 2 : 1); ++b) {
        if ((rc = dd[b].alloc(max_el * 4 + 16))) return rc;
        if (ebwt && (rc = de[b].alloc(max_el + 16))) return rc;
        if ((rc = dc[b].alloc(max_cl * sizeof(lime_cluster_t)))) return rc;
    }
    Feeder feeder;
    if ((rc = feeder.init(c, max_el * 5 + max_cl * sizeof(lime_cluster_t) + 64, plan.size(),
                          Feeder::pinned(da) && Feeder::pinned(ebwt) && Feeder::pinned(cl), [&](uint64_t k, Piece *pc) {
            const Plan &q = plan[k];
            const int b = (int)(k & 1);
            int np = 0;
            pc[np++] = Piece{da + q.lo, (size_t)(q.end - q.lo) * 4, dd[b].p};
            if (ebwt) pc[np++] = Piece{ebwt + q.lo, (size_t)(q.end - q.lo), de[b].p};
            pc[np++] = Piece{cl + q.i, (size_t)(q.j - q.i) * sizeof(lime_cluster_t), dc[b].p};
            return np;
        }))) return rc;
    if ((rc = feeder.feed(0, pp.copy))) return rc;
    HIP_TRY(hipEventRecord(pp.copied[0], pp.copy));
    for (uint64_t k = 0; k < plan.size(); ++k) {
        const int b = (int)(k & 1);
        const Plan &q = plan[k];
        if (k + 1 < plan.size()) {                        // the next chunk's copy goes out before this chunk's scoring is waited for
            const int nb = (int)((k + 1) & 1);
            if (k + 1 >= 2) HIP_TRY(hipStreamWaitEvent(pp.copy, pp.consumed[nb], 0));
            if ((rc = feeder.feed(k + 1, pp.copy))) { (void)hipDeviceSynchronize(); return rc; }
            HIP_TRY(hipEventRecord(pp.copied[nb], pp.copy));
        }
        HIP_TRY(hipStreamWaitEvent(pp.comp, pp.copied[b], 0));
        rc = score_dev_impl(c, (const uint32_t *)dd[b].p, ebwt ? (const uint8_t *)de[b].p : nullptr, q.end - q.lo,
                            (const lime_cluster_t *)dc[b].p, q.j - q.i, n_reads, n_refs, d_sim, 0, q.lo, pp.comp);
        if (rc) { (void)hipDeviceSynchronize(); return rc; }
        HIP_TRY(hipEventRecord(pp.consumed[b], pp.comp));
        lime_stats_t st;
        if ((rc = lime_get_stats(c, &st, pp.comp))) { (void)hipDeviceSynchronize(); return rc; }
    }
    HIP_TRY(hipStreamSynchronize(pp.copy));
    return LIME_OK;
}

extern "C" int lime_score(lime_ctx *c, const uint32_t *da, const uint8_t *ebwt, uint64_t n,
                          const lime_cluster_t *clusters, uint64_t n_clusters, uint32_t n_reads,
                          uint32_t n_refs, uint8_t *sim)
{
    int rc = check_ctx(c, "lime_score"); if (rc) return rc;
    if (!sim || (n && !da) || (n_clusters && !clusters)) return fail(LIME_ERR_ARG, "lime_score: NULL array");
    if (!n_reads || !n_refs) return fail(LIME_ERR_ARG, "lime_score: n_reads and n_refs must be > 0");
    DevBuf ds;
    if ((rc = ds.alloc(lime_sim_bytes(n_reads, n_refs)))) return rc;
    if ((rc = score_in_chunks(c, da, ebwt, n, clusters, n_clusters, n_reads, n_refs, (uint8_t *)ds.p))) return rc;
    if ((rc = d2h_pageable(c, sim, ds.p, (size_t)n_reads * n_refs, nullptr))) return rc;
    return LIME_OK;
}

extern "C" int lime_fused(lime_ctx *c, const uint32_t *lcp, const uint32_t *da, const uint8_t *ebwt, uint64_t n,
                          uint32_t n_reads, uint32_t n_refs, uint32_t alpha, uint8_t *sim, uint64_t *n_clusters,
                          uint64_t *max_len)
{
    int rc = check_ctx(c, "lime_fused"); if (rc) return rc;
    if (!sim || (n && (!lcp || !da))) return fail(LIME_ERR_ARG, "lime_fused: NULL array");
    if (n > 4 * STREAM_CHUNK)                             // large collections: bounded HBM footprint, copy under scan
        return lime_fused_stream(c, lcp, da, ebwt, n, n_reads, n_refs, alpha, 0, sim, n_clusters, max_len);
    DevBuf dl, dd, de, ds;
    if ((rc = dl.upload(lcp, n * 4))) return rc;
    if ((rc = dd.upload(da, n * 4))) return rc;
    if (ebwt && (rc = de.upload(ebwt, n))) return rc;
    if ((rc = ds.alloc(lime_sim_bytes(n_reads, n_refs)))) return rc;
    rc = lime_fused_dev(c, (const uint32_t *)dl.p, (const uint32_t *)dd.p, ebwt ? (const uint8_t *)de.p : nullptr,
                        n, n, 1, n_reads, n_refs, alpha, (uint8_t *)ds.p, 1, nullptr);
    if (rc) return rc;
    lime_stats_t s;
    rc = lime_get_stats(c, &s, nullptr);
    if (n_clusters) *n_clusters = s.n_clusters;
    if (max_len) *max_len = s.max_len;
    if (rc) return rc;
    if ((rc = d2h_pageable(c, sim, ds.p, (size_t)n_reads * n_refs, nullptr))) return rc;
    return LIME_OK;
}

extern "C" int lime_choose(lime_ctx *c, const uint8_t *sim, uint32_t n_reads, uint32_t n_refs, uint8_t *row_max,
                           uint32_t *row_nnz)
{
    int rc = check_ctx(c, "lime_choose"); if (rc) return rc;
    if (!sim || !row_max || !row_nnz) return fail(LIME_ERR_ARG, "lime_choose: NULL array");
    if (!n_reads) return LIME_OK;
    DevBuf ds, dm, dz;
    if ((rc = ds.alloc(lime_sim_bytes(n_reads, n_refs)))) return rc;
    HIP_TRY(hipMemcpy(ds.p, sim, (size_t)n_reads * n_refs, hipMemcpyHostToDevice));
    if ((rc = dm.alloc(n_reads))) return rc;
    if ((rc = dz.alloc((size_t)n_reads * 4))) return rc;
    if ((rc = lime_choose_dev(c, (const uint8_t *)ds.p, n_reads, n_refs, (uint8_t *)dm.p, (uint32_t *)dz.p, nullptr))) return rc;
    HIP_TRY(hipMemcpy(row_max, dm.p, n_reads, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(row_nnz, dz.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost));
    return LIME_OK;
}

// Device memory -> freshly allocated pageable host memory (the (idRef, sim) lists: 0.7 GB when most of configs[2]'s rows pass): through two pinned
// slots, the copy of piece k + 1 under the host's copy of piece k by the staging threads.  The runtime's own staged copy does this on one thread,
// first-touch page faults included: 0.7 GB took 170 ms = 4 GB/s.
static int d2h_pageable(const lime_ctx *c, void *dst, const void *d_src, size_t bytes, hipStream_t st)
{
    constexpr size_t SLOT = (size_t)8 << 20, PIECE = (size_t)1 << 20;
    if (bytes < 4 * SLOT || (c && c->no_staging) || Feeder::pinned(dst)) {     // (pinned destinations: the copy engine writes them directly)
        HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return LIME_OK;
    }
    struct Ring {
        void *slot[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr};
        ~Ring() { for (int i = 0; i < 2; ++i) { if (slot[i]) (void)hipHostFree(slot[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); } }
    } r;
    for (int i = 0; i < 2; ++i) { HIP_TRY(hipHostMalloc(&r.slot[i], SLOT)); HIP_TRY(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming)); }
    const int threads = staging_threads(c);
    IoPool pool;
    if (threads > 1) pool.start(threads);
    const size_t n_chunks = (bytes + SLOT - 1) / SLOT;
    auto len_of = [&](size_t k) { return k + 1 < n_chunks ? SLOT : bytes - k * SLOT; };
    for (size_t k = 0; k <= n_chunks; ++k) {
        if (k < n_chunks) {
            HIP_TRY(hipMemcpyAsync(r.slot[k & 1], static_cast<const uint8_t *>(d_src) + k * SLOT, len_of(k), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(r.ev[k & 1], st));
        }
        if (k >= 1) {                                      // (slot (k - 1) & 1 is written again by copy k + 1, issued after this returns)
            const size_t j = k - 1, len = len_of(j);
            HIP_TRY(hipEventSynchronize(r.ev[j & 1]));
            std::vector<IoPool::Task> t;
            for (size_t o = 0; o < len; o += PIECE)
                t.push_back({static_cast<uint8_t *>(dst) + j * SLOT + o, static_cast<const uint8_t *>(r.slot[j & 1]) + o, len - o < PIECE ? len - o : PIECE});
            pool.run(std::move(t));
        }
    }
    return LIME_OK;
}

static int ensure_choose(lime_ctx *c, size_t dev_bytes, size_t host_bytes, hipStream_t st)
{
    int rc;
    if (dev_bytes > c->choose_cap) {
        HIP_TRY(hipStreamSynchronize(st)); c->choose_cap = 0;
        if ((rc = regrow(c->d_choose, dev_bytes))) return rc;
        c->choose_cap = dev_bytes;
    }
    if (host_bytes > c->h_choose_cap) {
        if (c->h_choose) { (void)hipHostFree(c->h_choose); c->h_choose = nullptr; c->h_choose_cap = 0; }
        HIP_TRY(hipHostMalloc(&c->h_choose, host_bytes));
        c->h_choose_cap = host_bytes;
    }
    return LIME_OK;
}
// the reference's test `float(max) / norm > beta`, in the reference's types (ClusterBWT_DA.cpp:404-406), once for each of the 256 values a row's
// maximum takes instead of once per read
static void choose_pass_table(uint32_t norm, float beta, bool pass[256])
{
    for (uint32_t v = 0; v < 256u; ++v) {
        const uint8_t mx = (uint8_t)v;
        const float top = static_cast<float>(mx) / norm;
        pass[v] = top > beta;
    }
}

// clusterChoose on the device, compact results to the host
extern "C" int lime_choose_pairs_dev(lime_ctx *c, const uint8_t *d_sim, uint32_t n_reads, uint32_t n_refs,
                                     uint32_t norm, float beta, uint8_t *row_max, uint64_t *row_off,
                                     lime_pair_t **pairs, uint64_t *n_pairs, void *stream)
{
    int rc = check_ctx(c, "lime_choose_pairs_dev"); if (rc) return rc;
    if (!pairs || !n_pairs || !row_off || (n_reads && (!d_sim || !row_max)))
        return fail(LIME_ERR_ARG, "lime_choose_pairs_dev: NULL array");
    *pairs = nullptr; *n_pairs = 0; row_off[0] = 0;
    if (!n_reads) return LIME_OK;
    if (misaligned(d_sim, 16)) return fail(LIME_ERR_ARG, "lime_choose_pairs_dev: d_sim must be 16-byte aligned (and lime_sim_bytes() long)");
    hipStream_t st = (hipStream_t)stream;
    DevBuf doff, dp;
    const size_t nz_off = ((size_t)n_reads + 15u) & ~(size_t)15u;      // row non-zero counts behind the row maxima, in both buffers
    if ((rc = ensure_choose(c, nz_off + (size_t)n_reads * 4, nz_off + (size_t)n_reads * 4, st))) return rc;
    launch_choose(d_sim, n_reads, n_refs, c->d_choose, reinterpret_cast<uint32_t *>(c->d_choose + nz_off), st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_choose, c->d_choose, nz_off + (size_t)n_reads * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint8_t *hmx = static_cast<const uint8_t *>(c->h_choose);
    const uint32_t *nnz = reinterpret_cast<const uint32_t *>(hmx + nz_off);
    bool pass[256];
    choose_pass_table(norm, beta, pass);
    uint64_t total = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint8_t mx = hmx[r];
        row_max[r] = mx;
        row_off[r] = total;
        if (pass[mx]) total += nnz[r];
    }
    row_off[n_reads] = total;
    *n_pairs = total;
    if (!total) return LIME_OK;
    if ((rc = doff.upload(row_off, ((size_t)n_reads + 1) * 8))) return rc;
    if ((rc = dp.alloc((size_t)total * sizeof(lime_pair_t)))) return rc;
    launch_gather_pairs(d_sim, n_reads, n_refs, (const uint64_t *)doff.p, (lime_pair_t *)dp.p, st);
    HIP_TRY(hipGetLastError());
    lime_pair_t *h = (lime_pair_t *)malloc((size_t)total * sizeof(lime_pair_t));
    if (!h) return fail(LIME_ERR_NOMEM, "lime_choose_pairs_dev: out of host memory");
    if ((rc = d2h_pageable(c, h, dp.p, (size_t)total * sizeof(lime_pair_t), st))) { free(h); return rc; }
    *pairs = h;
    return LIME_OK;
}

// ClusterLCP scan + clusterAnalyze + clusterChoose on device-resident arrays in one call.  Where the binned update path serves the pass with its
// second level by tiles (tables beyond 64 MB) the TABLE IS NEVER WRITTEN: the pass stops at the binned records (the long clusters' updates as
// records of their own, bucketed by region), k_sort_tiles sorts them into tile rows once, and k_apply_tiles builds every 64 KB region in LDS
// twice -- first for the rows' maxima and non-zero counts (clusterChoose's row scan, ClusterBWT_DA.cpp:385-402), then, after the host's test
// `float(max) / norm > beta` (:404-406), for the passing rows' (idRef, sim) lists (:408-423; regions without a passing row are skipped).  Against
// table + k_choose + k_gather_pairs that saves writing T bytes and reading them once or twice.  Elsewhere (small tables, short passes, n_refs < 256):
// the table is built and scanned as before.  Outputs as lime_choose_pairs_dev; *stats (may be NULL) as lime_get_stats.
extern "C" int lime_fused_choose_dev(lime_ctx *c, const uint32_t *d_lcp, const uint32_t *d_da, const uint8_t *d_ebwt, uint64_t n,
                                     uint32_t n_reads, uint32_t n_refs, uint32_t alpha, uint32_t norm, float beta,
                                     uint8_t *row_max, uint64_t *row_off, lime_pair_t **pairs, uint64_t *n_pairs,
                                     lime_stats_t *stats, void *stream)
{
    int rc = check_ctx(c, "lime_fused_choose_dev"); if (rc) return rc;
    if (!pairs || !n_pairs || !row_off || !row_max) return fail(LIME_ERR_ARG, "lime_fused_choose_dev: NULL output");
    if (!n_reads || !n_refs) return fail(LIME_ERR_ARG, "lime_fused_choose_dev: n_reads and n_refs must be > 0");
    *pairs = nullptr; *n_pairs = 0; row_off[0] = 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t sim_bytes = lime_sim_bytes(n_reads, n_refs);
    const bool bin_fits = !(sim_bytes > ((size_t)BIN_MAX << BIN_SHIFT_MAX) || sim_bytes >= (1ull << CELL_BITS) || sim_bytes > ((uint64_t)MAX_SUB << 32));
    uint32_t n_bins = 0, bin_shift = REGION_SHIFT;
    if (bin_fits) bin_layout(c, sim_bytes, &n_bins, &bin_shift);
    // (lime_set_option "choose_free": 1 = without the table wherever the layout has a second level, 0 = never)
    bool table_free = bin_fits && c->by_tiles && bin_shift > REGION_SHIFT && n && c->upd_pref != 0 && n_refs < MAX_REFS &&
                      (c->choose_free >= 0 ? c->choose_free != 0 : (n_refs >= 256u && n >= (1u << 24)));
    lime_stats_t s;
    memset(&s, 0, sizeof s);
    if (!table_free) {
        DevBuf ds;
        if ((rc = ds.alloc(sim_bytes))) return rc;
        if ((rc = lime_fused_dev(c, d_lcp, d_da, d_ebwt, n, n, 1, n_reads, n_refs, alpha, (uint8_t *)ds.p, 1, stream))) return rc;
        rc = lime_get_stats(c, &s, stream);
        if (stats) *stats = s;
        if (rc) return rc;
        return lime_choose_pairs_dev(c, (const uint8_t *)ds.p, n_reads, n_refs, norm, beta, row_max, row_off, pairs, n_pairs, stream);
    }
    ++c->n_table_free;
    if ((rc = fused_dev_impl(c, d_lcp, d_da, d_ebwt, n, n, 1, n_reads, n_refs, alpha, nullptr, 1, false, st, nullptr, false, true))) return rc;
    rc = lime_get_stats(c, &s, stream);                          // (waits; repeats the pass if the record pool or the long clusters' list was too small)
    if (stats) *stats = s;
    if (rc) return rc;
    uint32_t nb = 0;
    HIP_TRY(hipMemcpyAsync(&nb, c->d_bigrec_n, sizeof nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (nb > c->bigrec_cap) return fail(LIME_ERR_NOMEM, "more update records of long clusters (%u) than their list holds (%u)", nb, c->bigrec_cap);
    const uint32_t n_regions = (uint32_t)((sim_bytes + ((size_t)1 << REGION_SHIFT) - 1) >> REGION_SHIFT);
    DevBuf bcnt, bcur, boff, bout, doff, dp;
    // the ctx's scratch: [region words 16 R][row max 4 n][row nnz 4 n][last nnz 4 R]; the rows' two arrays come back in one copy
    const size_t rows_off = (size_t)n_regions * 16, rows_bytes = (size_t)n_reads * 8, last_off = rows_off + rows_bytes;
    if ((rc = ensure_choose(c, last_off + (size_t)n_regions * 4, rows_bytes, st))) return rc;
    uint32_t *dmax = reinterpret_cast<uint32_t *>(c->d_choose + rows_off), *dnnz = dmax + n_reads;
    void *drr = c->d_choose;
    HIP_TRY(hipMemsetAsync(dmax, 0, rows_bytes, st));
    ApplyFin fin;
    memset(&fin, 0, sizeof fin);
    fin.n_refs = n_refs; fin.table_bytes = (uint64_t)n_reads * n_refs;
    fin.row_max = dmax; fin.row_nnz = dnnz; fin.last_nnz = reinterpret_cast<uint32_t *>(c->d_choose + last_off);
    fin.region_rows = (const uint4 *)drr;
    launch_region_rows(n_regions, n_refs, fin.table_bytes, nullptr, drr, st);
    if (nb) {
        if ((rc = bcnt.alloc((size_t)n_regions * 4)) || (rc = bcur.alloc((size_t)n_regions * 4)) || (rc = boff.alloc(((size_t)n_regions + 1) * 8)) ||
            (rc = bout.alloc((size_t)nb * 8))) return rc;
        launch_bigrec_buckets(c->d_bigrec, nb, n_regions, (uint32_t *)bcnt.p, (uint32_t *)bcur.p, (uint64_t *)boff.p, (uint64_t *)bout.p, st);
        fin.big_off = (const uint64_t *)boff.p; fin.bigrecs = (const uint64_t *)bout.p;
    }
    const double expect = (double)s.n_updates;
    uint16_t *rows = reinterpret_cast<uint16_t *>(c->d_pool);
    launch_sort_tiles(c->d_recs, c->d_binbase, n_bins, bin_shift, c->d_tbase, c->d_tidx, rows, st, big_rows_of(c, expect));      // (k_tile_bases inside: the pass stopped at the records and numbered no tiles)
    launch_apply_tiles_fin(1, sim_bytes, bin_shift, c->d_tbase, c->d_tidx, rows, many_records_of(c, expect), fin, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_choose, dmax, rows_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t *hm = static_cast<const uint32_t *>(c->h_choose), *hz = hm + n_reads;
    bool pass[256];
    choose_pass_table(norm, beta, pass);                         // the reference's test, in the reference's types (ClusterBWT_DA.cpp:404-406)
    uint64_t total = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint8_t mx = (uint8_t)hm[r];
        row_max[r] = mx;
        row_off[r] = total;
        if (pass[mx]) total += hz[r];
    }
    row_off[n_reads] = total;
    *n_pairs = total;
    if (!total) return LIME_OK;
    if ((rc = doff.upload(row_off, ((size_t)n_reads + 1) * 8))) return rc;
    if ((rc = dp.alloc((size_t)total * sizeof(lime_pair_t)))) return rc;
    fin.row_off = (const uint64_t *)doff.p; fin.pairs = (lime_pair_t *)dp.p;
    launch_region_rows(n_regions, n_refs, fin.table_bytes, fin.row_off, drr, st);          // (now with the regions that have nothing to gather marked)
    launch_apply_tiles_fin(2, sim_bytes, bin_shift, c->d_tbase, c->d_tidx, rows, many_records_of(c, expect), fin, st);
    HIP_TRY(hipGetLastError());
    lime_pair_t *h = (lime_pair_t *)malloc((size_t)total * sizeof(lime_pair_t));
    if (!h) return fail(LIME_ERR_NOMEM, "lime_fused_choose_dev: out of host memory");
    if ((rc = d2h_pageable(c, h, dp.p, (size_t)total * sizeof(lime_pair_t), st))) { free(h); return rc; }
    *pairs = h;
    return LIME_OK;
}

extern "C" int lime_score_choose(lime_ctx *c, const uint32_t *da, const uint8_t *ebwt, uint64_t n,
                                 const lime_cluster_t *clusters, uint64_t n_clusters, uint32_t n_reads,
                                 uint32_t n_refs, uint32_t norm, float beta, uint8_t *row_max, uint64_t *row_off,
                                 lime_pair_t **pairs, uint64_t *n_pairs, uint8_t *sim)
{
    int rc = check_ctx(c, "lime_score_choose"); if (rc) return rc;
    if ((n && !da) || (n_clusters && !clusters)) return fail(LIME_ERR_ARG, "lime_score_choose: NULL array");
    if (!n_reads || !n_refs) return fail(LIME_ERR_ARG, "lime_score_choose: n_reads and n_refs must be > 0");
    DevBuf ds;
    if ((rc = ds.alloc(lime_sim_bytes(n_reads, n_refs)))) return rc;
    if ((rc = score_in_chunks(c, da, ebwt, n, clusters, n_clusters, n_reads, n_refs, (uint8_t *)ds.p))) return rc;
    if ((rc = lime_choose_pairs_dev(c, (const uint8_t *)ds.p, n_reads, n_refs, norm, beta, row_max, row_off, pairs,
                                    n_pairs, nullptr))) return rc;
    if (sim && (rc = d2h_pageable(c, sim, ds.p, (size_t)n_reads * n_refs, nullptr))) return rc;
    return LIME_OK;
}

// ---- clusterAnalyze + clusterChoose on several GPUs of one process ------------------------------------
// The cluster list is cut by position into n_dev parts of equal symbol counts (the reference cuts it by cluster
// count over OpenMP threads, ClusterBWT_DA.cpp:641-648; any cut gives the same table); device k scores its part
// into its own table; ONE reduce-scatter (RCCL, sum modulo 256) leaves device k with the block of read rows
// [k * rpd, (k+1) * rpd); each device runs the row scan and the list compaction on its block; the host appends
// the blocks' results in row order.  A host thread per device does the uploads and launches.
int lime_internal_reduce_scatter(int n_dev, const int *devs, uint8_t *const *d_sim, uint8_t *const *d_blk, size_t blk);   // lime_comm.cpp

extern "C" int lime_score_choose_multi(int n_dev, const int *devices, const uint32_t *da, const uint8_t *ebwt, uint64_t n,
                                       const lime_cluster_t *clusters, uint64_t n_clusters, uint32_t n_reads, uint32_t n_refs,
                                       uint32_t norm, float beta, uint8_t *row_max, uint64_t *row_off, lime_pair_t **pairs,
                                       uint64_t *n_pairs)
{
    if (n_dev < 1 || !pairs || !n_pairs || !row_off || (n_reads && !row_max) || (n && !da) || (n_clusters && !clusters))
        return fail(LIME_ERR_ARG, "lime_score_choose_multi: bad argument");
    if (!n_reads || !n_refs) return fail(LIME_ERR_ARG, "lime_score_choose_multi: n_reads and n_refs must be > 0");
    if (n_dev > lime_device_count()) return fail(LIME_ERR_ARG, "lime_score_choose_multi: %d devices asked, %d visible", n_dev, lime_device_count());
    *pairs = nullptr; *n_pairs = 0; row_off[0] = 0;
    std::vector<int> devs(n_dev);
    for (int k = 0; k < n_dev; ++k) devs[k] = devices ? devices[k] : k;
    // row blocks: a multiple of 16 rows each, so that every block starts 16-byte aligned whatever n_refs is
    const uint64_t rpd = (((uint64_t)n_reads + n_dev - 1) / n_dev + 15u) & ~15ull;
    const size_t blk = (size_t)rpd * n_refs, tbl = blk * (size_t)n_dev;
    // the clusters in position order, cut where the running symbol count passes k/n_dev of the total
    std::vector<lime_cluster_t> order;
    const lime_cluster_t *cl = clusters;
    bool sorted = true;
    uint64_t total_len = 0;
    for (uint64_t i = 0; i < n_clusters; ++i) { if (i && clusters[i].pStart < clusters[i - 1].pStart) sorted = false; total_len += clusters[i].len; }
    if (!sorted) {
        order.assign(clusters, clusters + n_clusters);
        std::sort(order.begin(), order.end(), [](const lime_cluster_t &x, const lime_cluster_t &y) { return x.pStart < y.pStart; });
        cl = order.data();
    }
    std::vector<uint64_t> cut(n_dev + 1, n_clusters);
    cut[0] = 0;
    { uint64_t run = 0; int k = 1; for (uint64_t i = 0; i < n_clusters && k < n_dev; ++i) { run += cl[i].len; while (k < n_dev && run * (uint64_t)n_dev >= total_len * (uint64_t)k) cut[k++] = i + 1; } }
    struct Dev { lime_ctx *ctx = nullptr; uint8_t *sim = nullptr, *blkp = nullptr; int rc = LIME_OK; std::string err; };
    std::vector<Dev> dv(n_dev);
    auto cleanup = [&]() {
        for (int k = 0; k < n_dev; ++k) { (void)hipSetDevice(devs[k]); (void)hipFree(dv[k].sim); (void)hipFree(dv[k].blkp); if (dv[k].ctx) lime_shutdown(dv[k].ctx); }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < n_dev; ++k)
        th.emplace_back([&, k]() {
            Dev &d = dv[k];
            auto bad = [&](int rc, const char *what) { d.rc = rc; d.err = std::string(what) + ": " + lime_last_error(); };
            if (hipSetDevice(devs[k]) != hipSuccess) { d.rc = LIME_ERR_HIP; d.err = "hipSetDevice"; return; }
            int rc = lime_init(devs[k], &d.ctx);
            if (rc) { bad(rc, "lime_init"); return; }
            if (hipMalloc(&d.sim, tbl + 16) != hipSuccess || hipMalloc(&d.blkp, blk + 16) != hipSuccess) { d.rc = LIME_ERR_NOMEM; d.err = "hipMalloc of the table"; return; }
            if (hipMemset(d.sim, 0, tbl + 16) != hipSuccess) { d.rc = LIME_ERR_HIP; d.err = "hipMemset"; return; }
            rc = score_in_chunks(d.ctx, da, ebwt, n, cl + cut[k], cut[k + 1] - cut[k], n_reads, n_refs, d.sim);
            if (rc) { bad(rc, "scoring"); return; }
            if (hipDeviceSynchronize() != hipSuccess) { d.rc = LIME_ERR_HIP; d.err = "hipDeviceSynchronize"; }
        });
    for (auto &t : th) t.join();
    for (int k = 0; k < n_dev; ++k) if (dv[k].rc) { const int rc = dv[k].rc; const std::string e = dv[k].err; cleanup(); return fail(rc, "device %d: %s", devs[k], e.c_str()); }
    int rc = LIME_OK;
    if (n_dev > 1 || dv[0].ctx->force_rccl) {
        std::vector<uint8_t *> sims(n_dev), blks(n_dev);
        for (int k = 0; k < n_dev; ++k) { sims[k] = dv[k].sim; blks[k] = dv[k].blkp; }
        if ((rc = lime_internal_reduce_scatter(n_dev, devs.data(), sims.data(), blks.data(), blk))) { cleanup(); return fail(rc, "%s", lime_comm_error()); }
    } else {
        (void)hipSetDevice(devs[0]);
        if (hipMemcpy(dv[0].blkp, dv[0].sim, blk, hipMemcpyDeviceToDevice) != hipSuccess) { cleanup(); return fail(LIME_ERR_HIP, "hipMemcpy"); }
    }
    // row scan + compaction per block, appended in row order
    std::vector<lime_pair_t *> pp(n_dev, nullptr);
    std::vector<uint64_t> np(n_dev, 0);
    std::vector<std::vector<uint64_t>> off(n_dev);
    uint64_t total = 0;
    for (int k = 0; k < n_dev && !rc; ++k) {
        const uint64_t r0 = rpd * (uint64_t)k;
        if (r0 >= n_reads) break;
        const uint32_t rows = (uint32_t)(n_reads - r0 < rpd ? n_reads - r0 : rpd);
        off[k].resize((size_t)rows + 2);
        (void)hipSetDevice(devs[k]);
        rc = lime_choose_pairs_dev(dv[k].ctx, dv[k].blkp, rows, n_refs, norm, beta, row_max + r0, off[k].data(), &pp[k], &np[k], nullptr);
        if (!rc) { for (uint32_t r = 0; r < rows; ++r) row_off[r0 + r] = total + off[k][r]; total += np[k]; }
    }
    std::string err = rc ? lime_last_error() : "";
    if (!rc) {
        row_off[n_reads] = total;
        lime_pair_t *all = total ? (lime_pair_t *)malloc((size_t)total * sizeof(lime_pair_t)) : nullptr;
        if (total && !all) { rc = LIME_ERR_NOMEM; err = "out of host memory"; }
        else {
            uint64_t at = 0;
            for (int k = 0; k < n_dev; ++k) if (np[k]) { memcpy(all + at, pp[k], (size_t)np[k] * sizeof(lime_pair_t)); at += np[k]; }
            *pairs = all; *n_pairs = total;
        }
    }
    for (int k = 0; k < n_dev; ++k) free(pp[k]);
    cleanup();
    return rc ? fail(rc, "%s", err.c_str()) : LIME_OK;
}

#ifdef LIME_DEBUG_CNT
// debug builds only: the per-window counts of accepted clusters the last scan left (see k_scan)
extern "C" int lime_debug_tile_counts(lime_ctx *c, uint32_t *out, uint32_t n)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, c->d_tile_cnt, (size_t)n * 4, hipMemcpyDeviceToHost));
    return LIME_OK;
}
#endif

// ---- pure host helpers ------------------------------------------------------------------
extern "C" uint8_t lime_sym_index(uint8_t b) { return (uint8_t)sym_index(b); }

extern "C" uint8_t lime_pair_score(const uint8_t cr[16], const uint8_t cg[16])
{
    uint32_t r[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) { r[i >> 2] |= (uint32_t)cr[i] << ((i & 3) * 8); g[i >> 2] |= (uint32_t)cg[i] << ((i & 3) * 8); }
    return (uint8_t)pair_score(r, g);
}
